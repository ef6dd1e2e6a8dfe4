// zslab_rccl.cpp -- kfusion::cuda::ZSlabComm: the RCCL side of the Z-slab sharding (kfusion/cuda/zslab.hpp).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <kfusion/cuda/zslab.hpp>

using namespace kfusion;
using namespace kfusion::cuda;

#define ZS_NCCL(expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) { std::fprintf(stderr, "RCCL: %s at %s:%d\n", ncclGetErrorString(r__), __FILE__, __LINE__); std::exit(1); } } while (0)
#define ZS_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { std::fprintf(stderr, "HIP: %s at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); std::exit(1); } } while (0)

ZSlabComm::ZSlabComm(int rank, int world, const std::string& id_path) : rank_(rank), world_(world), comm_(nullptr), stream_(nullptr)
{
    if (world < 1 || world > 255 || rank < 0 || rank >= world) { std::fprintf(stderr, "ZSlabComm: rank %d of %d (1..255 ranks: the merge key carries the rank in 8 bits)\n", rank, world); std::exit(1); }
    ncclUniqueId id;
    if (rank == 0) {
        ZS_NCCL(ncclGetUniqueId(&id));
        const std::string tmp = id_path + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) { std::perror("ZSlabComm id file"); std::exit(1); }
        std::fclose(f);
        if (std::rename(tmp.c_str(), id_path.c_str()) != 0) { std::perror("ZSlabComm id rename"); std::exit(1); }   // atomic publish
    } else {
        for (int tries = 0;; ++tries) {
            FILE* f = std::fopen(id_path.c_str(), "rb");
            if (f) { const size_t n = std::fread(&id, sizeof(id), 1, f); std::fclose(f); if (n == 1) break; }
            if (tries > 6000) { std::fprintf(stderr, "ZSlabComm: no RCCL id at %s after 60 s\n", id_path.c_str()); std::exit(1); }
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    ncclComm_t c;
    ZS_NCCL(ncclCommInitRank(&c, world, id, rank));
    comm_ = c;
    token_.create(1);
}

ZSlabComm::~ZSlabComm()
{
    if (comm_) { (void)hipDeviceSynchronize(); (void)ncclCommDestroy((ncclComm_t)comm_); }
}

void ZSlabComm::slabRange(int Z, int rank, int world, int& z_own0, int& z_own_n)
{
    const int per = (Z % 8 == 0) ? ((Z / 8 + world - 1) / world) * 8 : (Z + world - 1) / world;
    const int lo = std::min(Z, rank * per), hi = std::min(Z, lo + per);
    z_own0 = lo; z_own_n = hi - lo;
}

int ZSlabComm::haloPlanes(float trunc_dist, float step_factor, float delta_factor, float voxel_z)
{
    return (int)std::ceil((double)trunc_dist * step_factor / voxel_z + delta_factor) + 2;
}

bool ZSlabComm::partitionOk(int Z, int world, int halo, std::string* why)
{
    for (int r = 0; r < world; ++r) {
        int z0, n; slabRange(Z, r, world, z0, n);
        if (n <= 0) { if (why) *why = "rank " + std::to_string(r) + " would own no plane"; return false; }
        if (world > 1 && n < halo) { if (why) *why = "rank " + std::to_string(r) + " owns fewer planes than the halo"; return false; }
    }
    return true;
}

void ZSlabComm::broadcast(void* device_ptr, size_t bytes, int root)
{
    if (world_ == 1 || !bytes) return;
    ZS_NCCL(ncclBroadcast(device_ptr, device_ptr, bytes, ncclUint8, root, (ncclComm_t)comm_, (hipStream_t)stream_));
}

void ZSlabComm::exchangeHalos(TsdfVolume& slab, int halo)
{
    if (world_ == 1) return;
    const Vec3i d = slab.getDims();
    const size_t plane = (size_t)d[0] * d[1];                      // voxels per plane (4 bytes each)
    int* base = slab.data().ptr<int>();
    const int lo_local = slab.slabOwn0() - slab.slabStore0(), hi_local = lo_local + slab.slabOwnN();
    const int n_lo = lo_local, n_hi = slab.slabStoreN() - hi_local;
    if ((rank_ > 0 && (n_lo != halo || slab.slabOwnN() < halo)) || (rank_ < world_ - 1 && (n_hi != halo || slab.slabOwnN() < halo))) {
        std::fprintf(stderr, "ZSlabComm::exchangeHalos: rank %d slab does not hold %d halo planes (ask partitionOk first)\n", rank_, halo); std::exit(1);
    }
    ncclComm_t c = (ncclComm_t)comm_; hipStream_t st = (hipStream_t)stream_;
    ZS_NCCL(ncclGroupStart());
    if (rank_ > 0) {                                                // lower neighbour: my first own planes out, its last ones in
        ZS_NCCL(ncclSend(base + (size_t)lo_local * plane, (size_t)halo * plane, ncclInt32, rank_ - 1, c, st));
        ZS_NCCL(ncclRecv(base, (size_t)n_lo * plane, ncclInt32, rank_ - 1, c, st));
    }
    if (rank_ < world_ - 1) {
        ZS_NCCL(ncclSend(base + (size_t)(hi_local - halo) * plane, (size_t)halo * plane, ncclInt32, rank_ + 1, c, st));
        ZS_NCCL(ncclRecv(base + (size_t)hi_local * plane, (size_t)n_hi * plane, ncclInt32, rank_ + 1, c, st));
    }
    ZS_NCCL(ncclGroupEnd());
}

void ZSlabComm::raycast(TsdfVolume& slab, const Affine3f& camera_pose, const Intr& intr, int cols, int rows, Cloud& points, Normals& normals, int dst)
{
    ncclComm_t c = (ncclComm_t)comm_; hipStream_t st = (hipStream_t)stream_;
    const size_t px = (size_t)cols * rows;
    slab.raycastMarch(camera_pose, intr, cols, rows, (unsigned)rank_, keys64_, vertex_);
    if (world_ > 1) {
        ZS_NCCL(ncclAllReduce(keys64_.ptr(), keys64_.ptr(), px, ncclInt64, ncclMin, c, st));            // first event per pixel, over ranks
        TsdfVolume::raycastSelect(keys64_, (unsigned)rank_, vertex_, cols, rows);
        ZS_NCCL(ncclAllReduce(vertex_.ptr(), vertex_.ptr(), px * 4, ncclInt32, ncclSum, c, st));        // the winners' vertex bits
    }
    points_.create(px); normals_.create(px);
    points = Cloud(rows, cols, points_.ptr(), (size_t)cols * sizeof(Point));
    normals = Normals(rows, cols, normals_.ptr(), (size_t)cols * sizeof(Normal));
    slab.raycastShade(camera_pose, intr, vertex_, keys64_, points, normals);
    if (world_ > 1) {
        ZS_NCCL(ncclReduce(points_.ptr(), points_.ptr(), px * 4, ncclInt32, ncclSum, dst, c, st));
        ZS_NCCL(ncclReduce(normals_.ptr(), normals_.ptr(), px * 4, ncclInt32, ncclSum, dst, c, st));
    }
}

void ZSlabComm::barrier()
{
    if (world_ > 1) ZS_NCCL(ncclAllReduce(token_.ptr(), token_.ptr(), 1, ncclInt32, ncclSum, (ncclComm_t)comm_, (hipStream_t)stream_));
    ZS_HIP(hipDeviceSynchronize());
}
