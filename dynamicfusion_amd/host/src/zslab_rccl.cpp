// zslab_rccl.cpp -- kfusion::cuda::ZSlabComm: the RCCL side of the Z-slab sharding (kfusion/cuda/zslab.hpp).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <kfusion/cuda/zslab.hpp>

using namespace kfusion;
using namespace kfusion::cuda;

// Errors do not end the process: the failing call records what failed (lastError()) and returns false; after a failed collective the
// communicator is unusable (the other ranks may be blocked in it) and every later call fails at once.
#define ZS_NCCL(expr) do { ncclResult_t r__ = (expr); if (r__ != ncclSuccess) return fail(std::string("RCCL: ") + ncclGetErrorString(r__) + " in " #expr); } while (0)
#define ZS_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) return fail(std::string("HIP: ") + hipGetErrorString(e__) + " in " #expr); } while (0)

bool ZSlabComm::fail(const std::string& what)
{
    error_ = what; ok_ = false;
    return false;
}

// The rendezvous file carries {magic, nonce, ncclUniqueId}.  Rank 0 removes whatever an earlier run left under the name BEFORE it
// publishes (atomically, by rename) and again once every rank holds the communicator; a reader only accepts a file whose nonce is the
// one the launcher gave every rank (DFUSION_ZSLAB_NONCE, default 0) -- so ranks of a new run cannot pick up the id of an old one.
namespace { struct IdFile { unsigned long long magic, nonce; ncclUniqueId id; }; const unsigned long long ID_MAGIC = 0x44465a534c414231ull; }

ZSlabComm::ZSlabComm(int rank, int world, const std::string& id_path) : rank_(rank), world_(world), comm_(nullptr), stream_(nullptr), ok_(true)
{
    if (world < 1 || world > 128 || rank < 0 || rank >= world) { fail("ZSlabComm: rank " + std::to_string(rank) + " of " + std::to_string(world) + " (1..128 ranks: the merge key carries the rank in 7 bits)"); return; }
    const char* ne = std::getenv("DFUSION_ZSLAB_NONCE");
    IdFile f; f.magic = ID_MAGIC; f.nonce = ne ? std::strtoull(ne, nullptr, 10) : 0ull;
    if (rank == 0) {
        (void)std::remove(id_path.c_str());                                  // a stale id of an earlier run
        if (ncclGetUniqueId(&f.id) != ncclSuccess) { fail("ncclGetUniqueId"); return; }
        const std::string tmp = id_path + ".tmp." + std::to_string((long long)getpid());
        FILE* fp = std::fopen(tmp.c_str(), "wb");
        if (!fp || std::fwrite(&f, sizeof(f), 1, fp) != 1) { if (fp) std::fclose(fp); fail("ZSlabComm: cannot write " + tmp); return; }
        std::fclose(fp);
        if (std::rename(tmp.c_str(), id_path.c_str()) != 0) { fail("ZSlabComm: cannot publish " + id_path); return; }   // atomic publish
    } else {
        for (int tries = 0;; ++tries) {
            IdFile g;
            FILE* fp = std::fopen(id_path.c_str(), "rb");
            if (fp) {
                const size_t n = std::fread(&g, sizeof(g), 1, fp); std::fclose(fp);
                if (n == 1 && g.magic == ID_MAGIC && g.nonce == f.nonce) { f = g; break; }
            }
            if (tries > 6000) { fail("ZSlabComm: no RCCL id for this run at " + id_path + " after 60 s"); return; }
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    ncclComm_t c;
    if (ncclCommInitRank(&c, world, f.id, rank) != ncclSuccess) { fail("ncclCommInitRank"); return; }
    comm_ = c;
    token_.create(1);
    if (rank == 0) (void)std::remove(id_path.c_str());                       // every rank is in: the file has done its job
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

std::vector<int> ZSlabComm::slabBounds(int Z, int world, int halo, const std::vector<double>& weights)
{
    std::vector<int> b;
    if ((int)weights.size() != Z) { for (int r = 0; r < world; ++r) { int z0, n; slabRange(Z, r, world, z0, n); b.push_back(z0); } b.push_back(Z); return b; }
    const int step = (Z % 8 == 0) ? 8 : 1;
    int min_planes = std::max(halo, step);
    min_planes = (min_planes + step - 1) / step * step;
    std::vector<double> cum((size_t)Z + 1, 0.0);
    for (int z = 0; z < Z; ++z) cum[z + 1] = cum[z] + weights[z];
    const double total = cum[Z] > 0 ? cum[Z] : 1.0;
    b.push_back(0);
    for (int r = 1; r < world; ++r) {
        const double target = total * r / world;
        int z = (int)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());      // numpy.searchsorted(cum, target)
        z = (int)std::nearbyint((double)z / step) * step;                                    // Python's round(): half to even
        z = std::max(z, b.back() + min_planes);
        z = std::min(z, Z - (world - r) * min_planes);
        b.push_back(z);
    }
    b.push_back(Z);
    return b;
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

bool ZSlabComm::broadcast(void* device_ptr, size_t bytes, int root)
{
    if (!ok_) return false;
    if (world_ == 1 || !bytes) return true;
    ZS_NCCL(ncclBroadcast(device_ptr, device_ptr, bytes, ncclUint8, root, (ncclComm_t)comm_, (hipStream_t)stream_));
    return true;
}

bool ZSlabComm::exchangeHalos(TsdfVolume& slab, int halo)
{
    if (!ok_) return false;
    if (world_ == 1) return true;
    const Vec3i d = slab.getDims();
    const size_t plane = (size_t)d[0] * d[1];                      // voxels per plane (4 bytes each)
    int* base = slab.data().ptr<int>();
    const int lo_local = slab.slabOwn0() - slab.slabStore0(), hi_local = lo_local + slab.slabOwnN();
    const int n_lo = lo_local, n_hi = slab.slabStoreN() - hi_local;
    // (a local precondition, checked the same way on every rank by partitionOk: no collective has been entered yet)
    if ((rank_ > 0 && (n_lo != halo || slab.slabOwnN() < halo)) || (rank_ < world_ - 1 && (n_hi != halo || slab.slabOwnN() < halo)))
        return fail("ZSlabComm::exchangeHalos: the slab does not hold " + std::to_string(halo) + " halo planes (ask partitionOk first)");
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
    return true;
}

bool ZSlabComm::raycast(TsdfVolume& slab, const Affine3f& camera_pose, const Intr& intr, int cols, int rows, Cloud& points, Normals& normals, int dst)
{
    if (!ok_) return false;
    ncclComm_t c = (ncclComm_t)comm_; hipStream_t st = (hipStream_t)stream_;
    const size_t px = (size_t)cols * rows;
    slab.raycastMarch(camera_pose, intr, cols, rows, (unsigned)rank_, keys64_);
    // ONE merge collective: the per-pixel MIN of the keys is the first event along every ray, its owner and its Ts (dfusion.h)
    if (world_ > 1) ZS_NCCL(ncclAllReduce(keys64_.ptr(), keys64_.ptr(), px, ncclInt64, ncclMin, c, st));
    out_.create(2 * px);                                            // normals (what the reduce sums), then the points
    normals = Normals(rows, cols, out_.ptr(), (size_t)cols * sizeof(Normal));
    points = Cloud(rows, cols, out_.ptr() + px, (size_t)cols * sizeof(Point));
    slab.raycastShadeNormals(camera_pose, intr, keys64_, normals);
    // only the normals cross GPUs (4.9 MB at 640 x 480; every summand but one is integer zero): the points follow from the merged keys
    if (world_ > 1) ZS_NCCL(ncclReduce(out_.ptr(), out_.ptr(), px * 4, ncclInt32, ncclSum, dst, c, st));
    if (rank_ == dst) slab.raycastPointsOfKeys(camera_pose, intr, keys64_, normals, points);
    return true;
}

bool ZSlabComm::barrier()
{
    if (!ok_) return false;
    if (world_ > 1) ZS_NCCL(ncclAllReduce(token_.ptr(), token_.ptr(), 1, ncclInt32, ncclSum, (ncclComm_t)comm_, (hipStream_t)stream_));
    ZS_HIP(hipDeviceSynchronize());
    return true;
}
