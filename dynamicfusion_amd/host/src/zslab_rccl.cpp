// zslab_rccl.cpp -- kfusion::cuda::ZSlabComm: the RCCL side of the Z-slab sharding (kfusion/cuda/zslab.hpp).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <new>
#include <thread>
#include <vector>
#include <cerrno>
#include <csignal>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <kfusion/cuda/zslab.hpp>
#include "dfusion.h"

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

ZSlabComm::ZSlabComm(int rank, int world, const std::string& id_path, Backend backend)
    : backend_(backend), seg_(nullptr), seg_bytes_(0), slot_bytes_(0), rank_(rank), world_(world), comm_(nullptr), stream_(nullptr), ok_(true)
{
    if (world < 1 || world > 128 || rank < 0 || rank >= world) { fail("ZSlabComm: rank " + std::to_string(rank) + " of " + std::to_string(world) + " (1..128 ranks: the merge key carries the rank in 7 bits)"); return; }
    if (backend_ == FROM_ENV) { const char* b = std::getenv("DFUSION_ZSLAB_BACKEND"); backend_ = (b && !std::strcmp(b, "host")) ? HOST_STAGED : RCCL; }
    { const char* m = std::getenv("DFUSION_ZSLAB_MERGE"); row_merge_ = (m && !std::strcmp(m, "a2a")) ? ALL_TO_ALL : REDUCE_SCATTER; }
    { const char* m = std::getenv("DFUSION_ZSLAB_KEY_MERGE"); key_merge_ = (m && !std::strcmp(m, "ring")) ? KEYS_RING : KEYS_DIRECT; }
    { const char* m = std::getenv("DFUSION_ZSLAB_BCAST"); bcast_direct_ = !(m && !std::strcmp(m, "ring")); }
    // DFUSION_ZSLAB_FORCE_COLLECTIVES=1: issue the RCCL calls of the casts and the broadcast with ONE rank too (every one of them is the
    // identity there: an all-reduce over one rank, a send to oneself) -- a dry run of the calls' types, counts and grouping on a one-GPU box
    { const char* m = std::getenv("DFUSION_ZSLAB_FORCE_COLLECTIVES"); force_ = m && m[0] == '1' && backend_ != HOST_STAGED; }
    if (backend_ == HOST_STAGED) initHost(id_path); else initRccl(id_path);
}

bool ZSlabComm::initRccl(const std::string& id_path)
{
    const int rank = rank_, world = world_;
    const char* ne = std::getenv("DFUSION_ZSLAB_NONCE");
    IdFile f; f.magic = ID_MAGIC; f.nonce = ne ? std::strtoull(ne, nullptr, 10) : 0ull;
    if (rank == 0) {
        (void)std::remove(id_path.c_str());                                  // a stale id of an earlier run
        if (ncclGetUniqueId(&f.id) != ncclSuccess) return fail("ncclGetUniqueId");
        const std::string tmp = id_path + ".tmp." + std::to_string((long long)getpid());
        FILE* fp = std::fopen(tmp.c_str(), "wb");
        if (!fp || std::fwrite(&f, sizeof(f), 1, fp) != 1) { if (fp) std::fclose(fp); return fail("ZSlabComm: cannot write " + tmp); }
        std::fclose(fp);
        if (std::rename(tmp.c_str(), id_path.c_str()) != 0) return fail("ZSlabComm: cannot publish " + id_path);   // atomic publish
    } else {
        for (int tries = 0;; ++tries) {
            IdFile g;
            FILE* fp = std::fopen(id_path.c_str(), "rb");
            if (fp) {
                const size_t n = std::fread(&g, sizeof(g), 1, fp); std::fclose(fp);
                if (n == 1 && g.magic == ID_MAGIC && g.nonce == f.nonce) { f = g; break; }
            }
            if (tries > 6000) return fail("ZSlabComm: no RCCL id for this run at " + id_path + " after 60 s");
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    ncclComm_t c;
    if (ncclCommInitRank(&c, world, f.id, rank) != ncclSuccess) return fail("ncclCommInitRank");
    comm_ = c;
    token_.create(1);
    if (rank == 0) (void)std::remove(id_path.c_str());                       // every rank is in: the file has done its job
    return true;
}

// ---- HOST_STAGED: a file-backed shared segment {header, one slot per rank}.  Rank 0 creates it (a stale one removed first) and
// publishes it by rename, with the run's nonce in the header; the others map it once magic and nonce match.  The barrier is a
// sense-reversing counter in the header (lock-free atomics on MAP_SHARED memory are process-shared on this platform).
// owner_pid: the process that made the segment (rank 0 of ITS run).  A segment whose maker is gone is a crashed run's, whatever its
// barrier counters say (a stale header with arrive == world - 1 let a new rank's first barrier pass at once, before this run's
// rank 0 had poisoned it; the nonce, 0 by default, does not separate runs).
namespace { struct SegHeader { unsigned long long magic, nonce; unsigned long long slot_bytes; std::atomic<unsigned> arrive, gen, attached; long long owner_pid; char pad[16]; }; const unsigned long long SEG_MAGIC = 0x44465a53484d3032ull; }
static bool df_pid_alive(long long pid) { return pid > 0 && (::kill((pid_t)pid, 0) == 0 || errno == EPERM); }

bool ZSlabComm::initHost(const std::string& id_path)
{
    const char* ne = std::getenv("DFUSION_ZSLAB_NONCE");
    const unsigned long long nonce = ne ? std::strtoull(ne, nullptr, 10) : 0ull;
    const char* sm = std::getenv("DFUSION_ZSLAB_HOST_SLOT_MB");
    const size_t slot = (size_t)(sm ? std::max(1, std::atoi(sm)) : 64) << 20;            // sparse: only touched pages exist
    const std::string path = id_path + ".shm";
    const size_t bytes = sizeof(SegHeader) + (size_t)world_ * slot;
    int fd = -1;
    if (rank_ == 0) {
        // A segment left behind by a crashed run with the same nonce (0 by default) would pass the other ranks' header test: POISON it before
        // removing the name -- a rank that has already mapped it sees its magic go and attaches again -- then publish ours.
        { const int ofd = ::open(path.c_str(), O_RDWR); if (ofd >= 0) { const unsigned long long zero = 0; (void)!pwrite(ofd, &zero, sizeof(zero), 0); ::close(ofd); } }
        (void)std::remove(path.c_str());
        const std::string tmp = path + ".tmp." + std::to_string((long long)getpid());
        fd = ::open(tmp.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { if (fd >= 0) ::close(fd); return fail("ZSlabComm(host): cannot create " + tmp); }
        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (m == MAP_FAILED) return fail("ZSlabComm(host): mmap");
        SegHeader* h = new (m) SegHeader();
        h->magic = SEG_MAGIC; h->nonce = nonce; h->slot_bytes = slot; h->arrive = 0; h->gen = 0; h->attached = 1; h->owner_pid = (long long)getpid();
        seg_ = m; seg_bytes_ = bytes; slot_bytes_ = slot;
        if (std::rename(tmp.c_str(), path.c_str()) != 0) return fail("ZSlabComm(host): cannot publish " + path);
    } else {
        for (int attempt = 0;; ++attempt) {
            for (int tries = 0;; ++tries) {
                fd = ::open(path.c_str(), O_RDWR);
                if (fd >= 0) {
                    struct stat sb;
                    if (fstat(fd, &sb) == 0 && (size_t)sb.st_size == bytes) {
                        void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                        ::close(fd);
                        if (m != MAP_FAILED) {
                            SegHeader* h = (SegHeader*)m;
                            if (h->magic == SEG_MAGIC && h->nonce == nonce && h->slot_bytes == slot && df_pid_alive(h->owner_pid)) { seg_ = m; seg_bytes_ = bytes; slot_bytes_ = slot; h->attached.fetch_add(1); break; }
                            munmap(m, bytes);
                        }
                    } else ::close(fd);
                }
                if (tries > 6000) return fail("ZSlabComm(host): no segment for this run at " + path + " after 60 s");
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
            // the first barrier doubles as the check that this is THIS run's segment: rank 0 poisons a stale one's magic before it publishes
            const int b = hostBarrierImpl(true);
            // (passed: once more -- a barrier a stale header completed at once returns before the poison lands; this run's rank 0 is alive
            // and never poisons its own segment)
            if (b == 1 && *(volatile unsigned long long*)&((SegHeader*)seg_)->magic == SEG_MAGIC && df_pid_alive(((SegHeader*)seg_)->owner_pid)) break;
            if (b == 1) { munmap(seg_, seg_bytes_); seg_ = nullptr; if (attempt > 100) return fail("ZSlabComm(host): only stale segments at " + path); continue; }
            if (b == 0) return false;
            munmap(seg_, seg_bytes_); seg_ = nullptr;                          // stale segment: attach again
            if (attempt > 100) return fail("ZSlabComm(host): only stale segments at " + path);
        }
        return true;
    }
    if (!hostBarrier()) return false;
    if (rank_ == 0) (void)std::remove(path.c_str());                          // everybody has it mapped: the name has done its job
    return true;
}

char* ZSlabComm::hostSlot(int rank) const { return (char*)seg_ + sizeof(SegHeader) + (size_t)rank * slot_bytes_; }

bool ZSlabComm::hostBarrier() { return hostBarrierImpl(false) == 1; }
// 1 = passed, 0 = failed (lastError set), -1 = (watch_magic only) the segment's magic was poisoned while waiting: a stale segment
int ZSlabComm::hostBarrierImpl(bool watch_magic)
{
    SegHeader* h = (SegHeader*)seg_;
    const unsigned g = h->gen.load();
    if (h->arrive.fetch_add(1) + 1 == (unsigned)world_) { h->arrive.store(0); h->gen.store(g + 1); return 1; }
    for (long spins = 0; h->gen.load() == g; ++spins) {
        if (watch_magic && *(volatile unsigned long long*)&h->magic != SEG_MAGIC) return -1;
        if (spins > 1200000) { fail("ZSlabComm(host): barrier timed out after 120 s (a rank is gone)"); return 0; }
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(100)); else std::this_thread::yield();
    }
    return 1;
}

ZSlabComm::~ZSlabComm()
{
    if (comm_) { (void)hipDeviceSynchronize(); (void)ncclCommDestroy((ncclComm_t)comm_); }
    if (seg_) munmap(seg_, seg_bytes_);
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

// dynamicfusion_amd/sharded.py slab_bounds_minmax, step for step (the same 48 bisections on the same doubles): boundaries that minimise
// the LARGEST rank's cost -- its own planes plus the halo planes it integrates itself
std::vector<int> ZSlabComm::slabBoundsMinMax(int Z, int world, int halo, const std::vector<double>& weights, bool count_halo)
{
    if ((int)weights.size() != Z || world < 1) return slabBounds(Z, world, halo, weights);
    const int step = (Z % 8 == 0) ? 8 : 1;
    const int min_planes = (std::max(halo, step) + step - 1) / step * step;
    const int H = count_halo ? halo : 0;
    std::vector<double> cum((size_t)Z + 1, 0.0);
    for (int z = 0; z < Z; ++z) cum[z + 1] = cum[z] + weights[z];
    auto cost = [&](int z0, int z1) { return cum[std::min(Z, z1 + H)] - cum[std::max(0, z0 - H)]; };
    auto pack = [&](double T, std::vector<int>& b) -> bool {
        b.assign(1, 0);
        for (int r = 0; r < world - 1; ++r) {
            const int z0 = b.back(), hi = Z - (world - 1 - r) * min_planes;      // every later rank keeps its minimum
            int z1 = z0 + min_planes;
            if (z1 > hi) return false;
            while (z1 + step <= hi && cost(z0, z1 + step) <= T) z1 += step;
            b.push_back(z1);
        }
        b.push_back(Z);
        for (int r = 0; r < world; ++r) if (!(cost(b[r], b[r + 1]) <= T)) return false;
        return true;
    };
    double lo = 0.0, hi = cum[Z] + 1.0;
    std::vector<int> best, b;
    if (!pack(hi, best)) return slabBounds(Z, world, halo, weights);            // (the minimum slab thickness alone does not fit: equal shares)
    for (int it = 0; it < 48; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (pack(mid, b)) { hi = mid; best = b; } else lo = mid;
    }
    return best;
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
    if (!on() || !bytes) return true;
    if (backend_ == HOST_STAGED) {
        if (bytes > slot_bytes_) return fail("ZSlabComm(host): broadcast larger than a slot (DFUSION_ZSLAB_HOST_SLOT_MB)");
        if (rank_ == root) ZS_HIP(hipMemcpy(hostSlot(root), device_ptr, bytes, hipMemcpyDeviceToHost));
        if (!hostBarrier()) return false;
        if (rank_ != root) ZS_HIP(hipMemcpy(device_ptr, hostSlot(root), bytes, hipMemcpyHostToDevice));
        return hostBarrier();
    }
    if (bcast_direct_) {
        // round 6: N - 1 point-to-point copies in one group, each on its own xGMI link -- one step instead of a ring's N - 1
        ncclComm_t c = (ncclComm_t)comm_; hipStream_t st = (hipStream_t)stream_;
        ZS_NCCL(ncclGroupStart());
        bool sent = true;
        if (rank_ == root) { for (int r = 0; r < world_ && sent; ++r) if (r != root) sent = ncclSend(device_ptr, bytes, ncclUint8, r, c, st) == ncclSuccess; }   // (one rank: an empty group)
        else sent = ncclRecv(device_ptr, bytes, ncclUint8, root, c, st) == ncclSuccess;
        const bool closed = ncclGroupEnd() == ncclSuccess;
        if (!sent || !closed) return fail("ZSlabComm: ncclSend / ncclRecv of the frame inputs");
        return true;
    }
    ZS_NCCL(ncclBroadcast(device_ptr, device_ptr, bytes, ncclUint8, root, (ncclComm_t)comm_, (hipStream_t)stream_));
    return true;
}

// The per-pixel MIN of the keys over the ranks -- the first event along every ray, its owner and its Ts (dfusion.h) -- in keys64_[0 .. px).
// KEYS_RING: one ncclAllReduce(MIN).  KEYS_DIRECT (round 6): all-to-all of the keys' row bands, per-key minimum of the N pieces
// (dfusion_raycast_min_pieces), all-gather of the merged bands: two exchanges of one step each over the pairwise links.  keys64_ is a view
// of keys_pad_ (keyImage): whole row bands for every rank.
// keys64_ = a view of the first cols * rows keys of keys_pad_, which holds whole row bands for every rank (world * per * cols keys; the keys
// past the image are all-ones: larger than every key, never read as pixels)
bool ZSlabComm::keyImage(int cols, int rows)
{
    const size_t px = (size_t)cols * rows, pad = (size_t)bandRowsPerRank(rows) * cols * (size_t)world_;
    keys_pad_.create(pad);
    if (pad > px) ZS_HIP(hipMemsetAsync(keys_pad_.ptr() + px, 0xff, (pad - px) * 8, (hipStream_t)stream_));
    keys64_ = DeviceArray<unsigned long long>(keys_pad_.ptr(), px);
    return true;
}

bool ZSlabComm::mergeKeys(int cols, int rows)
{
    if (!on()) return true;
    ncclComm_t c = (ncclComm_t)comm_; hipStream_t st = (hipStream_t)stream_;
    const size_t px = (size_t)cols * rows;
    if (backend_ == HOST_STAGED) {
        if (px * 8 > slot_bytes_) return fail("ZSlabComm(host): image larger than a slot (DFUSION_ZSLAB_HOST_SLOT_MB)");
        ZS_HIP(hipMemcpy(hostSlot(rank_), keys64_.ptr(), px * 8, hipMemcpyDeviceToHost));
        if (!hostBarrier()) return false;
        std::vector<long long> m((const long long*)hostSlot(0), (const long long*)hostSlot(0) + px);
        for (int r = 1; r < world_; ++r) { const long long* o = (const long long*)hostSlot(r); for (size_t i = 0; i < px; ++i) m[i] = std::min(m[i], o[i]); }
        if (!hostBarrier()) return false;                           // (everybody has read every slot before anyone reuses its own)
        ZS_HIP(hipMemcpy(keys64_.ptr(), m.data(), px * 8, hipMemcpyHostToDevice));
        return true;
    }
    if (key_merge_ == KEYS_RING) { ZS_NCCL(ncclAllReduce(keys64_.ptr(), keys64_.ptr(), px, ncclInt64, ncclMin, c, st)); return true; }
    const size_t band_k = (size_t)bandRowsPerRank(rows) * cols;     // keys per band
    key_pieces_.create(band_k * (size_t)world_ + band_k);           // the world pieces of my band, then the merged band
    ZS_NCCL(ncclGroupStart());
    bool sent = true;
    for (int r = 0; r < world_ && sent; ++r)
        sent = ncclSend(keys_pad_.ptr() + (size_t)r * band_k, band_k, ncclInt64, r, c, st) == ncclSuccess &&
               ncclRecv(key_pieces_.ptr() + (size_t)r * band_k, band_k, ncclInt64, r, c, st) == ncclSuccess;
    const bool closed = ncclGroupEnd() == ncclSuccess;
    if (!sent || !closed) return fail("ZSlabComm: ncclSend / ncclRecv of the key bands");
    unsigned long long* merged = key_pieces_.ptr() + band_k * (size_t)world_;
    const int rc = dfusion_raycast_min_pieces(key_pieces_.ptr(), world_, (unsigned long long)band_k, merged, stream_);
    if (rc != 0) return fail(std::string("dfusion_raycast_min_pieces: ") + dfusion_error_string(rc));
    ZS_NCCL(ncclAllGather(merged, keys_pad_.ptr(), band_k, ncclInt64, c, st));
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
    if (backend_ == HOST_STAGED) {
        // my first `halo` own planes at the front of my slot, my last ones behind them; then each neighbour's facing half comes in
        const size_t hb = (size_t)halo * plane * sizeof(int);
        if (2 * hb > slot_bytes_) return fail("ZSlabComm(host): halo planes larger than a slot (DFUSION_ZSLAB_HOST_SLOT_MB)");
        ZS_HIP(hipMemcpy(hostSlot(rank_), base + (size_t)lo_local * plane, hb, hipMemcpyDeviceToHost));
        ZS_HIP(hipMemcpy(hostSlot(rank_) + hb, base + (size_t)(hi_local - halo) * plane, hb, hipMemcpyDeviceToHost));
        if (!hostBarrier()) return false;
        if (rank_ > 0) ZS_HIP(hipMemcpy(base, hostSlot(rank_ - 1) + hb, hb, hipMemcpyHostToDevice));                       // its last own planes
        if (rank_ < world_ - 1) ZS_HIP(hipMemcpy(base + (size_t)hi_local * plane, hostSlot(rank_ + 1), hb, hipMemcpyHostToDevice));   // its first ones
        return hostBarrier();
    }
    ncclComm_t c = (ncclComm_t)comm_; hipStream_t st = (hipStream_t)stream_;
    // (a failed send / receive must not leave the group open: close it, then report the first error)
    ncclResult_t first = ncclGroupStart();
    auto acc = [&](ncclResult_t r) { if (first == ncclSuccess) first = r; };
    if (first == ncclSuccess) {
        if (rank_ > 0) {                                            // lower neighbour: my first own planes out, its last ones in
            acc(ncclSend(base + (size_t)lo_local * plane, (size_t)halo * plane, ncclInt32, rank_ - 1, c, st));
            acc(ncclRecv(base, (size_t)n_lo * plane, ncclInt32, rank_ - 1, c, st));
        }
        if (rank_ < world_ - 1) {
            acc(ncclSend(base + (size_t)(hi_local - halo) * plane, (size_t)halo * plane, ncclInt32, rank_ + 1, c, st));
            acc(ncclRecv(base + (size_t)hi_local * plane, (size_t)n_hi * plane, ncclInt32, rank_ + 1, c, st));
        }
        const ncclResult_t end = ncclGroupEnd();
        acc(end);
    }
    if (first != ncclSuccess) return fail(std::string("RCCL: ") + ncclGetErrorString(first) + " in exchangeHalos");
    return true;
}

bool ZSlabComm::raycast(TsdfVolume& slab, const Affine3f& camera_pose, const Intr& intr, int cols, int rows, Cloud& points, Normals& normals, int dst)
{
    if (!ok_) return false;
    ncclComm_t c = (ncclComm_t)comm_; hipStream_t st = (hipStream_t)stream_;
    const size_t px = (size_t)cols * rows;
    if (!keyImage(cols, rows)) return false;
    slab.raycastMarch(camera_pose, intr, cols, rows, (unsigned)rank_, keys64_);
    if (!mergeKeys(cols, rows)) return false;
    out_.create(2 * px);                                            // normals (what the reduce sums), then the points
    normals = Normals(rows, cols, out_.ptr(), (size_t)cols * sizeof(Normal));
    points = Cloud(rows, cols, out_.ptr() + px, (size_t)cols * sizeof(Point));
    slab.raycastShadeNormals(camera_pose, intr, keys64_, normals);
    // only the normals cross GPUs (4.9 MB at 640 x 480; every summand but one is integer zero): the points follow from the merged keys
    if (on() && backend_ == HOST_STAGED) {
        ZS_HIP(hipMemcpy(hostSlot(rank_), out_.ptr(), px * 16, hipMemcpyDeviceToHost));
        if (!hostBarrier()) return false;
        if (rank_ == dst) {
            std::vector<int> sum((const int*)hostSlot(0), (const int*)hostSlot(0) + px * 4);
            for (int r = 1; r < world_; ++r) { const int* o = (const int*)hostSlot(r); for (size_t i = 0; i < px * 4; ++i) sum[i] = (int)((unsigned)sum[i] + (unsigned)o[i]); }
            ZS_HIP(hipMemcpy(out_.ptr(), sum.data(), px * 16, hipMemcpyHostToDevice));
        }
        if (!hostBarrier()) return false;
    } else if (on()) ZS_NCCL(ncclReduce(out_.ptr(), out_.ptr(), px * 4, ncclInt32, ncclSum, dst, c, st));
    if (rank_ == dst) slab.raycastPointsOfKeys(camera_pose, intr, keys64_, normals, points);
    return true;
}

bool ZSlabComm::raycastRowBands(TsdfVolume& slab, const Affine3f& camera_pose, const Intr& intr, int cols, int rows, Cloud& points, Normals& normals)
{
    if (!ok_) return false;
    ncclComm_t c = (ncclComm_t)comm_; hipStream_t st = (hipStream_t)stream_;
    const size_t px = (size_t)cols * rows;
    const int per = bandRowsPerRank(rows), row0 = bandRow0(rows), nrows = bandRows(rows);
    const size_t band_px = (size_t)per * cols, pad_px = band_px * (size_t)world_;
    if (!keyImage(cols, rows)) return false;
    slab.raycastMarch(camera_pose, intr, cols, rows, (unsigned)rank_, keys64_);
    if (!mergeKeys(cols, rows)) return false;
    // out_: the padded normals every rank shades into (world * per rows; the rows past the image stay zero), this rank's band of
    // the summed normals, this rank's band of points
    out_.create(pad_px + 2 * band_px);
    Normals shaded(rows, cols, out_.ptr(), (size_t)cols * sizeof(Normal));
    if (pad_px > px) ZS_HIP(hipMemsetAsync(out_.ptr() + px, 0, (pad_px - px) * sizeof(Point), st));
    slab.raycastShadeNormals(camera_pose, intr, keys64_, shaded);
    Point* band_n = out_.ptr() + pad_px; Point* band_p = band_n + band_px;
    if (on() && row_merge_ == ALL_TO_ALL) {
        // the direct form: piece r of pieces_ = rank r's shading of MY band; then the pieces are added on the device
        pieces_.create(pad_px);
        if (backend_ == HOST_STAGED) {
            ZS_HIP(hipMemcpy(hostSlot(rank_), out_.ptr(), pad_px * 16, hipMemcpyDeviceToHost));
            if (!hostBarrier()) return false;
            for (int r = 0; r < world_; ++r)
                ZS_HIP(hipMemcpy(pieces_.ptr() + (size_t)r * band_px, hostSlot(r) + (size_t)rank_ * band_px * 16, band_px * 16, hipMemcpyHostToDevice));
            if (!hostBarrier()) return false;
        } else {
            ZS_NCCL(ncclGroupStart());
            bool sent = true;
            for (int r = 0; r < world_ && sent; ++r) {
                sent = ncclSend(out_.ptr() + (size_t)r * band_px, band_px * 4, ncclInt32, r, c, st) == ncclSuccess &&
                       ncclRecv(pieces_.ptr() + (size_t)r * band_px, band_px * 4, ncclInt32, r, c, st) == ncclSuccess;
            }
            const bool closed = ncclGroupEnd() == ncclSuccess;               // (a failed send / receive must not leave the group open)
            if (!sent || !closed) return fail("ZSlabComm: ncclSend / ncclRecv of the row-band pieces");
        }
        const int rc = dfusion_raycast_sum_pieces((const uint32_t*)pieces_.ptr(), world_, (unsigned long long)band_px * 4, (uint32_t*)band_n, stream_);
        if (rc != 0) return fail(std::string("dfusion_raycast_sum_pieces: ") + dfusion_error_string(rc));
    } else if (on() && backend_ == HOST_STAGED) {
        ZS_HIP(hipMemcpy(hostSlot(rank_), out_.ptr(), pad_px * 16, hipMemcpyDeviceToHost));
        if (!hostBarrier()) return false;
        std::vector<int> sum(band_px * 4, 0);
        for (int r = 0; r < world_; ++r) { const int* o = (const int*)hostSlot(r) + (size_t)rank_ * band_px * 4; for (size_t i = 0; i < band_px * 4; ++i) sum[i] = (int)((unsigned)sum[i] + (unsigned)o[i]); }
        if (!hostBarrier()) return false;
        ZS_HIP(hipMemcpy(band_n, sum.data(), band_px * 16, hipMemcpyHostToDevice));
    } else if (on()) {
        ZS_NCCL(ncclReduceScatter(out_.ptr(), band_n, band_px * 4, ncclInt32, ncclSum, c, st));
    } else {
        ZS_HIP(hipMemcpyAsync(band_n, out_.ptr(), band_px * 16, hipMemcpyDeviceToDevice, st));
    }
    normals = Normals(nrows, cols, band_n, (size_t)cols * sizeof(Normal));
    points = Cloud(nrows, cols, band_p, (size_t)cols * sizeof(Point));
    if (nrows > 0) slab.raycastPointsOfKeysRows(camera_pose, intr, keys64_, rows, row0, normals, points);
    return true;
}

bool ZSlabComm::barrier()
{
    if (!ok_) return false;
    if (backend_ == HOST_STAGED) { ZS_HIP(hipDeviceSynchronize()); return world_ > 1 ? hostBarrier() : true; }
    if (on()) ZS_NCCL(ncclAllReduce(token_.ptr(), token_.ptr(), 1, ncclInt32, ncclSum, (ncclComm_t)comm_, (hipStream_t)stream_));
    ZS_HIP(hipDeviceSynchronize());
    return true;
}
