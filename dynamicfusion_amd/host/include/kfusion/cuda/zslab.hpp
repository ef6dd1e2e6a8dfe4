// kfusion/cuda/zslab.hpp -- Z-slab sharding of kfusion::cuda::TsdfVolume over the GPUs of one node, in C++ over RCCL (xGMI).
//
// No counterpart in the reference (single GPU).  One process per GPU; rank g holds the planes ZSlabComm::slabRange gives it as a
// TsdfVolume::setSlab shard.  What crosses GPUs per frame (BASELINE.json north star):
//   broadcast()      the frame inputs from rank 0 (depth image, node transforms) -- N - 1 point-to-point copies in one group (or one ncclBroadcast)
//   exchangeHalos()  after the integrate: the H boundary planes to / from both Z neighbours, paired ncclSend / ncclRecv in one
//                    group (ring neighbours only: two of the seven xGMI links).  Not needed when every rank integrates its halo
//                    planes itself -- TsdfVolume::setSlab(z0, n, H, /*integrate_halo=*/true): the integrate is a pure function of
//                    the broadcast inputs, the own range stays the non-overlapping one -- the harness offers both
//   raycast()        two stages, because the zero-crossing refinement can move a vertex into ANOTHER rank's slab
//                    (tsdf_volume.cu:389): march on the global step lattice (each rank evaluates only the steps whose sample lies
//                    in a plane it owns) -> the per-pixel MIN of the int64 merge keys [step | hit | rank | Ts bits]
//                    (include/dfusion.h; as direct exchanges -- row bands all-to-all, local minimum, all-gather: setKeyMerge -- or
//                    one ncclAllReduce): first event along every ray, its owner, and its refined ray parameter Ts, from which
//                    every rank recomputes the vertex -> the owner of the vertex' plane computes the normal -> ONE ncclReduce(SUM) of the
//                    int32 view of the NORMALS to rank `dst` (every summand but one is integer zero) -> rank `dst` makes the points
//                    from the merged keys (vertex = origin + direction * Ts; a hit stands iff its normal does): bit-identical with the
//                    unsharded ray-cast.  (Round 2 exchanged the winners' vertices with a second all-reduce, 4.9 MB per frame.)
// The same sequence, collective for collective, as dynamicfusion_amd/sharded.py (torch.distributed), which the world-size-2/3 gloo
// tests and the one-GPU 8-slab emulation exercise; this file is what a C++ host (KinFu) links instead.
#pragma once
#include <algorithm>
#include <string>
#include <vector>
#include <kfusion/types.hpp>
#include <kfusion/cuda/tsdf_volume.hpp>

namespace kfusion { namespace cuda {

class ZSlabComm
{
public:
    /// What carries the collectives.  RCCL: one rank per GPU over xGMI (the product).  HOST_STAGED: every collective staged through a
    /// shared-memory segment on the host (device -> host, a barrier, host -> device) -- slow, but it lets N processes share ONE GPU
    /// (RCCL refuses two ranks on a device), so the C++ collective SEQUENCE runs with N > 1 on a one-GPU box (tests) and on nodes
    /// without a working RCCL.  FROM_ENV: DFUSION_ZSLAB_BACKEND=host selects HOST_STAGED, anything else RCCL.
    enum Backend { FROM_ENV = 0, RCCL = 1, HOST_STAGED = 2 };
    /// Collective over all ranks of the node.  `id_path`: a file every rank can read (e.g. under /tmp): rank 0 publishes the RCCL
    /// unique id there (removing a stale one first, and the file itself once every rank is in), the others wait for it; give every
    /// rank of a run the same DFUSION_ZSLAB_NONCE (e.g. the launcher's pid) and a leftover file of another run is never accepted.
    /// The calling thread's current HIP device is the rank's GPU.  Check ok() afterwards.
    ZSlabComm(int rank, int world, const std::string& id_path, Backend backend = FROM_ENV);
    Backend backend() const { return backend_; }
    ~ZSlabComm();
    ZSlabComm(const ZSlabComm&) = delete;
    ZSlabComm& operator=(const ZSlabComm&) = delete;
    int rank() const { return rank_; }
    int world() const { return world_; }

    /// planes owned by `rank`: contiguous, brick-aligned (multiples of 8 where Z allows)
    static void slabRange(int Z, int rank, int world, int& z_own0, int& z_own_n);
    /// the same from WORK weights (one non-negative number per plane: e.g. the share of the plane inside the frustum and in front of
    /// the first frame's surface -- dynamicfusion_amd/sharded.py frustum_plane_weights): boundaries b[0] = 0 < ... < b[world] = Z,
    /// multiples of 8 where Z allows, every slab at least max(halo, 8) planes, about the same weight per rank.  Equal plane counts
    /// leave the far ranks several times the near ranks' work (a frustum's cross-section grows with the square of the depth).
    static std::vector<int> slabBounds(int Z, int world, int halo, const std::vector<double>& weights);
    /// boundaries that minimise the LARGEST rank's cost instead (round 6; sharded.slab_bounds_minmax, the same bisection): a rank's cost
    /// is the weight of its own planes plus -- count_halo -- of the halo planes it integrates itself (setSlab(..., integrate_halo = true));
    /// interior ranks carry two halos, and at 8 ranks 16 halo planes are a quarter of a slab.  What to cut with after the first frames'
    /// WarpField::aliveBlocksPerLayer, summed over the ranks, has said where the work is.
    static std::vector<int> slabBoundsMinMax(int Z, int world, int halo, const std::vector<double>& weights, bool count_halo = true);
    /// planes of the neighbour a slab must hold for the ray-cast: the march's `next` sample is one time_step beyond `curr`
    /// (tsdf_volume.cu:378-380), trilinear taps read g+1 (:236-243), gradient probes reach +-gradient_delta (:413-423)
    static int haloPlanes(float trunc_dist, float step_factor, float delta_factor, float voxel_z);
    /// false (with a message) if a rank would own no plane or fewer planes than the halo its neighbours need -- the same verdict on
    /// every rank, to be asked BEFORE the first collective
    static bool partitionOk(int Z, int world, int halo, std::string* why = nullptr);

    /// Every collective returns false on an RCCL / HIP error (lastError() says which) instead of ending the process; ok() is false
    /// from then on -- and from the start if the communicator could not be made -- and further calls fail at once.
    bool ok() const { return ok_; }
    const std::string& lastError() const { return error_; }
    bool broadcast(void* device_ptr, size_t bytes, int root = 0);
    bool exchangeHalos(TsdfVolume& slab, int halo);
    /// result on rank `dst`.  points / normals become dense cols x rows VIEWS of a buffer this object owns (valid until the next
    /// raycast); on the other ranks they hold that rank's partial image
    bool raycast(TsdfVolume& slab, const Affine3f& camera_pose, const Intr& intr, int cols, int rows, Cloud& points, Normals& normals, int dst = 0);
    /// The same cast with the second collective as ONE ncclReduceScatter of the normals by PIXEL ROWS (round 4): rank r receives the
    /// summed normals of its band of rows -- bandRow0(rows), bandRows(rows) -- and makes the band's points itself; points / normals
    /// become bandRows x cols views of the band.  1 / world of the image lands on a rank and none of them is a hot spot; the image
    /// stays row-sharded for a row-sharded consumer.
    bool raycastRowBands(TsdfVolume& slab, const Affine3f& camera_pose, const Intr& intr, int cols, int rows, Cloud& points, Normals& normals);
    /// How raycastRowBands moves the normals (round 5).  REDUCE_SCATTER: one ncclReduceScatter (a ring: N - 1 steps).  ALL_TO_ALL: every rank
    /// sends each other rank ITS band of the normals it shaded -- fixed-size pieces, one ncclSend / ncclRecv group, each byte over one xGMI
    /// link once, no count exchange -- and adds the N pieces of its own band (dfusion_raycast_sum_pieces).  Same bits either way.
    /// Default: DFUSION_ZSLAB_MERGE=a2a selects ALL_TO_ALL, anything else REDUCE_SCATTER.
    enum RowMerge { REDUCE_SCATTER = 0, ALL_TO_ALL = 1 };
    void setRowMerge(RowMerge m) { row_merge_ = m; }
    RowMerge rowMerge() const { return row_merge_; }
    /// How the casts merge the keys (round 6).  KEYS_RING: one ncclAllReduce(MIN) -- 2 (N - 1) ring steps.  KEYS_DIRECT (default): every rank
    /// sends each other rank ITS band of the key image (one ncclSend / ncclRecv group), takes the per-key minimum of the N pieces of its own
    /// band (dfusion_raycast_min_pieces) and one ncclAllGather hands every rank the merged image: two exchanges of one step each over the
    /// pairwise xGMI links.  Same bits.  DFUSION_ZSLAB_KEY_MERGE=ring selects the ring.
    enum KeyMerge { KEYS_RING = 0, KEYS_DIRECT = 1 };
    void setKeyMerge(KeyMerge m) { key_merge_ = m; }
    KeyMerge keyMerge() const { return key_merge_; }
    /// broadcast() as N - 1 point-to-point copies in one group, each on its own link (default), or one ncclBroadcast
    /// (DFUSION_ZSLAB_BCAST=ring)
    void setBroadcastDirect(bool on) { bcast_direct_ = on; }
    int bandRowsPerRank(int rows) const { return (rows + world_ - 1) / world_; }
    int bandRow0(int rows) const { return std::min(rows, rank_ * bandRowsPerRank(rows)); }
    int bandRows(int rows) const { return std::max(0, std::min(rows, (rank_ + 1) * bandRowsPerRank(rows)) - bandRow0(rows)); }
    bool barrier();
private:
    bool fail(const std::string& what);
    bool initRccl(const std::string& id_path);
    bool initHost(const std::string& id_path);
    bool hostBarrier();
    int hostBarrierImpl(bool watch_magic);
    char* hostSlot(int rank) const;
    Backend backend_;
    void* seg_; size_t seg_bytes_, slot_bytes_;   // HOST_STAGED: the mapped segment {header, world slots}
    int rank_, world_;
    void* comm_;                 // ncclComm_t
    void* stream_;               // hipStream_t: the null stream (the C++ mirror enqueues everything there)
    bool ok_;
    std::string error_;
    DeviceArray<unsigned long long> keys64_;
    DeviceArray<Point> out_;     // normals (summed over the ranks), then the points (made from the merged keys on rank dst)
    DeviceArray<int> token_;
    RowMerge row_merge_;
    DeviceArray<Point> pieces_;  // ALL_TO_ALL: the world pieces of this rank's band
    KeyMerge key_merge_;
    bool bcast_direct_;
    bool force_ = false;         // DFUSION_ZSLAB_FORCE_COLLECTIVES: the RCCL calls are issued with one rank too
    bool on() const { return world_ > 1 || force_; }
    bool keyImage(int cols, int rows);
    bool mergeKeys(int cols, int rows);
    DeviceArray<unsigned long long> keys_pad_;     // the key image + padding to whole row bands (keys64_ is a view of its first cols * rows keys)
    DeviceArray<unsigned long long> key_pieces_;   // KEYS_DIRECT: the world pieces of this rank's band, then the merged band
};

} }
