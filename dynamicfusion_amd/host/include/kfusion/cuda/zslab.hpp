// kfusion/cuda/zslab.hpp -- Z-slab sharding of kfusion::cuda::TsdfVolume over the GPUs of one node, in C++ over RCCL (xGMI).
//
// No counterpart in the reference (single GPU).  One process per GPU; rank g holds the planes ZSlabComm::slabRange gives it as a
// TsdfVolume::setSlab shard.  What crosses GPUs per frame (BASELINE.json north star):
//   broadcast()      the frame inputs from rank 0 (depth image, node transforms) -- one ncclBroadcast of bytes each
//   exchangeHalos()  after the integrate: the H boundary planes to / from both Z neighbours, paired ncclSend / ncclRecv in one
//                    group (ring neighbours only: two of the seven xGMI links).  Not needed when every rank integrates its halo
//                    planes itself (TsdfVolume::setSlab(z0 - H, n + 2H, 0) as the integrate view; the integrate is a pure function
//                    of the broadcast inputs) -- the harness offers both
//   raycast()        two stages, because the zero-crossing refinement can move a vertex into ANOTHER rank's slab
//                    (tsdf_volume.cu:389): march on the global step lattice (each rank evaluates only the steps whose sample lies
//                    in a plane it owns) -> ncclAllReduce(MIN) of (event key << 8 | rank) as int64 -> winners' vertices by
//                    ncclAllReduce(SUM) on their int32 view (every summand but one is integer zero) -> the owner of the vertex'
//                    plane shades -> ncclReduce(SUM) of the int32 views to rank `dst`: bit-identical with the unsharded ray-cast.
// The same sequence, collective for collective, as dynamicfusion_amd/sharded.py (torch.distributed), which the world-size-2/3 gloo
// tests and the one-GPU 8-slab emulation exercise; this file is what a C++ host (KinFu) links instead.
#pragma once
#include <string>
#include <kfusion/types.hpp>
#include <kfusion/cuda/tsdf_volume.hpp>

namespace kfusion { namespace cuda {

class ZSlabComm
{
public:
    /// Collective over all ranks of the node.  `id_path`: a file every rank can read (e.g. under /tmp): rank 0 publishes the RCCL
    /// unique id there, the others wait for it.  The calling thread's current HIP device is the rank's GPU.
    ZSlabComm(int rank, int world, const std::string& id_path);
    ~ZSlabComm();
    ZSlabComm(const ZSlabComm&) = delete;
    ZSlabComm& operator=(const ZSlabComm&) = delete;
    int rank() const { return rank_; }
    int world() const { return world_; }

    /// planes owned by `rank`: contiguous, brick-aligned (multiples of 8 where Z allows)
    static void slabRange(int Z, int rank, int world, int& z_own0, int& z_own_n);
    /// planes of the neighbour a slab must hold for the ray-cast: the march's `next` sample is one time_step beyond `curr`
    /// (tsdf_volume.cu:378-380), trilinear taps read g+1 (:236-243), gradient probes reach +-gradient_delta (:413-423)
    static int haloPlanes(float trunc_dist, float step_factor, float delta_factor, float voxel_z);
    /// false (with a message) if a rank would own no plane or fewer planes than the halo its neighbours need -- the same verdict on
    /// every rank, to be asked BEFORE the first collective
    static bool partitionOk(int Z, int world, int halo, std::string* why = nullptr);

    void broadcast(void* device_ptr, size_t bytes, int root = 0);
    void exchangeHalos(TsdfVolume& slab, int halo);
    /// result on rank `dst`.  points / normals become dense cols x rows VIEWS of buffers this object owns (valid until the next
    /// raycast); on the other ranks they hold that rank's partial image
    void raycast(TsdfVolume& slab, const Affine3f& camera_pose, const Intr& intr, int cols, int rows, Cloud& points, Normals& normals, int dst = 0);
    void barrier();
private:
    int rank_, world_;
    void* comm_;                 // ncclComm_t
    void* stream_;               // hipStream_t: the null stream (the C++ mirror enqueues everything there)
    DeviceArray<unsigned long long> keys64_;
    DeviceArray<Point> vertex_, points_, normals_;
    DeviceArray<int> token_;
};

} }
