// kfusion/warp_field.hpp -- WarpField with the reference's hot-path interface
// (/root/reference/kfusion/include/kfusion/warp_field.hpp:41-88): host node store + GPU k-NN / DQB / warp through the
// C-ABI.  Solver-side members (energy*, Ceres, getNodesAsMat) are out of scope.
#pragma once
#include <vector>
#include <kfusion/types.hpp>
#include <kfusion/utils/dual_quaternion.hpp>

#define KNN_NEIGHBOURS 8     // warp_field.hpp:10; a runtime parameter (k) here

struct DfWarpField;

namespace kfusion
{
    namespace cuda { class TsdfVolume; }

    struct deformation_node                                 // warp_field.hpp:35-40
    {
        Vec3f vertex;
        utils::DualQuaternion<float> transform;
        float weight = 0;
    };

    class WarpField
    {
    public:
        explicit WarpField(int k = KNN_NEIGHBOURS);
        ~WarpField();
        WarpField(const WarpField&) = delete;
        WarpField& operator=(const WarpField&) = delete;

        /// warp_field.cpp:68-88: one identity node per point, dg_w = 3 (NaN points skipped -- the reference
        /// leaves zero-position, zero-weight nodes in their place; fixed)
        void init(const std::vector<Vec3f>& first_frame);
        const std::vector<deformation_node>* getNodes() const { return &nodes_; }
        std::vector<deformation_node>* getNodes() { return &nodes_; }
        /// push host-side node edits (positions or transforms) to the device; `positions_changed` invalidates the k-NN index
        void commit(bool positions_changed);

        /// warp_field.cpp:180-195, on the GPU; vectors are modified in place
        void warp(std::vector<Vec3f>& points, std::vector<Vec3f>& normals) const;
        /// the same on device-resident packed float3 arrays (n points; normals may be empty)
        void warp(cuda::DeviceArray<float>& points, cuda::DeviceArray<float>& normals, int n) const;
        /// warp_field.cpp:247-251; results via getRetIndex / getDistSquared like the reference's globals
        void KNN(Vec3f point) const;
        std::vector<float>* getDistSquared() const { return &out_dist_sqr_; }
        std::vector<size_t>* getRetIndex() const { return &ret_index_; }
        void setWarpToLive(const Affine3f& pose) { warp_to_live_ = pose; }
        const Affine3f& getWarpToLive() const { return warp_to_live_; }
        void buildKDTree() { commit(true); }                 // warp_field.cpp:275-282

        int k() const { return k_; }
        DfWarpField* handle() const { return handle_; }
        /// exact k-NN index (+ per-voxel tables) for one volume; rebuilt only when positions / geometry change
        void ensureIndex(const cuda::TsdfVolume& volume) const;
    private:
        std::vector<deformation_node> nodes_;
        Affine3f warp_to_live_;
        int k_;
        DfWarpField* handle_;
        mutable std::vector<float> out_dist_sqr_;
        mutable std::vector<size_t> ret_index_;
        mutable bool index_ok_;
        mutable const void* index_volume_;
    };
}
