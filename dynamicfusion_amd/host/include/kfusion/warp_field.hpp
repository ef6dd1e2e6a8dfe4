// kfusion/warp_field.hpp -- WarpField with the reference's hot-path interface
// (/root/reference/kfusion/include/kfusion/warp_field.hpp:41-88): host node store + GPU k-NN / DQB / warp through the
// C-ABI.  energy_data is the GPU data-term solve; energy / energy_reg / clear exist with the reference's (empty) behaviour -- no
// regularisation term is ever added to the reference's problem either.  tests/test_mirror_headers.py checks every public name of the
// reference's header against this one.
#pragma once
#include <utility>
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
#ifdef KFUSION_USE_OPENCV
        /// warp_field.cpp:41-63: every 50th point of every 50th row of a cloud image (KinFu passes the 1 x N extracted cloud,
        /// kinfu.cpp:252: every 50th point); only the sampled, non-NaN points become nodes
        void init(const cv::Mat& first_frame);
#endif
        /// warp_field.cpp:98-108: checks that the two images have the same size, nothing else (as in the reference)
        void energy(const cuda::Cloud& frame, const cuda::Normals& normals, const Affine3f& pose, const cuda::TsdfVolume& tsdfVolume,
                    const std::vector<std::pair<utils::DualQuaternion<float>, utils::DualQuaternion<float>>>& edges);
        /// warp_field.cpp:168-172: empty in the reference, empty here
        void energy_reg(const std::vector<std::pair<utils::DualQuaternion<float>, utils::DualQuaternion<float>>>& edges);
        /// warp_field.cpp:203-217: the dual-quaternion blend at one vertex (k-NN on the GPU, sums on the host in the reference's order)
        utils::DualQuaternion<float> DQB(const Vec3f& vertex) const;
        /// warp_field.cpp:225-230: KNN(vertex), then weights[i] = weighting(dist_i^2, dg_w of neighbour i)
        void getWeightsAndUpdateKNN(const Vec3f& vertex, float weights[KNN_NEIGHBOURS]) const;
        /// warp_field.cpp:238-241: exp(-d^2 / (2 dg_w^2)), the exponential in double as the reference's overload resolves it
        float weighting(float squared_dist, float weight) const;
        /// warp_field.cpp:298-301: empty in the reference, empty here
        void clear();
        /// the host node store; after a GPU solve (energy_data) the transforms are fetched from the device on the first access
        const std::vector<deformation_node>* getNodes() const { pullNodes(); return &nodes_; }
        std::vector<deformation_node>* getNodes() { pullNodes(); return &nodes_; }
        /// number of nodes, without touching the transforms (no device round trip)
        size_t nodeCount() const { return nodes_.size(); }
        /// push host-side node edits (positions or transforms) to the device; `positions_changed` invalidates the k-NN index
        void commit(bool positions_changed);
        /// node transforms that are ALREADY on the device (8 floats a node: rotation_, translation_ -- what a device-side solver leaves):
        /// no host round trip, nothing synchronises; the host node store follows on its next access (getNodes)
        void setTransformsDevice(const cuda::DeviceArray<float>& dq8);

        /// warp_field.cpp:180-195, on the GPU; vectors are modified in place
        void warp(std::vector<Vec3f>& points, std::vector<Vec3f>& normals) const;
        /// the same on device-resident packed float3 arrays (n points; normals may be empty)
        void warp(cuda::DeviceArray<float>& points, cuda::DeviceArray<float>& normals, int n) const;
        /// warp_field.cpp:117-163 (Ceres) == WarpFieldOptimiser::optimiseWarpData (Opt): least-squares update of the node
        /// translations from the data term, solved on the GPU (dfusion_warp_solve_data_term); the nodes are updated like
        /// WarpProblem::updateWarp / copyResultToCPUFromFloat3 do.  The normals are unused, as in the reference's energy.
        void energy_data(const std::vector<Vec3f>& canonical_vertices, const std::vector<Vec3f>& canonical_normals,
                         const std::vector<Vec3f>& live_vertices, const std::vector<Vec3f>& live_normals);
        void energy_data(const cuda::DeviceArray<float>& canonical_vertices, const cuda::DeviceArray<float>& live_vertices, int n);
        /// conjugate-gradient steps of energy_data (Opt's linearIter = 100, kinfu.cpp:118) and its damping
        void setSolverIterations(int iters) { solver_iters_ = iters; }
        int getSolverIterations() const { return solver_iters_; }
        void setSolverDamping(float lambda) { solver_lambda_ = lambda; }
        /// E before / after the last energy_data (evaluated only when asked for: two extra passes over the points)
        void setTrackEnergy(bool on) { track_energy_ = on; }
        float lastEnergyBefore() const { return last_energy_[0]; }
        float lastEnergyAfter() const { return last_energy_[1]; }
        /// warp_field.cpp:247-251; results via getRetIndex / getDistSquared like the reference's globals
        void KNN(Vec3f point) const;
        std::vector<float>* getDistSquared() const { return &out_dist_sqr_; }
        std::vector<size_t>* getRetIndex() const { return &ret_index_; }
        void setWarpToLive(const Affine3f& pose) { warp_to_live_ = pose; }
        const Affine3f& getWarpToLive() const { return warp_to_live_; }
        void buildKDTree() { commit(true); }                 // warp_field.cpp:275-282
        /// warp_field.cpp:284-293: every node's position moved by its own translation, one Vec3f per node -- what the demo shows as
        /// the "warp_field" cloud (apps/demo.cpp:67).  The reference returns a 1 x N CV_32FC3 cv::Mat: the same N x 3 floats.
#ifdef KFUSION_USE_OPENCV
        typedef cv::Mat NodesMat;                                          // 1 x N, CV_32FC3, as in the reference
#else
        typedef std::vector<Vec3f> NodesMat;
#endif
        const NodesMat getNodesAsMat() const;
        /// the same N x 3 floats whatever the build
        std::vector<Vec3f> getNodesAsVector() const;

        int k() const { return k_; }
        DfWarpField* handle() const { return handle_; }
        /// exact k-NN index (+ per-voxel tables) for one volume; rebuilt only when positions / geometry change
        /// `tables` = also the per-voxel k-NN + weight caches the warped integrate streams (6 GiB at 512^3); without them only the
        /// brick candidate lists are built, which is all point queries (KNN, warp, energy_data) need
        void ensureIndex(const cuda::TsdfVolume& volume, bool tables = true) const;
        /// What the last warped integrate of `volume` found alive: 8 x 8 x 8 blocks kept by its verdict pass, per 8-plane layer of the
        /// GLOBAL volume (dims.z / 8 entries; zero outside the slab's own planes).  Summed over the ranks it is the measured profile of
        /// the sweep's work along z -- the weights cuda::ZSlabComm::slabBounds wants for a re-balance after the first frames
        /// (one weight per plane: repeat every entry 8 times).  Empty when no warped integrate with a launch plan has run yet.
        std::vector<unsigned long long> aliveBlocksPerLayer(const cuda::TsdfVolume& volume) const;
    private:
        void pullNodes() const;                              // device transforms -> nodes_ when a solve has made them newer
        mutable std::vector<deformation_node> nodes_;
        mutable cuda::DeviceArray<float> solve_dq_, solve_energy_;   // outputs of the last energy_data (kept: no per-frame allocation)
        mutable bool nodes_stale_ = false;
        Affine3f warp_to_live_;
        int k_;
        DfWarpField* handle_;
        mutable std::vector<float> out_dist_sqr_;
        mutable std::vector<size_t> ret_index_;
        mutable bool index_ok_;
        mutable const void* index_volume_;
        mutable bool index_tables_ = false;
        mutable float index_key_[20] = {0};                  // dims, voxel size, pose, slab of the geometry the index was built for
        int solver_iters_ = 100;
        float solver_lambda_ = 0.f;
        float last_energy_[2] = {0.f, 0.f};
        bool track_energy_ = false;
    };
}
