// kfusion/kinfu.hpp -- kfusion::KinFuParams / kfusion::KinFu with the reference's interface
// (/root/reference/kfusion/include/kfusion/kinfu.hpp:15-112) minus viz (renderImage) and the Opt/Ceres solver.
// Extensions: max_warp_nodes (node ids are 16-bit here; the seed cloud's stride-50 sampling, warp_field.cpp:49-60, widens if needed),
// warped_fusion (call the per-voxel warped integrate instead of surface_fusion; default off = the reference's behaviour) and
// device_resident (default on: same results as the reference's host staging, bit for bit, without the ~50 MB/frame of PCIe traffic;
// switch off when optimiseWarp is overridden, which needs the host vectors).
#pragma once
#include <memory>
#include <vector>
#include <kfusion/types.hpp>
#include <kfusion/cuda/projective_icp.hpp>
#include <kfusion/cuda/tsdf_volume.hpp>
#include <kfusion/warp_field.hpp>

namespace kfusion
{
    struct KinFuParams
    {
        static KinFuParams default_params();                   // kinfu.cpp:55-89
        static KinFuParams default_params_dynamicfusion();     // kinfu.cpp:15-50

        int cols, rows;
        Intr intr;
        Vec3i volume_dims;
        Vec3f volume_size;
        Affine3f volume_pose;
        float bilateral_sigma_depth, bilateral_sigma_spatial;
        int bilateral_kernel_size;
        float icp_truncate_depth_dist, icp_dist_thres, icp_angle_thres;
        std::vector<int> icp_iter_num;
        float tsdf_min_camera_movement, tsdf_trunc_dist;
        int tsdf_max_weight;
        float raycast_step_factor, gradient_delta_factor;
        Vec3f light_pose;
        // extensions
        int max_warp_nodes = 65535;
        bool warped_fusion = false;
        bool use_depth_pyramids = false; // the reference's USE_DEPTH build (internal.hpp:6, commented out there): depth + normals pyramids, masked depth ICP
        bool device_resident = true;   // keep dynamicfusion()'s point sets on the GPU (no host staging); false = the reference's data flow
        int warp_solver_iterations = 40;  // conjugate-gradient steps of the warp data term per frame, 0 = off (Opt is capped at linearIter = 100,
                                          // kinfu.cpp:118; on the synthetic sequence the energy has converged to 4 digits by 40)
    };

    class KinFu
    {
    public:
        typedef std::shared_ptr<KinFu> Ptr;
        KinFu(const KinFuParams& params);
        virtual ~KinFu() {}

        const KinFuParams& params() const { return params_; }
        KinFuParams& params() { return params_; }
        const cuda::TsdfVolume& tsdf() const { return *volume_; }
        cuda::TsdfVolume& tsdf() { return *volume_; }
        const cuda::ProjectiveICP& icp() const { return *icp_; }
        cuda::ProjectiveICP& icp() { return *icp_; }
        const WarpField& getWarp() const { return *warp_; }
        WarpField& getWarp() { return *warp_; }

        void reset();
        bool operator()(const cuda::Depth& depth, const cuda::Image& image = cuda::Image());
        void dynamicfusion(cuda::Depth& depth, cuda::Cloud live_frame, cuda::Normals current_normals);
        Affine3f getCameraPose(int time = -1) const;

    protected:
        /// optimiser_->optimiseWarpData(canonical, canonical_normals, live, canonical_normals) (kinfu.cpp:389) on the host-staged data
        /// flow (device_resident = false); the default runs WarpField::energy_data, the GPU data-term solve
        virtual void optimiseWarp(std::vector<Vec3f>& canonical, std::vector<Vec3f>& canonical_normals, const std::vector<Vec3f>& live);
    private:
        void allocate_buffers();
        int frame_counter_;
        KinFuParams params_;
        std::vector<Affine3f> poses_;
        cuda::Dists dists_;
        cuda::Frame curr_, prev_, first_;
        std::unique_ptr<cuda::TsdfVolume> volume_;
        std::unique_ptr<cuda::ProjectiveICP> icp_;
        std::unique_ptr<WarpField> warp_;
        // scratch of the device-resident dynamicfusion()
        cuda::Cloud df_cloud_; cuda::Normals df_normals_;
        cuda::DeviceArray<float> df_points3_, df_normals3_, df_live3_;
        cuda::DeviceArray<Point> df_warped4_;
        cuda::Normals df_live_normals_;
    };
}
