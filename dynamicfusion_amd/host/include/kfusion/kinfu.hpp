// kfusion/kinfu.hpp -- kfusion::KinFuParams / kfusion::KinFu with the reference's interface
// (/root/reference/kfusion/include/kfusion/kinfu.hpp:15-112) minus the Opt/Ceres solver (the data term is solved on the GPU instead).
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
    // The knobs of the pipeline, field for field those of the reference's KinFuParams (kinfu.hpp:15-48), followed by the extensions
    // of this implementation.  Units: pixels, metres, radians, frames.
    struct KinFuParams
    {
        static KinFuParams default_params();                   // 512^3 / 3 m volume, fx = fy = 525                    kinfu.cpp:55-89
        static KinFuParams default_params_dynamicfusion();     // 256^3 / 1 m volume, fx = fy = 570.342 (what demo.cpp uses)  :15-50

        // sensor
        int cols, rows;
        Intr intr;
        // TSDF volume
        Vec3i volume_dims;                   // voxels per axis (dims[0] % 32 == 0)
        Vec3f volume_size;                   // metres per axis
        Affine3f volume_pose;                // volume -> world
        float tsdf_trunc_dist;               // clamped to >= 2.1 voxels by TsdfVolume::setTruncDist
        int tsdf_max_weight;
        float tsdf_min_camera_movement;      // unused by the pipeline, as in the reference
        float raycast_step_factor, gradient_delta_factor;      // in voxel sizes
        // depth front-end
        float bilateral_sigma_depth, bilateral_sigma_spatial;
        int bilateral_kernel_size;
        float icp_truncate_depth_dist;       // 0 = no truncation
        // tracker
        float icp_dist_thres, icp_angle_thres;
        std::vector<int> icp_iter_num;       // per pyramid level, finest first
        Vec3f light_pose;                    // metres, camera frame: renderImage's light

        // ---- extensions (defaults reproduce the reference's observable behaviour unless noted)
        int max_warp_nodes = 65535;          // node ids are 16-bit: the stride-50 sampling of the seed cloud widens if it must
        bool warped_fusion = false;          // true: per-voxel warped integrate (the north-star kernel) instead of surface_fusion
        bool use_depth_pyramids = false;     // the reference's USE_DEPTH build (internal.hpp:6): depth pyramids + masked-depth ICP
        bool device_resident = true;         // dynamicfusion() keeps its point sets on the GPU; false = the reference's host staging
        int warp_solver_iterations = 40;     // CG steps of the warp data term per frame, 0 = off (Opt's cap is linearIter = 100,
                                             // kinfu.cpp:118; the synthetic sequence has converged to 4 digits by 40)
    };

    class KinFu
    {
    public:
#ifdef KFUSION_USE_OPENCV
        typedef cv::Ptr<KinFu> Ptr;                                         // kinfu.hpp:52
#else
        typedef std::shared_ptr<KinFu> Ptr;
#endif
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
        /// kinfu.cpp:312-343: the model as the tracker last saw it (prev_ pyramid level 0).  flags 1 (and anything outside 1..3): Phong
        /// shading, 2: normals as colours, 3: both side by side (image is then 2 * cols wide)
        void renderImage(cuda::Image& image, int flags = 0);
        void dynamicfusion(cuda::Depth& depth, cuda::Cloud live_frame, cuda::Normals current_normals);
        /// kinfu.cpp:408-436: the same views of a fresh ray-cast from `pose`
        void renderImage(cuda::Image& image, const Affine3f& pose, int flags = 0);
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
        cuda::Cloud points_; cuda::Normals normals_; cuda::Depth depths_;   // renderImage(image, pose, flags)
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
