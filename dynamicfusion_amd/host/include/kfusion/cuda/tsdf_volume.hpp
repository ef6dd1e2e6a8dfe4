// kfusion/cuda/tsdf_volume.hpp -- kfusion::cuda::TsdfVolume, source compatible with
// /root/reference/kfusion/include/kfusion/cuda/tsdf_volume.hpp:11-100 for the hot path; every method forwards to the
// C-ABI in include/dfusion.h.  get_cloud_host()/get_normal_host() return std::vector<Point> (the reference returns
// cv::Mat 1xN CV_32FC4 -- same bytes); psdf / surface_fusion run on the GPU (dfusion_project_and_remove);
// getGridOrigin / setGridOrigin are declared but never defined in the reference (:49-50) and omitted.
#pragma once
#include <kfusion/types.hpp>

namespace kfusion
{
    class WarpField;
    namespace cuda
    {
        class TsdfVolume
        {
        public:
            TsdfVolume(const Vec3i& dims);
            virtual ~TsdfVolume();

            void create(const Vec3i& dims);

            Vec3i getDims() const;
            Vec3f getVoxelSize() const;

            const CudaData data() const;
            CudaData data();

            Vec3f getSize() const;
            void setSize(const Vec3f& size);

            float getTruncDist() const;
            void setTruncDist(float distance);

            int getMaxWeight() const;
            void setMaxWeight(int weight);

            Affine3f getPose() const;
            void setPose(const Affine3f& pose);

            float getRaycastStepFactor() const;
            void setRaycastStepFactor(float factor);

            float getGradientDeltaFactor() const;
            void setGradientDeltaFactor(float factor);

            virtual void clear();
            virtual void applyAffine(const Affine3f& affine);
            virtual void integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr);
            virtual void raycast(const Affine3f& camera_pose, const Intr& intr, Depth& depth, Normals& normals);
            virtual void raycast(const Affine3f& camera_pose, const Intr& intr, Cloud& points, Normals& normals);

            /// The north-star fusion step (what surface_fusion, tsdf_volume.cpp:228-255, is meant to be): every voxel is
            /// warped by the dual-quaternion blend of its k nearest nodes before the projective update.
            virtual void integrate(const Dists& dists, const Affine3f& camera_pose, const Intr& intr, const WarpField& warp);

            void swap(CudaData& data);

            /// tsdf_volume.cpp:228-255.  Observable behaviour of the reference: depth pixels explained by a warped model
            /// point are zeroed, the leftover depth is fused RIGIDLY; the per-point weights loop (:241-254) has no effect
            /// there (update lines commented out) and is not run.  psdf is handed dists computed from `depth` (the
            /// reference binds the mm image as half, :235/:281 -- fixed).
            void surface_fusion(const WarpField& warp_field, std::vector<Vec3f> warped, std::vector<Vec3f> canonical,
                                cuda::Depth& depth, const Affine3f& camera_pose, const Intr& intr);
            /// tsdf_volume.cpp:266-292: ro[i] = dists at the projection of warped[i] - warped[i].z (NaN when the point is
            /// NaN or projects outside); pixels of `dists` hit by a point are zeroed.
            std::vector<float> psdf(const std::vector<Vec3f>& warped, Dists& dists, const Intr& intr);
            /// tsdf_volume.cpp:300-306
            float weighting(const std::vector<float>& dist_sqr, int k) const;

            DeviceArray<Point> fetchCloud(DeviceArray<Point>& cloud_buffer) const;                        // tsdf_volume.cpp:181-199
            void fetchNormals(const DeviceArray<Point>& cloud, DeviceArray<Normal>& normals) const;       // :206-218
            void compute_points();                                                                        // :313-318
            void compute_normals();                                                                       // :320-325
            /// host copies are fetched lazily: compute_points / compute_normals only run the device kernels, the download
            /// happens on the first get_*_host() after them (the reference downloads eagerly, tsdf_volume.cpp:316,323)
            const std::vector<Point>& get_cloud_host() const;
            const std::vector<Normal>& get_normal_host() const;
            const DeviceArray<Point>& get_cloud_device() const { return cloud_; }

            /// surface_fusion with the warped model points already on the device (float4, camera... world frame as the host overload):
            /// same result, no host staging
            void surface_fusion(const WarpField& warp_field, DeviceArray<Point>& warped, cuda::Depth& depth, const Affine3f& camera_pose,
                                const Intr& intr);

        private:
            DeviceArray<Point> cloud_buffer_, cloud_;
            DeviceArray<Normal> normal_buffer_;
            mutable std::vector<Point> cloud_host_;
            mutable std::vector<Normal> normal_host_;
            mutable bool cloud_host_stale_ = false, normal_host_stale_ = false;
            Dists fusion_dists_;                            // scratch of surface_fusion
            CudaData data_;
            float trunc_dist_;
            float max_weight_;                              // stored as float in the reference too (tsdf_volume.hpp:86)
            Vec3i dims_;
            Vec3f size_;
            Affine3f pose_;
            float gradient_delta_factor_;
            float raycast_step_factor_;
        };
    }
}
