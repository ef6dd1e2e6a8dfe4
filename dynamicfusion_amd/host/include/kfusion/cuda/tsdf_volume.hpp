// kfusion/cuda/tsdf_volume.hpp -- kfusion::cuda::TsdfVolume, source compatible with
// /root/reference/kfusion/include/kfusion/cuda/tsdf_volume.hpp:11-100 for the hot path; every method forwards to the
// C-ABI in include/dfusion.h.  get_cloud_host()/get_normal_host() return std::vector<Point> (the reference returns
// cv::Mat 1xN CV_32FC4 -- same bytes); psdf / surface_fusion's CPU loop is a SURVEY.md 8(f) "next" row and not declared;
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

            DeviceArray<Point> fetchCloud(DeviceArray<Point>& cloud_buffer) const;                        // tsdf_volume.cpp:181-199
            void fetchNormals(const DeviceArray<Point>& cloud, DeviceArray<Normal>& normals) const;       // :206-218
            void compute_points();                                                                        // :313-318
            void compute_normals();                                                                       // :320-325
            const std::vector<Point>& get_cloud_host() const { return cloud_host_; }
            const std::vector<Normal>& get_normal_host() const { return normal_host_; }

        private:
            DeviceArray<Point> cloud_buffer_, cloud_;
            DeviceArray<Normal> normal_buffer_;
            std::vector<Point> cloud_host_;
            std::vector<Normal> normal_host_;
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
