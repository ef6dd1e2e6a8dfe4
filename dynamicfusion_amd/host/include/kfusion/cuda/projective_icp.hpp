// kfusion/cuda/projective_icp.hpp -- kfusion::cuda::ProjectiveICP with the reference's interface
// (/root/reference/kfusion/include/kfusion/cuda/projective_icp.hpp:9-48).  The correspondence search and the 27-sum
// reduction run on the GPU (dfusion_icp_sums_*), the 6x6 solve on the host like the reference.  The never-implemented
// Frame overload (projective_icp.cpp:113-127, CV_Assert(!"Not implemented")) is omitted.
#pragma once
#include <vector>
#include <kfusion/types.hpp>

namespace kfusion
{
    namespace cuda
    {
        class ProjectiveICP
        {
        public:
            enum { MAX_PYRAMID_LEVELS = 4 };
            typedef std::vector<Depth> DepthPyr;
            typedef std::vector<Cloud> PointsPyr;
            typedef std::vector<Normals> NormalsPyr;

            ProjectiveICP();
            virtual ~ProjectiveICP();

            float getDistThreshold() const;
            void setDistThreshold(float distance);
            float getAngleThreshold() const;
            void setAngleThreshold(float angle);
            void setIterationsNum(const std::vector<int>& iters);
            int getUsedLevelsNum() const;
            /// true (default): the whole Gauss-Newton loop is one enqueue with the 6x6 solve on the GPU; false: the reference's control
            /// flow (stream synchronise + host solve per iteration).  Poses agree to ~1e-6.
            void setDeviceLoop(bool on) { device_loop_ = on; }

            /** masked depth: "if depth(y,x) is not zero, then normals(y,x) surely is not qnan" */
            virtual bool estimateTransform(Affine3f& affine, const Intr& intr, const DepthPyr& dcurr, const NormalsPyr ncurr, const DepthPyr dprev, const NormalsPyr nprev);
            virtual bool estimateTransform(Affine3f& affine, const Intr& intr, const PointsPyr& vcurr, const NormalsPyr ncurr, const PointsPyr vprev, const NormalsPyr nprev);
        private:
            bool iterate(Affine3f& affine, const Intr& intr, const void* const* curr, const NormalsPyr& ncurr, const void* const* prev,
                         const NormalsPyr& nprev, const size_t* curr_step, const size_t* prev_step, bool depth_variant);
            std::vector<int> iters_;
            float angle_thres_;
            float dist_thres_;
            bool device_loop_ = true;
            DeviceArray<float> buffer_;                      // partial sums + the 27 results
        };
    }
}
