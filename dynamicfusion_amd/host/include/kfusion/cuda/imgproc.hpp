// kfusion/cuda/imgproc.hpp -- the image ops KinFu::operator() calls (/root/reference/kfusion/include/kfusion/cuda/imgproc.hpp);
// rendering (renderImage, renderTangentColors, cloudToDepth, mergePointNormal) is viz and out of scope.
#pragma once
#include <kfusion/types.hpp>
namespace kfusion
{
    namespace cuda
    {
        /// depth mm -> ray length metres as half bits (/root/reference/kfusion/src/imgproc.cpp:87-91)
        void computeDists(const Depth& depth, Dists& dists, const Intr& intr);
        /// cudaDeviceSynchronize stand-in (imgproc.cpp:41-44)
        void waitAllDefaultStream();
        /// imgproc.cpp:10-14, :21-24, :32-36 (sigma_depth and threshold in metres)
        void depthBilateralFilter(const Depth& in, Depth& out, int ksz, float sigma_spatial, float sigma_depth);
        void depthTruncation(Depth& depth, float threshold);
        void depthBuildPyramid(const Depth& depth, Depth& pyramid, float sigma_depth);
        /// imgproc.cpp:52-60, :69-79
        void computeNormalsAndMaskDepth(const Intr& intr, Depth& depth, Normals& normals);
        void computePointNormals(const Intr& intr, const Depth& depth, Cloud& points, Normals& normals);
        /// imgproc.cpp:112-122, :131-141
        void resizeDepthNormals(const Depth& depth, const Normals& normals, Depth& depth_out, Normals& normals_out);
        void resizePointsNormals(const Cloud& points, const Normals& normals, Cloud& points_out, Normals& normals_out);
    }
}
