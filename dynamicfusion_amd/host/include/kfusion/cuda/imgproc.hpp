// kfusion/cuda/imgproc.hpp -- the image operators KinFu::operator() calls, with the reference's names and argument order
// (/root/reference/kfusion/include/kfusion/cuda/imgproc.hpp; host wrappers kfusion/src/imgproc.cpp).  Every function allocates its
// outputs like the reference's wrapper does and enqueues one or two HIP kernels through the C-ABI (include/dfusion.h); nothing
// synchronises except waitAllDefaultStream.
//
//   image types   Depth / Dists : DeviceArray2D<unsigned short>   (millimetres / IEEE-half bits of the ray length in metres)
//                 Cloud / Normals: DeviceArray2D<Point>            (float4, invalid pixels are NaN)
//   sigma_depth, threshold are in METRES (converted to millimetres inside, imgproc.cu:50,83,130)
#pragma once
#include <kfusion/types.hpp>

namespace kfusion { namespace cuda {

// ---- synchronisation (imgproc.cpp:41-44: cudaDeviceSynchronize)
void waitAllDefaultStream();

// ---- per-frame inputs of the fusion
// ray length per pixel for TsdfVolume::integrate: dists = half(depth_mm * sqrt(xl^2 + yl^2 + 1) / 1000)      imgproc.cpp:87-91
void computeDists(const Depth& depth, Dists& dists, const Intr& intr);

// points image -> depth image in millimetres (z * 1000; a NaN point gives 0)                                   imgproc.cpp:98-103
void cloudToDepth(const Cloud& cloud, Depth& depth);

// ---- depth pyramid of the tracker
// edge-preserving smoothing, window ksz x ksz                                                                 imgproc.cpp:10-14
void depthBilateralFilter(const Depth& in, Depth& out, int ksz, float sigma_spatial, float sigma_depth);
// depth > threshold <- 0, in place                                                                            imgproc.cpp:21-24
void depthTruncation(Depth& depth, float threshold);
// half-resolution level: mean of the 5x5 neighbours within 3 sigma_depth of the centre                        imgproc.cpp:32-36
void depthBuildPyramid(const Depth& depth, Depth& pyramid, float sigma_depth);

// ---- vertex / normal maps of one pyramid level (intr = that level's intrinsics, Intr::operator()(level))
// USE_DEPTH build: normals from the depth image, pixels without a normal are zeroed in `depth`                imgproc.cpp:52-60
void computeNormalsAndMaskDepth(const Intr& intr, Depth& depth, Normals& normals);
// default build: back-projected points + normals                                                              imgproc.cpp:69-79
void computePointNormals(const Intr& intr, const Depth& depth, Cloud& points, Normals& normals);

// ---- coarser levels of the ray-cast model (2x2 averages, NaN / 0 if any of the four is invalid)             imgproc.cpp:112-141
void resizeDepthNormals(const Depth& depth, const Normals& normals, Depth& depth_out, Normals& normals_out);
void resizePointsNormals(const Cloud& points, const Normals& normals, Cloud& points_out, Normals& normals_out);

// ---- the views KinFu::renderImage returns (BGRA, `image` is created rows x cols)                              imgproc.cpp:152-201
// Phong shading of a depth image / a points image with its normals, light at light_pose (metres, camera frame)
void renderImage(const Depth& depth, const Normals& normals, const Intr& intr, const Vec3f& light_pose, Image& image);
void renderImage(const Cloud& points, const Normals& normals, const Intr& intr, const Vec3f& light_pose, Image& image);
// normals as colours
void renderTangentColors(const Normals& normals, Image& image);

} }
