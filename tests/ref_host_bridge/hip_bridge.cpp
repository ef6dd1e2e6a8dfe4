// hip_bridge.cpp -- INTEGRATION.md section A as a file: the ONE translation unit a maintainer of mihaibujanca/dynamicfusion adds in place of
// kfusion/src/cuda/tsdf_volume.cu and imgproc.cu.  It defines the free functions kfusion::device::* that the reference's host classes call
// (declared in the reference's private header kfusion/src/internal.hpp:104-143) and forwards each to the C-ABI of libdfusion_hip.so
// (include/dfusion.h).  Compiled by the HOST compiler against the reference's own headers; nothing here is device code.
//
// tests/ref_host_bridge/build_ref_host.py compiles it together with the reference's UNMODIFIED kfusion/src/{tsdf_volume, imgproc, precomp,
// device_memory}.cpp (read where they lie under /root/reference) into tests/ref_host_bridge/_build/libkfusion_refhost.so, and
// tests/test_gpu_ref_host_bridge.py drives the REFERENCE's kfusion::cuda::TsdfVolume on the MI355X through it.
//
// Not bridged (nothing compiled here calls them): device::project (tsdf_volume.cu:180, no caller in the reference),
// device::mergePointNormal, device::ComputeIcpHelper
// (proj_icp.cu; the C-ABI's dfusion_icp_* are its counterpart, bound by the mirror's ProjectiveICP).
#include "precomp.hpp"           // the reference's: device::TsdfVolume, Aff3f, Projector, Reprojector, Dists, Points, Normals, Depth, Image
#include "dfusion.h"
#include <algorithm>
#include <cmath>
#include <cstdio>

using namespace kfusion;
using namespace kfusion::device;

static_assert(sizeof(Aff3f) == 12 * sizeof(float), "device::Aff3f = 3 rows of float3 + float3 t: the C-ABI's 12 packed floats (internal.hpp:26-27)");
static_assert(sizeof(device::Mat3f) == 9 * sizeof(float), "device::Mat3f = 9 packed floats");

static DfVolume c_vol(const device::TsdfVolume& v)            // field for field, internal.hpp:29-49
{
    DfVolume r;
    r.data = v.data; r.dims[0] = v.dims.x; r.dims[1] = v.dims.y; r.dims[2] = v.dims.z;
    r.voxel_size[0] = v.voxel_size.x; r.voxel_size[1] = v.voxel_size.y; r.voxel_size[2] = v.voxel_size.z;
    r.trunc_dist = v.trunc_dist; r.max_weight = v.max_weight;
    return r;
}
static void chk(int rc, const char* file, int line) { if (rc) kfusion::cuda::error(dfusion_error_string(rc), file, line, ""); }
#define DF_CHK(expr) chk((expr), __FILE__, __LINE__)

// Which forwards have run (a test hook, nothing a maintainer needs): tests/ref_host_bridge/ref_host_frame.cpp prints the report and
// tests/test_gpu_ref_host_bridge.py asserts that every one of the 20 did.
static const char* const g_forward_names[20] = {
    "clear_volume", "integrate", "raycast(Points)", "raycast(Depth)", "project_and_remove(PtrStepSz&)", "project_and_remove(const PtrStepSz&)",
    "extractCloud", "extractNormals", "compute_dists", "truncateDepth", "bilateralFilter", "depthPyr", "resizeDepthNormals", "resizePointsNormals",
    "computeNormalsAndMaskDepth", "computePointNormals", "renderImage(Depth)", "renderImage(Points)", "renderTangentColors", "cloud_to_depth"};
static unsigned g_forward_calls[20];
#define DF_FORWARD(i) (++g_forward_calls[i])
extern "C" int hip_bridge_forward_report(char* buf, int cap)       // "ran/20: name=count ..." ; returns the number of forwards that ran
{
    int ran = 0, o = 0;
    for (int i = 0; i < 20; ++i) ran += g_forward_calls[i] ? 1 : 0;
    o += std::snprintf(buf + o, o < cap ? cap - o : 0, "%d/20:", ran);
    for (int i = 0; i < 20; ++i) o += std::snprintf(buf + (o < cap ? o : cap), o < cap ? cap - o : 0, " %s=%u", g_forward_names[i], g_forward_calls[i]);
    return ran;
}

// The front-end entry points of the C-ABI take the intrinsics (fx, fy, cx, cy) and form 1/fx, 1/fy themselves, exactly as
// device::Reprojector's constructor does (precomp.cpp:55); the reference's device functions receive the finished Reprojector.  Any f with
// RN(1.f / f) == finv gives the library the same reprojector bits (these kernels read nothing else of the intrinsics), and the original
// fx is always such an f: search the neighbours of RN(1 / finv).
static float focal_of(float finv)
{
    const float f0 = 1.f / finv;
    for (int d = 0; d <= 4; ++d) {
        float up = f0, dn = f0;
        for (int i = 0; i < d; ++i) { up = std::nextafterf(up, INFINITY); dn = std::nextafterf(dn, -INFINITY); }
        if (1.f / up == finv) return up;
        if (1.f / dn == finv) return dn;
    }
    kfusion::cuda::error("hip_bridge: no focal length reproduces the reprojector", __FILE__, __LINE__, "");
    return f0;
}
static void intr_of(const Reprojector& r, float out[4]) { out[0] = focal_of(r.finv.x); out[1] = focal_of(r.finv.y); out[2] = r.c.x; out[3] = r.c.y; }

// ---------------------------------------------------------------------------------------- tsdf volume (internal.hpp:104-115)
void kfusion::device::clear_volume(TsdfVolume volume)                                               // tsdf_volume.cu:32
{
    DF_FORWARD(0);
    DF_CHK(dfusion_clear(c_vol(volume), nullptr, 0));
}

void kfusion::device::integrate(const Dists& dists, TsdfVolume& volume, const Aff3f& aff, const Projector& proj)   // tsdf_volume.cu:141
{
    DF_FORWARD(1);
    const float p[4] = {proj.f.x, proj.f.y, proj.c.x, proj.c.y};
    DF_CHK(dfusion_integrate(dists.data, dists.step, dists.cols, dists.rows, c_vol(volume), nullptr, (const float*)&aff, p, nullptr, 0));
    cudaSafeCall(cudaDeviceSynchronize());                                                          // tsdf_volume.cu:160
}

void kfusion::device::raycast(const TsdfVolume& volume, const Aff3f& aff, const Mat3f& Rinv, const Reprojector& reproj,
                              Points& points, Normals& normals, float step_factor, float delta_factor)   // tsdf_volume.cu:459
{
    DF_FORWARD(2);
    const float rp[4] = {reproj.finv.x, reproj.finv.y, reproj.c.x, reproj.c.y};
    DF_CHK(dfusion_raycast_points(c_vol(volume), nullptr, (const float*)&aff, (const float*)&Rinv, rp, (float*)points.ptr(), points.step(),
                                  (float*)normals.ptr(), normals.step(), points.cols(), points.rows(), step_factor, delta_factor, nullptr, 0));
}

void kfusion::device::raycast(const TsdfVolume& volume, const Aff3f& aff, const Mat3f& Rinv, const Reprojector& reproj,
                              Depth& depth, Normals& normals, float step_factor, float delta_factor)     // tsdf_volume.cu:441
{
    DF_FORWARD(3);
    const float rp[4] = {reproj.finv.x, reproj.finv.y, reproj.c.x, reproj.c.y};
    DF_CHK(dfusion_raycast_depth(c_vol(volume), nullptr, (const float*)&aff, (const float*)&Rinv, rp, depth.ptr(), depth.step(),
                                 (float*)normals.ptr(), normals.step(), depth.cols(), depth.rows(), step_factor, delta_factor, 0));
}

// the reference samples `dists` and zeroes it in the same launch (racy, tsdf_volume.cu:114-139); the C-ABI samples a snapshot
static void project_and_remove_impl(ushort* data, size_t step, int cols, int rows, Points& vertices, const Projector& proj)
{
    DeviceArray2D<ushort> snapshot(rows, cols);
    cudaSafeCall(cudaMemcpy2D(snapshot.ptr(), snapshot.step(), data, step, (size_t)cols * sizeof(ushort), rows, cudaMemcpyDeviceToDevice));
    const float p[4] = {proj.f.x, proj.f.y, proj.c.x, proj.c.y};
    DF_CHK(dfusion_project_and_remove(snapshot.ptr(), snapshot.step(), data, step, cols, rows, (float*)vertices.ptr(),
                                      (unsigned long long)vertices.rows() * vertices.cols(), p, nullptr, nullptr, 0));
    cudaSafeCall(cudaDeviceSynchronize());
}
void kfusion::device::project_and_remove(PtrStepSz<ushort>& dists, Points& vertices, const Projector& proj)          // internal.hpp:108
{
    DF_FORWARD(4);
    project_and_remove_impl(dists.data, dists.step, dists.cols, dists.rows, vertices, proj);
}
void kfusion::device::project_and_remove(const PtrStepSz<ushort>& dists, Points& vertices, const Projector& proj)    // tsdf_volume.cu:164
{
    DF_FORWARD(5);
    project_and_remove_impl(dists.data, dists.step, dists.cols, dists.rows, vertices, proj);
}

size_t kfusion::device::extractCloud(const TsdfVolume& volume, const Aff3f& aff, PtrSz<Point> output)               // tsdf_volume.cu:799
{
    DF_FORWARD(6);
    DeviceArray<unsigned long long> count(1);
    DF_CHK(dfusion_extract_cloud(c_vol(volume), nullptr, (const float*)&aff, (float*)output.data, output.size, count.ptr(), 0));
    unsigned long long n = 0;
    count.download(&n);
    return (size_t)std::min<unsigned long long>(n, output.size);
}

void kfusion::device::extractNormals(const TsdfVolume& volume, const PtrSz<Point>& points, const Aff3f& aff, const Mat3f& Rinv,
                                     float gradient_delta_factor, float4* output)                                    // tsdf_volume.cu:817
{
    DF_FORWARD(7);
    DF_CHK(dfusion_extract_normals(c_vol(volume), nullptr, (const float*)&aff, (const float*)&Rinv, (const float*)points.data, points.size,
                                   gradient_delta_factor, (float*)output, 0));
    cudaSafeCall(cudaDeviceSynchronize());
}

// ---------------------------------------------------------------------------------------- image processing (internal.hpp:123-140)
void kfusion::device::compute_dists(const Depth& depth, Dists dists, float2 f, float2 c)                             // imgproc.cu:287
{
    DF_FORWARD(8);
    const float intr[4] = {f.x, f.y, c.x, c.y};
    DF_CHK(dfusion_compute_dists(depth.ptr(), depth.step(), dists.data, dists.step, depth.cols(), depth.rows(), intr, 0));
}
void kfusion::device::truncateDepth(Depth& depth, float max_dist)                                                    // imgproc.cu:78
{
    DF_FORWARD(9);
    DF_CHK(dfusion_truncate_depth(depth.ptr(), depth.step(), depth.cols(), depth.rows(), max_dist, 0));
}
void kfusion::device::bilateralFilter(const Depth& src, Depth& dst, int kernel_size, float sigma_spatial, float sigma_depth)   // imgproc.cu:47
{
    DF_FORWARD(10);
    DF_CHK(dfusion_bilateral_filter(src.ptr(), src.step(), dst.ptr(), dst.step(), src.cols(), src.rows(), kernel_size, sigma_spatial, sigma_depth, 0));
}
void kfusion::device::depthPyr(const Depth& source, Depth& pyramid, float sigma_depth)                               // imgproc.cu:127
{
    DF_FORWARD(11);
    DF_CHK(dfusion_depth_pyramid(source.ptr(), source.step(), source.cols(), source.rows(), pyramid.ptr(), pyramid.step(), sigma_depth, 0));
}
void kfusion::device::resizeDepthNormals(const Depth& depth, const Normals& normals, Depth& depth_out, Normals& normals_out)   // imgproc.cu:349
{
    DF_FORWARD(12);
    DF_CHK(dfusion_resize_depth_normals(depth.ptr(), depth.step(), (const float*)normals.ptr(), normals.step(), depth.cols(), depth.rows(),
                                        depth_out.ptr(), depth_out.step(), (float*)normals_out.ptr(), normals_out.step(), 0));
}
void kfusion::device::resizePointsNormals(const Points& points, const Normals& normals, Points& points_out, Normals& normals_out)   // imgproc.cu:404
{
    DF_FORWARD(13);
    DF_CHK(dfusion_resize_points_normals((const float*)points.ptr(), points.step(), (const float*)normals.ptr(), normals.step(), points.cols(),
                                         points.rows(), (float*)points_out.ptr(), points_out.step(), (float*)normals_out.ptr(), normals_out.step(), 0));
}
void kfusion::device::computeNormalsAndMaskDepth(const Reprojector& reproj, Depth& depth, Normals& normals)          // imgproc.cu:192
{
    DF_FORWARD(14);
    float intr[4]; intr_of(reproj, intr);
    DF_CHK(dfusion_compute_normals_mask_depth(depth.ptr(), depth.step(), (float*)normals.ptr(), normals.step(), depth.cols(), depth.rows(), intr, 0));
}
void kfusion::device::computePointNormals(const Reprojector& reproj, const Depth& depth, Points& points, Normals& normals)   // imgproc.cu:243
{
    DF_FORWARD(15);
    float intr[4]; intr_of(reproj, intr);
    DF_CHK(dfusion_compute_point_normals(depth.ptr(), depth.step(), (float*)points.ptr(), points.step(), (float*)normals.ptr(), normals.step(),
                                         depth.cols(), depth.rows(), intr, 0));
}
void kfusion::device::renderImage(const Depth& depth, const Normals& normals, const Reprojector& reproj, const Vec3f& light_pose, Image& image)   // imgproc.cu:530
{
    DF_FORWARD(16);
    float intr[4]; intr_of(reproj, intr);
    const float light[3] = {light_pose.x, light_pose.y, light_pose.z};
    DF_CHK(dfusion_render_image_depth(depth.ptr(), depth.step(), (const float*)normals.ptr(), normals.step(), depth.cols(), depth.rows(), intr, light,
                                      (unsigned char*)image.ptr(), image.step(), 0));
}
void kfusion::device::renderImage(const Points& points, const Normals& normals, const Reprojector&, const Vec3f& light_pose, Image& image)        // imgproc.cu:539
{
    DF_FORWARD(17);
    const float light[3] = {light_pose.x, light_pose.y, light_pose.z};
    DF_CHK(dfusion_render_image_points((const float*)points.ptr(), points.step(), (const float*)normals.ptr(), normals.step(), points.cols(),
                                       points.rows(), light, (unsigned char*)image.ptr(), image.step(), 0));
}
void kfusion::device::renderTangentColors(const Normals& normals, Image& image)                                     // imgproc.cu:576
{
    DF_FORWARD(18);
    DF_CHK(dfusion_render_tangent_colors((const float*)normals.ptr(), normals.step(), normals.cols(), normals.rows(), (unsigned char*)image.ptr(),
                                         image.step(), 0));
}
void kfusion::device::cloud_to_depth(const Points& cloud, Depth depth)                                              // imgproc.cu:296
{
    DF_FORWARD(19);
    DF_CHK(dfusion_cloud_to_depth((const float*)cloud.ptr(), cloud.step(), depth.ptr(), depth.step(), cloud.cols(), cloud.rows(), 0));
}
