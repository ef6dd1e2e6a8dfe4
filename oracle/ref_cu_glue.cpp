// C entry points over the reference's OWN CUDA sources, compiled for the host through oracle/cuda_shim/
// (see oracle/Makefile target `ref_cu`): kfusion/src/cuda/tsdf_volume.cu, imgproc.cu, proj_icp.cu,
// kfusion/src/device_memory.cpp and the three device-struct constructors of kfusion/src/precomp.cpp:24-55.
// TEST INFRASTRUCTURE ONLY -- this is what pins oracle/dfusion_oracle.c (and through it the HIP kernels) to
// the reference's code rather than to a reading of it.  Nothing here re-implements reference arithmetic:
// each function marshals dense host arrays into the reference's containers (kfusion::cuda::DeviceArray2D,
// device::TsdfVolume, Projector, Reprojector, Aff3f) and calls the reference's host wrapper
// (`kfusion::device::integrate`, `raycast`, `compute_dists`, ... as declared in kfusion/src/internal.hpp:104-141).
//
// Affines are 12 floats: R row-major (9) then t (3).  Images are dense (pitch = cols * sizeof(T)).
#include "cuda_runtime_api.h"
#include <kfusion/cuda/device_array.hpp>
#include "internal.hpp"          // /root/reference/kfusion/src/internal.hpp
#include <cstdio>
#include <vector>

using namespace kfusion;
using namespace kfusion::device;

#define RCU_API extern "C" __attribute__((visibility("default")))

namespace
{
Aff3f make_aff(const float a[12])
{
    Aff3f r;
    for (int i = 0; i < 3; ++i) r.R.data[i] = make_float3(a[3 * i], a[3 * i + 1], a[3 * i + 2]);
    r.t = make_float3(a[9], a[10], a[11]);
    return r;
}
Mat3f make_mat(const float m[9])
{
    Mat3f r;
    for (int i = 0; i < 3; ++i) r.data[i] = make_float3(m[3 * i], m[3 * i + 1], m[3 * i + 2]);
    return r;
}
struct Vol { unsigned int* data; int dims[3]; float vsize[3]; float trunc; int max_weight; };
device::TsdfVolume make_vol(const Vol& v)
{
    return device::TsdfVolume((ushort2*)v.data, make_int3(v.dims[0], v.dims[1], v.dims[2]),
                              make_float3(v.vsize[0], v.vsize[1], v.vsize[2]), v.trunc, v.max_weight);
}
struct Modes
{
    int fiber, par;
    Modes(int f, int p) : fiber(cuda_shim::g_fiber_mode), par(cuda_shim::g_parallel_blocks) { cuda_shim::g_fiber_mode = f; cuda_shim::g_parallel_blocks = p; }
    ~Modes() { cuda_shim::g_fiber_mode = fiber; cuda_shim::g_parallel_blocks = par; }
};
template <class T> void up(cuda::DeviceArray2D<T>& a, const void* host, int rows, int cols) { a.upload(host, (size_t)cols * sizeof(T), rows, cols); }
template <class T> void down(const cuda::DeviceArray2D<T>& a, void* host) { a.download(host, (size_t)a.cols() * sizeof(T)); }
}  // namespace

RCU_API int refcu_abi() { return 1; }

// ---- tsdf_volume.cu ----------------------------------------------------------------------------------------------
RCU_API void refcu_clear(Vol v)
{
    Modes m(0, 1);
    clear_volume(make_vol(v));                                                   // tsdf_volume.cu:32-41
}

RCU_API void refcu_integrate(const unsigned short* dists, int rows, int cols, Vol v, const float vol2cam[12], const float intr[4])
{
    Modes m(0, 1);
    cuda::DeviceArray2D<unsigned short> d; up(d, dists, rows, cols);
    device::TsdfVolume vol = make_vol(v);
    Projector proj(intr[0], intr[1], intr[2], intr[3]);
    integrate(d, vol, make_aff(vol2cam), proj);                                  // tsdf_volume.cu:141-162
}

RCU_API void refcu_raycast_points(Vol v, const float cam2vol[12], const float Rinv[9], const float intr[4], int rows, int cols,
                                  float step_factor, float delta_factor, float* points, float* normals)
{
    Modes m(0, 1);
    Points p(rows, cols); Normals n(rows, cols);
    Reprojector reproj(intr[0], intr[1], intr[2], intr[3]);
    raycast(make_vol(v), make_aff(cam2vol), make_mat(Rinv), reproj, p, n, step_factor, delta_factor);   // :458-474
    down(p, points); down(n, normals);
}

RCU_API void refcu_raycast_depth(Vol v, const float cam2vol[12], const float Rinv[9], const float intr[4], int rows, int cols,
                                 float step_factor, float delta_factor, unsigned short* depth, float* normals)
{
    Modes m(0, 1);
    Depth d(rows, cols); Normals n(rows, cols);
    Reprojector reproj(intr[0], intr[1], intr[2], intr[3]);
    raycast(make_vol(v), make_aff(cam2vol), make_mat(Rinv), reproj, d, n, step_factor, delta_factor);   // :440-456
    down(d, depth); down(n, normals);
}

// project_kernel (tsdf_volume.cu:113-139) through project_and_remove (:165-180).  The reference reads and zeroes the
// same image from concurrent threads; here threads run in block/thread order, so the result is the sequential one.
RCU_API void refcu_project_and_remove(unsigned short* dists, int rows, int cols, float* points, int prows, int pcols, const float intr[4])
{
    Modes m(0, 0);
    cuda::DeviceArray2D<unsigned short> d; up(d, dists, rows, cols);
    Points p; up(p, points, prows, pcols);
    Projector proj(intr[0], intr[1], intr[2], intr[3]);
    const PtrStepSz<ushort> dd = d;
    project_and_remove(dd, p, proj);
    down(d, dists); down(p, points);
}

RCU_API void refcu_extract_normals(Vol v, const float aff[12], const float Rinv[9], const float* points, unsigned long long n,
                                   float delta_factor, float* out)
{
    Modes m(0, 1);
    cuda::DeviceArray<Point> pts; pts.upload((const Point*)points, (size_t)n);
    cuda::DeviceArray<float4> o((size_t)n);
    PtrSz<Point> ps((Point*)pts.ptr(), (size_t)n);
    extractNormals(make_vol(v), ps, make_aff(aff), make_mat(Rinv), delta_factor, o.ptr());              // :813-830
    o.download((float4*)out);
}

// FullScan6 (tsdf_volume.cu:506-690): warp ballots + warp-synchronous shared-memory scan -> fiber mode.
RCU_API unsigned long long refcu_extract_cloud(Vol v, const float aff[12], float* out, unsigned long long capacity)
{
    Modes m(1, 0);
    // The kernel stores a warp's points BEFORE it tests `full` (tsdf_volume.cu:662-674), so every warp can overrun the
    // buffer once by up to 96 points; the reference never notices because its buffer is 256^3 points.  Slack for that.
    const size_t warps = (size_t)((v.dims[0] + 31) / 32) * ((v.dims[1] + 5) / 6) * 6;
    cuda::DeviceArray<Point> buf((size_t)capacity + warps * 96);
    PtrSz<Point> ps(buf.ptr(), (size_t)capacity);
    size_t n = extractCloud(make_vol(v), make_aff(aff), ps);                                            // :796-811
    std::vector<Point> h;
    buf.download(h);
    memcpy(out, h.data(), (size_t)(n < capacity ? n : capacity) * sizeof(Point));
    return n;
}

// ---- imgproc.cu --------------------------------------------------------------------------------------------------
RCU_API void refcu_compute_dists(const unsigned short* depth, int rows, int cols, const float intr[4], unsigned short* dists)
{
    Modes m(0, 1);
    Depth d; up(d, depth, rows, cols);
    cuda::DeviceArray2D<unsigned short> o(rows, cols);
    compute_dists(d, o, make_float2(intr[0], intr[1]), make_float2(intr[2], intr[3]));                  // imgproc.cu:286-294
    down(o, dists);
}

RCU_API void refcu_bilateral(const unsigned short* src, int rows, int cols, int ksz, float sigma_spatial, float sigma_depth, unsigned short* dst)
{
    Modes m(0, 1);
    Depth s; up(s, src, rows, cols);
    Depth o(rows, cols);
    bilateralFilter(s, o, ksz, sigma_spatial, sigma_depth);                                             // :46-57
    down(o, dst);
}

RCU_API void refcu_cloud_to_depth(const float* cloud, int rows, int cols, unsigned short* depth)
{
    Modes m(0, 1);
    Points p; up(p, cloud, rows, cols);
    Depth d(rows, cols);
    cloud_to_depth(p, d);                                                                                // :296-303 (kernel :273-282)
    down(d, depth);
}

RCU_API void refcu_truncate_depth(unsigned short* depth, int rows, int cols, float max_dist)
{
    Modes m(0, 1);
    Depth d; up(d, depth, rows, cols);
    truncateDepth(d, max_dist);                                                                          // :77-85
    down(d, depth);
}

RCU_API void refcu_depth_pyramid(const unsigned short* src, int rows, int cols, float sigma_depth, unsigned short* dst)
{
    Modes m(0, 1);
    Depth s; up(s, src, rows, cols);
    Depth o(rows / 2, cols / 2);
    depthPyr(s, o, sigma_depth);                                                                         // :126-136
    down(o, dst);
}

RCU_API void refcu_compute_normals_mask_depth(unsigned short* depth, int rows, int cols, const float intr[4], float* normals)
{
    Modes m(0, 1);
    Depth d; up(d, depth, rows, cols);
    Normals n(rows, cols);
    computeNormalsAndMaskDepth(Reprojector(intr[0], intr[1], intr[2], intr[3]), d, n);                   // :190-202
    down(d, depth); down(n, normals);
}

RCU_API void refcu_compute_point_normals(const unsigned short* depth, int rows, int cols, const float intr[4], float* points, float* normals)
{
    Modes m(0, 1);
    Depth d; up(d, depth, rows, cols);
    Points p(rows, cols); Normals n(rows, cols);
    computePointNormals(Reprojector(intr[0], intr[1], intr[2], intr[3]), d, p, n);                       // :242-250
    down(p, points); down(n, normals);
}

RCU_API void refcu_resize_depth_normals(const unsigned short* depth, const float* normals, int rows, int cols, unsigned short* dout, float* nout)
{
    Modes m(0, 1);
    Depth d; up(d, depth, rows, cols);
    Normals n; up(n, normals, rows, cols);
    Depth od(rows / 2, cols / 2); Normals on(rows / 2, cols / 2);
    resizeDepthNormals(d, n, od, on);                                                                    // :348-362
    down(od, dout); down(on, nout);
}

RCU_API void refcu_resize_points_normals(const float* points, const float* normals, int rows, int cols, float* pout, float* nout)
{
    Modes m(0, 1);
    Points p; up(p, points, rows, cols);
    Normals n; up(n, normals, rows, cols);
    Points op(rows / 2, cols / 2); Normals on(rows / 2, cols / 2);
    resizePointsNormals(p, n, op, on);                                                                   // :403-414
    down(op, pout); down(on, nout);
}

RCU_API void refcu_render_points(const float* points, const float* normals, int rows, int cols, const float intr[4], const float light[3], unsigned char* bgra)
{
    Modes m(0, 1);
    Points p; up(p, points, rows, cols);
    Normals n; up(n, normals, rows, cols);
    Image img(rows, cols);
    renderImage(p, n, Reprojector(intr[0], intr[1], intr[2], intr[3]), make_float3(light[0], light[1], light[2]), img);   // :539-546
    down(img, bgra);
}

RCU_API void refcu_render_depth(const unsigned short* depth, const float* normals, int rows, int cols, const float intr[4], const float light[3], unsigned char* bgra)
{
    Modes m(0, 1);
    Depth d; up(d, depth, rows, cols);
    Normals n; up(n, normals, rows, cols);
    Image img(rows, cols);
    renderImage(d, n, Reprojector(intr[0], intr[1], intr[2], intr[3]), make_float3(light[0], light[1], light[2]), img);   // :530-537
    down(img, bgra);
}

RCU_API void refcu_render_tangent_colors(const float* normals, int rows, int cols, unsigned char* bgra)
{
    Modes m(0, 1);
    Normals n; up(n, normals, rows, cols);
    Image img(rows, cols);
    renderTangentColors(n, img);                                                                         // :575-583
    down(img, bgra);
}

// ---- proj_icp.cu -------------------------------------------------------------------------------------------------
// ComputeIcpHelper::operator() (proj_icp.cu:398-444): per-block partial sums (Block::reduce, __syncthreads +
// warp-synchronous tail -> fiber mode with the lock-step patch) and the final 27-value reduction.
RCU_API void refcu_icp_sums_points(const float* vcurr, const float* ncurr, const float* vprev, const float* nprev, int rows, int cols,
                                   const float aff[12], const float intr[4], float dist2_thres, float min_cosine, float out27[27])
{
    Modes m(1, 0);
    Points vc; up(vc, vcurr, rows, cols);
    Normals nc; up(nc, ncurr, rows, cols);
    Points vp; up(vp, vprev, rows, cols);
    Normals np_; up(np_, nprev, rows, cols);
    ComputeIcpHelper helper(0.f, 0.f);                                          // projective_icp.cpp:11-15
    helper.dist2_thres = dist2_thres; helper.min_cosine = min_cosine;           // thresholds handed over already squared / as a cosine
    helper.rows = (float)rows; helper.cols = (float)cols;
    helper.setLevelIntr(0, intr[0], intr[1], intr[2], intr[3]);                 // projective_icp.cpp:17-23
    helper.aff = make_aff(aff);
    helper.vcurr = vc; helper.ncurr = nc;
    cuda::DeviceArray2D<float> buffer;
    helper(vp, np_, buffer, out27, 0);
}
