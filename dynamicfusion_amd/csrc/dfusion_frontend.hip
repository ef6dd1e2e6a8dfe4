// dfusion_frontend.hip -- depth front-end and projective-ICP reduction (gfx950), SURVEY.md 8(f) "next" #3.
//
// Replaces kfusion/src/cuda/imgproc.cu:11-252,309-414 (bilateral, truncation, pyramid, normals, resize) and
// kfusion/src/cuda/proj_icp.cu:30-397 (correspondence search + 27-sum reduction).  Small 2-D kernels: one wave64 covers a
// 64-pixel run of one image row (the reference's warp a 32-pixel run), blocks are 64 x 4.
// Arithmetic follows oracle/dfusion_frontend_oracle.c (see its header for the two hardware-defined operations of the
// reference, __expf and rsqrt, and what stands in for them).
#include "dfusion_internal.h"

#pragma clang fp contract(off)

#define FE_BX 64
#define FE_BY 4
#define FE_GRID(cols, rows) dim3(((cols) + FE_BX - 1) / FE_BX, ((rows) + FE_BY - 1) / FE_BY)
#define FE_XY const int x = blockIdx.x * FE_BX + (threadIdx.x & 63); const int y = blockIdx.y * FE_BY + (threadIdx.x >> 6)

__device__ __forceinline__ uint16_t ld16(const uint16_t* b, size_t pitch, int y, int x) { return *(const uint16_t*)((const char*)b + (size_t)y * pitch + 2 * (size_t)x); }
__device__ __forceinline__ void st16(uint16_t* b, size_t pitch, int y, int x, uint16_t v) { *(uint16_t*)((char*)b + (size_t)y * pitch + 2 * (size_t)x) = v; }
__device__ __forceinline__ float4 ld4(const float* b, size_t pitch, int y, int x) { return *(const float4*)((const char*)b + (size_t)y * pitch + 16 * (size_t)x); }
__device__ __forceinline__ void st4(float* b, size_t pitch, int y, int x, float4 v) { *(float4*)((char*)b + (size_t)y * pitch + 16 * (size_t)x) = v; }

// Reprojector::operator() device.hpp:42-47 / ComputeIcpHelper::reproj proj_icp.cu:39-44
struct FeIntr { float fx, fy, cx, cy, finvx, finvy; };
__device__ __forceinline__ f3 fe_reproj(const FeIntr& I, float u, float v, float z) { return mk3(z * (u - I.cx) * I.finvx, z * (v - I.cy) * I.finvy, z); }
static f3 mk3h(const float v[3]) { f3 r; r.x = v[0]; r.y = v[1]; r.z = v[2]; return r; }
static FeIntr fe_intr(const float intr[4])
{
    FeIntr I; I.fx = intr[0]; I.fy = intr[1]; I.cx = intr[2]; I.cy = intr[3]; I.finvx = 1.f / intr[0]; I.finvy = 1.f / intr[1];   // precomp.cpp:55
    return I;
}

// ------------------------------------------------------------------------------------------ bilateral (imgproc.cu:11-59)
__global__ __launch_bounds__(256) void df_bilateral_kernel(const uint16_t* __restrict__ src, size_t spitch, uint16_t* __restrict__ dst,
                                                           size_t dpitch, int cols, int rows, int ksz, float ss, float sd)
{
    FE_XY;
    if (x >= cols || y >= rows) return;
    const int value = ld16(src, spitch, y, x);
    const int tx = min(x - ksz / 2 + ksz, cols - 1);
    const int ty = min(y - ksz / 2 + ksz, rows - 1);
    float sum1 = 0.f, sum2 = 0.f;
    for (int cy = max(y - ksz / 2, 0); cy < ty; ++cy)
        for (int cx = max(x - ksz / 2, 0); cx < tx; ++cx) {
            const int depth = ld16(src, spitch, cy, cx);
            const float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
            const float color2 = (float)(int)((unsigned)(value - depth) * (unsigned)(value - depth));
            const float weight = (float)exp((double)(-(space2 * ss + color2 * sd)));     // __expf stand-in (oracle header)
            sum1 += (float)depth * weight;
            sum2 += weight;
        }
    st16(dst, dpitch, y, x, (uint16_t)(int)rintf(sum1 / sum2));                          // __float2int_rn
}

extern "C" int dfusion_bilateral_filter(const uint16_t* src, size_t src_pitch, uint16_t* dst, size_t dst_pitch, int cols, int rows,
                                        int kernel_size, float sigma_spatial, float sigma_depth, dfStream stream)
{
    if (!src || !dst || src == dst || cols <= 0 || rows <= 0 || kernel_size <= 0 || !(sigma_spatial > 0.f) || !(sigma_depth > 0.f)) return DF_E_INVALID;
    sigma_depth *= 1000;                                                                 // :50 metres -> mm
    hipLaunchKernelGGL(df_bilateral_kernel, FE_GRID(cols, rows), dim3(256), 0, (hipStream_t)stream, src, src_pitch, dst, dst_pitch, cols,
                       rows, kernel_size, 0.5f / (sigma_spatial * sigma_spatial), 0.5f / (sigma_depth * sigma_depth));   // :56
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ truncation (imgproc.cu:66-85)
__global__ __launch_bounds__(256) void df_truncate_kernel(uint16_t* depth, size_t pitch, int cols, int rows, uint16_t max_mm)
{
    FE_XY;
    if (x < cols && y < rows && ld16(depth, pitch, y, x) > max_mm) st16(depth, pitch, y, x, 0);
}

extern "C" int dfusion_truncate_depth(uint16_t* depth, size_t pitch, int cols, int rows, float max_dist, dfStream stream)
{
    if (!depth || cols <= 0 || rows <= 0 || !(max_dist >= 0.f) || !(max_dist * 1000.f < 65536.f)) return DF_E_INVALID;
    hipLaunchKernelGGL(df_truncate_kernel, FE_GRID(cols, rows), dim3(256), 0, (hipStream_t)stream, depth, pitch, cols, rows,
                       (uint16_t)(max_dist * 1000.f));                                   // :83
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ cloud -> depth (imgproc.cu:273-282, 296-303)
// depth(y, x) = cloud(y, x).z * 1000, converted float -> ushort as the reference's target converts (cvt.rzi.u16.f32: toward zero,
// saturating, NaN -> 0 -- a ray-cast miss becomes "no depth").
__global__ __launch_bounds__(256) void df_cloud_to_depth_kernel(const float* __restrict__ cloud, size_t cpitch, uint16_t* __restrict__ depth, size_t dpitch,
                                                                int cols, int rows)
{
    FE_XY;
    if (x >= cols || y >= rows) return;
    const float mm = ((const float*)((const char*)cloud + (size_t)y * cpitch))[4 * x + 2] * 1000;      // :280
    const unsigned v = mm >= 65535.f ? 65535u : (mm > 0.f ? (unsigned)mm : 0u);                        // (NaN fails both tests: 0)
    st16(depth, dpitch, y, x, (uint16_t)v);
}
extern "C" int dfusion_cloud_to_depth(const float* cloud, size_t cloud_pitch, uint16_t* depth, size_t depth_pitch, int cols, int rows, dfStream stream)
{
    if (!cloud || !depth || cols <= 0 || rows <= 0) return DF_E_INVALID;
    hipLaunchKernelGGL(df_cloud_to_depth_kernel, FE_GRID(cols, rows), dim3(256), 0, (hipStream_t)stream, cloud, cloud_pitch, depth, depth_pitch, cols, rows);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ pyramid (imgproc.cu:94-137)
__global__ __launch_bounds__(256) void df_pyramid_kernel(const uint16_t* __restrict__ src, size_t spitch, int scols, int srows,
                                                         uint16_t* __restrict__ dst, size_t dpitch, int dcols, int drows, float thr)
{
    FE_XY;
    if (x >= dcols || y >= drows) return;
    const int D = 5;
    const int center = ld16(src, spitch, 2 * y, 2 * x);
    const int tx = min(2 * x - D / 2 + D, scols - 1);
    const int ty = min(2 * y - D / 2 + D, srows - 1);
    int sum = 0, count = 0;
    for (int cy = max(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = max(0, 2 * x - D / 2); cx < tx; ++cx) {
            const int val = ld16(src, spitch, cy, cx);
            if ((float)abs(val - center) < thr) { sum += val; ++count; }
        }
    st16(dst, dpitch, y, x, (uint16_t)(count == 0 ? 0 : sum / count));
}

extern "C" int dfusion_depth_pyramid(const uint16_t* src, size_t src_pitch, int src_cols, int src_rows, uint16_t* dst, size_t dst_pitch,
                                     float sigma_depth, dfStream stream)
{
    if (!src || !dst || src_cols < 2 || src_rows < 2) return DF_E_INVALID;
    sigma_depth *= 1000;                                                                 // :130
    const int dc = src_cols / 2, dr = src_rows / 2;                                      // imgproc.cpp:36
    hipLaunchKernelGGL(df_pyramid_kernel, FE_GRID(dc, dr), dim3(256), 0, (hipStream_t)stream, src, src_pitch, src_cols, src_rows, dst,
                       dst_pitch, dc, dr, sigma_depth * 3);                              // :135
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ normals (imgproc.cu:145-252)
__device__ __forceinline__ bool fe_normal_at(const uint16_t* depth, size_t pitch, int cols, int rows, int x, int y, const FeIntr& I, f3* n,
                                             f3* v00)
{
    if (!(x < cols - 1 && y < rows - 1)) return false;
    const float z00 = (float)ld16(depth, pitch, y, x) * 0.001f;
    const float z01 = (float)ld16(depth, pitch, y, x + 1) * 0.001f;
    const float z10 = (float)ld16(depth, pitch, y + 1, x) * 0.001f;
    if (!(z00 * z01 * z10 != 0.f)) return false;
    *v00 = fe_reproj(I, (float)x, (float)y, z00);
    const f3 v01 = fe_reproj(I, (float)(x + 1), (float)y, z01);
    const f3 v10 = fe_reproj(I, (float)x, (float)(y + 1), z10);
    const f3 c = normalized3(cross3(sub3(v01, *v00), sub3(v10, *v00)));
    *n = mk3(-c.x, -c.y, -c.z);
    return true;
}

// MASK: also zero the depth pixel whose normal is NaN (mask_depth_kernel :177-188).  A pixel's normal reads the depth of its
// +x / +y neighbours, which another thread may be zeroing: so the mask is a second launch, like the reference.
__global__ __launch_bounds__(256) void df_normals_kernel(const uint16_t* __restrict__ depth, size_t dpitch, float* __restrict__ normals,
                                                         size_t npitch, int cols, int rows, FeIntr I)
{
    FE_XY;
    if (x >= cols || y >= rows) return;
    const float q = qnanf_();
    f3 n, v;
    if (fe_normal_at(depth, dpitch, cols, rows, x, y, I, &n, &v)) st4(normals, npitch, y, x, make_float4(n.x, n.y, n.z, 0.f));
    else st4(normals, npitch, y, x, make_float4(q, q, q, 0.f));
}
__global__ __launch_bounds__(256) void df_mask_depth_kernel(const float* __restrict__ normals, size_t npitch, uint16_t* depth, size_t dpitch,
                                                            int cols, int rows)
{
    FE_XY;
    if (x < cols && y < rows && isnan(ld4(normals, npitch, y, x).x)) st16(depth, dpitch, y, x, 0);
}

extern "C" int dfusion_compute_normals_mask_depth(uint16_t* depth, size_t depth_pitch, float* normals, size_t normals_pitch, int cols,
                                                  int rows, const float intr[4], dfStream stream)
{
    if (!depth || !normals || !intr || cols <= 0 || rows <= 0) return DF_E_INVALID;
    hipLaunchKernelGGL(df_normals_kernel, FE_GRID(cols, rows), dim3(256), 0, (hipStream_t)stream, depth, depth_pitch, normals, normals_pitch,
                       cols, rows, fe_intr(intr));
    DF_LAUNCH_CHECK();
    hipLaunchKernelGGL(df_mask_depth_kernel, FE_GRID(cols, rows), dim3(256), 0, (hipStream_t)stream, normals, normals_pitch, depth,
                       depth_pitch, cols, rows);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

__global__ __launch_bounds__(256) void df_point_normals_kernel(const uint16_t* __restrict__ depth, size_t dpitch, float* __restrict__ points,
                                                               size_t ppitch, float* __restrict__ normals, size_t npitch, int cols, int rows,
                                                               FeIntr I)
{
    FE_XY;
    if (x >= cols || y >= rows) return;
    const float q = qnanf_();
    f3 n, v;
    if (fe_normal_at(depth, dpitch, cols, rows, x, y, I, &n, &v)) {
        st4(normals, npitch, y, x, make_float4(n.x, n.y, n.z, 0.f));
        st4(points, ppitch, y, x, make_float4(v.x, v.y, v.z, 0.f));
    } else {
        st4(normals, npitch, y, x, make_float4(q, q, q, q));                             // :220
        st4(points, ppitch, y, x, make_float4(q, q, q, q));
    }
}

extern "C" int dfusion_compute_point_normals(const uint16_t* depth, size_t depth_pitch, float* points, size_t points_pitch, float* normals,
                                             size_t normals_pitch, int cols, int rows, const float intr[4], dfStream stream)
{
    if (!depth || !points || !normals || !intr || cols <= 0 || rows <= 0) return DF_E_INVALID;
    hipLaunchKernelGGL(df_point_normals_kernel, FE_GRID(cols, rows), dim3(256), 0, (hipStream_t)stream, depth, depth_pitch, points,
                       points_pitch, normals, normals_pitch, cols, rows, fe_intr(intr));
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ resize (imgproc.cu:309-414)
__global__ __launch_bounds__(256) void df_resize_depth_normals_kernel(const uint16_t* __restrict__ dsrc, size_t dspitch,
                                                                      const float* __restrict__ nsrc, size_t nspitch,
                                                                      uint16_t* __restrict__ ddst, size_t ddpitch, float* __restrict__ ndst,
                                                                      size_t ndpitch, int dcols, int drows)
{
    FE_XY;
    if (x >= dcols || y >= drows) return;
    const float q = qnanf_();
    const int xs = 2 * x, ys = 2 * y;
    const int d00 = ld16(dsrc, dspitch, ys, xs), d01 = ld16(dsrc, dspitch, ys, xs + 1);
    const int d10 = ld16(dsrc, dspitch, ys + 1, xs), d11 = ld16(dsrc, dspitch, ys + 1, xs + 1);
    uint16_t d = 0;
    float4 n = make_float4(q, q, q, q);
    if ((int)((unsigned)d00 * (unsigned)d01) != 0 && (int)((unsigned)d10 * (unsigned)d11) != 0) {
        d = (uint16_t)((d00 + d01 + d10 + d11) / 4);
        const float4 a = ld4(nsrc, nspitch, ys, xs), b = ld4(nsrc, nspitch, ys, xs + 1);
        const float4 c = ld4(nsrc, nspitch, ys + 1, xs), e = ld4(nsrc, nspitch, ys + 1, xs + 1);
        n.x = (((a.x + b.x) + c.x) + e.x) * 0.25f;                                       // :343-345 (x0.25 is exact in either precision)
        n.y = (((a.y + b.y) + c.y) + e.y) * 0.25f;
        n.z = (((a.z + b.z) + c.z) + e.z) * 0.25f;
    }
    st16(ddst, ddpitch, y, x, d);
    st4(ndst, ndpitch, y, x, n);
}

extern "C" int dfusion_resize_depth_normals(const uint16_t* depth, size_t depth_pitch, const float* normals, size_t normals_pitch,
                                            int src_cols, int src_rows, uint16_t* depth_out, size_t depth_out_pitch, float* normals_out,
                                            size_t normals_out_pitch, dfStream stream)
{
    if (!depth || !normals || !depth_out || !normals_out || src_cols < 2 || src_rows < 2) return DF_E_INVALID;
    const int dc = src_cols / 2, dr = src_rows / 2;
    hipLaunchKernelGGL(df_resize_depth_normals_kernel, FE_GRID(dc, dr), dim3(256), 0, (hipStream_t)stream, depth, depth_pitch, normals,
                       normals_pitch, depth_out, depth_out_pitch, normals_out, normals_out_pitch, dc, dr);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

__global__ __launch_bounds__(256) void df_resize_points_normals_kernel(const float* __restrict__ vsrc, size_t vspitch,
                                                                       const float* __restrict__ nsrc, size_t nspitch,
                                                                       float* __restrict__ vdst, size_t vdpitch, float* __restrict__ ndst,
                                                                       size_t ndpitch, int dcols, int drows)
{
    FE_XY;
    if (x >= dcols || y >= drows) return;
    const float q = qnanf_();
    const int xs = 2 * x, ys = 2 * y;
    float4 vo = make_float4(q, q, q, 0.f), no = make_float4(q, q, q, 0.f);
    const float4 a = ld4(vsrc, vspitch, ys, xs), b = ld4(vsrc, vspitch, ys, xs + 1);
    const float4 c = ld4(vsrc, vspitch, ys + 1, xs), e = ld4(vsrc, vspitch, ys + 1, xs + 1);
    if (!isnan(a.x * b.x * c.x * e.x)) {
        vo.x = (((a.x + b.x) + c.x) + e.x) * 0.25f; vo.y = (((a.y + b.y) + c.y) + e.y) * 0.25f; vo.z = (((a.z + b.z) + c.z) + e.z) * 0.25f;
        const float4 na = ld4(nsrc, nspitch, ys, xs), nb = ld4(nsrc, nspitch, ys, xs + 1);
        const float4 nc = ld4(nsrc, nspitch, ys + 1, xs), ne = ld4(nsrc, nspitch, ys + 1, xs + 1);
        no.x = (((na.x + nb.x) + nc.x) + ne.x) * 0.25f; no.y = (((na.y + nb.y) + nc.y) + ne.y) * 0.25f; no.z = (((na.z + nb.z) + nc.z) + ne.z) * 0.25f;
    }
    st4(vdst, vdpitch, y, x, vo);
    st4(ndst, ndpitch, y, x, no);
}

extern "C" int dfusion_resize_points_normals(const float* points, size_t points_pitch, const float* normals, size_t normals_pitch,
                                             int src_cols, int src_rows, float* points_out, size_t points_out_pitch, float* normals_out,
                                             size_t normals_out_pitch, dfStream stream)
{
    if (!points || !normals || !points_out || !normals_out || src_cols < 2 || src_rows < 2) return DF_E_INVALID;
    const int dc = src_cols / 2, dr = src_rows / 2;
    hipLaunchKernelGGL(df_resize_points_normals_kernel, FE_GRID(dc, dr), dim3(256), 0, (hipStream_t)stream, points, points_pitch, normals,
                       normals_pitch, points_out, points_out_pitch, normals_out, normals_out_pitch, dc, dr);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ renderImage / renderTangentColors
// device::renderImage (x2) and renderTangentColors (imgproc.cu:420-583): the Phong view KinFu::renderImage hands to the demo.
// One lane per pixel, 4-byte BGRA stores.  __powf(x, 20) -> (float)pow((double)x, 20.0) and rsqrt -> 1 / sqrtf on both sides of
// the parity tests (oracle/dfusion_frontend_oracle.c, and the reference's own kernel through oracle/cuda_shim).
__device__ __forceinline__ float fe_saturate(float a) { return a != a ? 0.f : fminf(fmaxf(a, 0.f), 1.f); }
__device__ __forceinline__ uint32_t fe_shade(bool have, f3 P, f3 N, f3 light, int y, int rows)
{
    f3 color;
    if (!have) {
        const f3 bgr1 = mk3(4.f / 255.f, 2.f / 255.f, 2.f / 255.f), bgr2 = mk3(236.f / 255.f, 120.f / 255.f, 120.f / 255.f);
        const float w = (float)y / (float)rows;
        color = add3(scale3(bgr1, 1.f - w), scale3(bgr2, w));                                       // :439-444
    } else {
        const f3 L = normalized3(sub3(light, P));
        const f3 V = normalized3(sub3(mk3(0.f, 0.f, 0.f), P));
        const f3 R = normalized3(sub3(scale3(scale3(N, 2.f), dot3(N, L)), L));                     // 2 * N * dot(N, L) - L
        // Ix = Ax*Ka*Dx + Lx*Kd*Dx * max(0, N.L) + Lx*Ks*Sx * max(0, R.V)^20 with Ax = Dx = Sx = Lx = 1, Ka = .3, Kd = .5, Ks = .2
        const float Ix = 0.3f + 0.5f * fmaxf(0.f, dot3(N, L)) + 0.2f * (float)pow((double)fmaxf(0.f, dot3(R, V)), 20.0);
        color = mk3(Ix, Ix, Ix);
    }
    const uint32_t b = (uint32_t)(fe_saturate(color.x) * 255.f), g = (uint32_t)(fe_saturate(color.y) * 255.f), r = (uint32_t)(fe_saturate(color.z) * 255.f);
    return b | (g << 8) | (r << 16);
}
template <bool DEPTH>
__global__ __launch_bounds__(256) void df_render_kernel(const void* __restrict__ src, size_t spitch, const float* __restrict__ normals,
                                                        size_t npitch, int cols, int rows, FeIntr I, f3 light, uint32_t* __restrict__ image,
                                                        size_t ipitch)
{
    FE_XY;
    if (x >= cols || y >= rows) return;
    const float4 n = ld4(normals, npitch, y, x);
    bool have; f3 P;
    if (DEPTH) {
        const int d = *(const uint16_t*)((const char*)src + (size_t)y * spitch + 2 * (size_t)x);
        have = d != 0;
        const float z = (float)d * 0.001f;
        P = mk3(z * ((float)x - I.cx) * I.finvx, z * ((float)y - I.cy) * I.finvy, z);              // Reprojector, device.hpp:42-48
    } else {
        const float4 p = ld4((const float*)src, spitch, y, x);
        have = !isnan(p.x);
        P = mk3(p.x, p.y, p.z);
    }
    *(uint32_t*)((char*)image + (size_t)y * ipitch + 4 * (size_t)x) = fe_shade(have, P, mk3(n.x, n.y, n.z), light, y, rows);
}
__global__ __launch_bounds__(256) void df_tangent_colors_kernel(const float* __restrict__ normals, size_t npitch, int cols, int rows,
                                                                uint32_t* __restrict__ image, size_t ipitch)
{
    FE_XY;
    if (x >= cols || y >= rows) return;
    const float4 n = ld4(normals, npitch, y, x);
    // (unsigned char)(float) of the reference: truncation; saturating / NaN -> 0 here as in the GPU conversion it compiles to
    const float r = (5.f - n.x * 3.5f) * 25.5f, g = (5.f - n.y * 2.5f) * 25.5f, b = (5.f - n.z * 3.5f) * 25.5f;
    const uint32_t rb = (uint32_t)(r != r ? 0.f : fminf(fmaxf(r, 0.f), 255.f)), gb = (uint32_t)(g != g ? 0.f : fminf(fmaxf(g, 0.f), 255.f)),
                   bb = (uint32_t)(b != b ? 0.f : fminf(fmaxf(b, 0.f), 255.f));
    *(uint32_t*)((char*)image + (size_t)y * ipitch + 4 * (size_t)x) = bb | (gb << 8) | (rb << 16);  // make_uchar4(b, g, r, 0)
}

extern "C" int dfusion_render_image_points(const float* points, size_t points_pitch, const float* normals, size_t normals_pitch, int cols,
                                           int rows, const float light_pose[3], unsigned char* image, size_t image_pitch, dfStream stream)
{
    if (!points || !normals || !light_pose || !image || cols <= 0 || rows <= 0) return DF_E_INVALID;
    FeIntr I; I.fx = I.fy = 1.f; I.cx = I.cy = 0.f; I.finvx = I.finvy = 1.f;
    hipLaunchKernelGGL((df_render_kernel<false>), FE_GRID(cols, rows), dim3(256), 0, (hipStream_t)stream, (const void*)points, points_pitch, normals,
                       normals_pitch, cols, rows, I, mk3h(light_pose), (uint32_t*)image, image_pitch);
    DF_LAUNCH_CHECK();
    return DF_OK;
}
extern "C" int dfusion_render_image_depth(const uint16_t* depth, size_t depth_pitch, const float* normals, size_t normals_pitch, int cols,
                                          int rows, const float intr[4], const float light_pose[3], unsigned char* image, size_t image_pitch,
                                          dfStream stream)
{
    if (!depth || !normals || !intr || !light_pose || !image || cols <= 0 || rows <= 0) return DF_E_INVALID;
    hipLaunchKernelGGL((df_render_kernel<true>), FE_GRID(cols, rows), dim3(256), 0, (hipStream_t)stream, (const void*)depth, depth_pitch, normals,
                       normals_pitch, cols, rows, fe_intr(intr), mk3h(light_pose), (uint32_t*)image, image_pitch);
    DF_LAUNCH_CHECK();
    return DF_OK;
}
extern "C" int dfusion_render_tangent_colors(const float* normals, size_t normals_pitch, int cols, int rows, unsigned char* image,
                                             size_t image_pitch, dfStream stream)
{
    if (!normals || !image || cols <= 0 || rows <= 0) return DF_E_INVALID;
    hipLaunchKernelGGL(df_tangent_colors_kernel, FE_GRID(cols, rows), dim3(256), 0, (hipStream_t)stream, normals, normals_pitch, cols, rows,
                       (uint32_t*)image, image_pitch);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ point-set glue (kinfu.cpp:353-383)
// The host loops of KinFu::dynamicfusion that turn the ray-cast image into `canonical` (inverse_pose * point) and strip the
// float4 padding, kept on the device.  Arithmetic of cv::Affine3f * Vec3f as the host mirror evaluates it
// (host/include/kfusion/types.hpp): R(i,0)*x + R(i,1)*y + R(i,2)*z + t(i), plain float products summed left to right.
__global__ __launch_bounds__(256) void df_transform_points_kernel(const float* __restrict__ in, size_t in_pitch, int in_stride,
                                                                 float* __restrict__ out, size_t out_pitch, int out_stride, int cols, int rows,
                                                                 DfAff A, int use_aff)
{
    FE_XY;
    if (x >= cols || y >= rows) return;
    const float* p = (const float*)((const char*)in + (size_t)y * in_pitch) + (size_t)x * in_stride;
    float* o = (float*)((char*)out + (size_t)y * out_pitch) + (size_t)x * out_stride;
    const float px = p[0], py = p[1], pz = p[2];
    float rx = px, ry = py, rz = pz;
    if (use_aff) {
        rx = A.R[0] * px + A.R[1] * py + A.R[2] * pz + A.t[0];
        ry = A.R[3] * px + A.R[4] * py + A.R[5] * pz + A.t[1];
        rz = A.R[6] * px + A.R[7] * py + A.R[8] * pz + A.t[2];
    }
    o[0] = rx; o[1] = ry; o[2] = rz;
    if (out_stride > 3) o[3] = 0.f;
}

extern "C" int dfusion_transform_points(const float* in, size_t in_pitch, int in_stride, float* out, size_t out_pitch, int out_stride,
                                        int cols, int rows, const float aff[12], dfStream stream)
{
    if (!in || !out || in == out || cols <= 0 || rows <= 0 || in_stride < 3 || out_stride < 3 || out_stride > 4) return DF_E_INVALID;
    DfAff A;
    memset(&A, 0, sizeof(A));
    if (aff) A = df_aff(aff);
    hipLaunchKernelGGL(df_transform_points_kernel, FE_GRID(cols, rows), dim3(256), 0, (hipStream_t)stream, in, in_pitch, in_stride, out,
                       out_pitch, out_stride, cols, rows, A, aff ? 1 : 0);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ projective ICP (proj_icp.cu:30-397)
struct DfIcpArgs {
    int cols, rows;
    DfAff aff;
    FeIntr I;
    float min_cosine, dist2_thres;
    const float* vcurr; size_t vcpitch; const float* ncurr; size_t ncpitch;
    const float* vprev; size_t vppitch; const float* nprev; size_t nppitch;
    const uint16_t* dcurr; size_t dcpitch; const uint16_t* dprev; size_t dppitch;
    float* partial; int partials;      // [27][partials]
    int* accepted;                     // nullable
    const float* state;                // nullable: device-resident estimate {affine[12], ok} (dfusion_icp_estimate); overrides `aff`
};

// find_coresp, :47-110.  DEPTH selects the USE_DEPTH build's variant.
template <bool DEPTH>
__device__ __forceinline__ int fe_find_coresp(const DfIcpArgs& A, int x, int y, f3* nd, f3* d, f3* s)
{
    if constexpr (DEPTH) {
        const int src_z = ld16(A.dcurr, A.dcpitch, y, x);
        if (src_z == 0) return 40;
        *s = aff_mul(A.aff, fe_reproj(A.I, (float)x, (float)y, (float)src_z * 0.001f));
    } else {
        const float4 p = ld4(A.vcurr, A.vcpitch, y, x);
        if (isnan(p.x)) return 40;
        *s = aff_mul(A.aff, mk3(p.x, p.y, p.z));
    }
    const float u = fmaf(A.I.fx, s->x / s->z, A.I.cx);                                   // :33-34
    const float v = fmaf(A.I.fy, s->y / s->z, A.I.cy);
    if (s->z <= 0.f || !(u >= 0.f && v >= 0.f && u < (float)A.cols && v < (float)A.rows)) return 80;   // (+NaN => outside)
    const int ui = (int)u, vi = (int)v;
    if constexpr (DEPTH) {
        const int dst_z = ld16(A.dprev, A.dppitch, vi, ui);
        if (dst_z == 0) return 120;
        *d = fe_reproj(A.I, u, v, (float)dst_z * 0.001f);
    } else {
        const float4 q = ld4(A.vprev, A.vppitch, vi, ui);
        if (isnan(q.x)) return 120;
        *d = mk3(q.x, q.y, q.z);
    }
    const f3 diff = sub3(*s, *d);
    if (dot3(diff, diff) > A.dist2_thres) return 160;
    const float4 nc = ld4(A.ncurr, A.ncpitch, y, x);
    const f3 ns = mat3_mul(A.aff.R, mk3(nc.x, nc.y, nc.z));
    const float4 np = ld4(A.nprev, A.nppitch, vi, ui);
    *nd = mk3(np.x, np.y, np.z);
    if (fabsf(dot3(ns, *nd)) < A.min_cosine) return 200;
    return 0;
}

// Block::reduce<256> (temp_utils.hpp:503-523) for NV values per thread at once.  Only thread 0's result is defined, and it is
// the reference's tree: pairs (t, t+s) for s = 128, 64 (through LDS: wave w+2 -> w, wave 1 -> 0, same lane), then s = 32..1
// inside wave 0 (lane t+s -> lane t).  Float addition is commutative, so op(val, other) order is immaterial.
template <int NV>
__device__ __forceinline__ void fe_tree256(float (&v)[NV], float* lds /* [NV][128] */)
{
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    if (w >= 2) {
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[k * 128 + (t - 128)] = v[k];
    }
    __syncthreads();
    if (w < 2) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = v[k] + lds[k * 128 + t];
    }
    __syncthreads();
    if (w == 1) {
#pragma unroll
        for (int k = 0; k < NV; ++k) lds[k * 128 + l] = v[k];
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            float a = v[k] + lds[k * 128 + l];
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) a = a + __shfl_down(a, s, 64);
            v[k] = a;
        }
    }
}

// icp_helper_kernel (:350-371) + partial_reduce (:112-348): the block is the reference's 32 x 8 pixel tile (the partial sums,
// hence the final float sums, depend on which pixels share a block), tid = ty * 32 + tx.
template <bool DEPTH>
__global__ __launch_bounds__(256) void df_icp_partial_kernel(const DfIcpArgs A)
{
    __shared__ float lds[27 * 128];
    const int t = threadIdx.x;
    const int x = (t & 31) + blockIdx.x * 32, y = (t >> 5) + blockIdx.y * 8;
    f3 n, d, s;
    float row[7];
    DfIcpArgs B = A;
    if (A.state) {                                                                       // estimate kept on the device between iterations
#pragma unroll
        for (int i = 0; i < 9; ++i) B.aff.R[i] = A.state[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) B.aff.t[i] = A.state[9 + i];
    }
    const int filtered = (x < A.cols && y < A.rows) ? fe_find_coresp<DEPTH>(B, x, y, &n, &d, &s) : 1;
    if (!filtered) {
        const f3 c = cross3(s, n);
        row[0] = c.x; row[1] = c.y; row[2] = c.z; row[3] = n.x; row[4] = n.y; row[5] = n.z;
        row[6] = dot3(n, sub3(d, s));
    } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) row[i] = 0.f;
    }
    float v[27];
    {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 7; ++j) v[k++] = row[i] * row[j];
    }
    fe_tree256<27>(v, lds);
    if (t == 0) {
        const int pos = blockIdx.x + gridDim.x * blockIdx.y;                             // :117
#pragma unroll
        for (int k = 0; k < 27; ++k) A.partial[(size_t)k * A.partials + pos] = v[k];
    }
    if (A.accepted) {
        const unsigned long long m = __ballot(!filtered);
        if ((t & 63) == 0 && m) atomicAdd(A.accepted, __popcll(m));
    }
}

// icp_final_reduce_kernel (:373-397): block k sums row k of the partials -- strided serial sums, then the same tree.
__global__ __launch_bounds__(256) void df_icp_final_kernel(const float* __restrict__ partial, int partials, float* __restrict__ out)
{
    __shared__ float lds[128];
    const float* beg = partial + (size_t)blockIdx.x * partials;
    float v[1];
    float sum = 0.f;
    for (int i = threadIdx.x; i < partials; i += 256) sum += beg[i];
    v[0] = sum;
    fe_tree256<1>(v, lds);
    if (threadIdx.x == 0) out[blockIdx.x] = v[0];
}

extern "C" int dfusion_icp_workspace_floats(int cols, int rows)
{
    if (cols <= 0 || rows <= 0) return 0;
    return 27 * (((cols + 31) / 32) * ((rows + 7) / 8));
}

static int df_icp_launch(DfIcpArgs& A, bool depth, const float aff[12], const float intr[4], float dist2_thres, float min_cosine,
                         float* workspace, float* sums, int* accepted, hipStream_t st)
{
    if (!aff || !intr || !workspace || !sums || A.cols <= 0 || A.rows <= 0) return DF_E_INVALID;
    A.aff = df_aff(aff); A.I = fe_intr(intr); A.min_cosine = min_cosine; A.dist2_thres = dist2_thres;
    const dim3 grid((A.cols + 31) / 32, (A.rows + 7) / 8);                              // :406-407
    A.partial = workspace; A.partials = (int)(grid.x * grid.y); A.accepted = accepted;
    if (depth) hipLaunchKernelGGL(df_icp_partial_kernel<true>, grid, dim3(256), 0, st, A);
    else hipLaunchKernelGGL(df_icp_partial_kernel<false>, grid, dim3(256), 0, st, A);
    DF_LAUNCH_CHECK();
    hipLaunchKernelGGL(df_icp_final_kernel, dim3(27), dim3(256), 0, st, workspace, A.partials, sums);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

extern "C" int dfusion_icp_sums_points(const float* vcurr, size_t vcurr_pitch, const float* ncurr, size_t ncurr_pitch, const float* vprev,
                                       size_t vprev_pitch, const float* nprev, size_t nprev_pitch, int cols, int rows, const float aff[12],
                                       const float intr[4], float dist2_thres, float min_cosine, float* workspace, float* sums,
                                       int* accepted, dfStream stream)
{
    if (!vcurr || !ncurr || !vprev || !nprev) return DF_E_INVALID;
    DfIcpArgs A;
    memset(&A, 0, sizeof(A));
    A.cols = cols; A.rows = rows;
    A.vcurr = vcurr; A.vcpitch = vcurr_pitch; A.ncurr = ncurr; A.ncpitch = ncurr_pitch;
    A.vprev = vprev; A.vppitch = vprev_pitch; A.nprev = nprev; A.nppitch = nprev_pitch;
    return df_icp_launch(A, false, aff, intr, dist2_thres, min_cosine, workspace, sums, accepted, (hipStream_t)stream);
}

extern "C" int dfusion_icp_sums_depth(const uint16_t* dcurr, size_t dcurr_pitch, const float* ncurr, size_t ncurr_pitch,
                                      const uint16_t* dprev, size_t dprev_pitch, const float* nprev, size_t nprev_pitch, int cols, int rows,
                                      const float aff[12], const float intr[4], float dist2_thres, float min_cosine, float* workspace,
                                      float* sums, int* accepted, dfStream stream)
{
    if (!dcurr || !ncurr || !dprev || !nprev) return DF_E_INVALID;
    DfIcpArgs A;
    memset(&A, 0, sizeof(A));
    A.cols = cols; A.rows = rows;
    A.dcurr = dcurr; A.dcpitch = dcurr_pitch; A.ncurr = ncurr; A.ncpitch = ncurr_pitch;
    A.dprev = dprev; A.dppitch = dprev_pitch; A.nprev = nprev; A.nppitch = nprev_pitch;
    return df_icp_launch(A, true, aff, intr, dist2_thres, min_cosine, workspace, sums, accepted, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ the whole ICP loop on the device
// projective_icp.cpp:129-213 without the per-iteration round trip (cudaMemcpyAsync + cudaStreamSynchronize + cv::solve on the host,
// :45,:150-163): after the two reduction kernels a one-thread kernel unpacks the 27 sums (StreamHelper::get), solves the 6x6 system
// (LU with partial pivoting in double -- the arithmetic of the host mirror's solve6), applies the determinant test, builds Tinc by
// Rodrigues and composes it into the estimate that lives in device memory; the next iteration's kernels read it from there.
// state = {affine R[9], t[3], ok (1 / 0)}.  Once ok == 0 (|det| < 1e-15 or NaN, :152-156) the estimate is frozen.
__global__ __launch_bounds__(64) void df_icp_solve_kernel(const float* __restrict__ sums, float* __restrict__ state)
{
    // One wave.  The augmented matrix lives in LDS (the pivot search indexes it dynamically); lane (i, j) = (lane / 8, lane % 8)
    // owns element M[i][j] during the elimination, so a pivot step is one parallel update instead of up to 35 dependent LDS
    // round trips -- each element still sees exactly the operations of the serial loop (f = M[i][c] / M[c][c]; M[i][j] -= f * M[c][j]).
    __shared__ double M[6][7];
    __shared__ int s_flag[2];                                  // [0] pivot row, [1] 1 = singular
    __shared__ double s_det;
    const int lane = threadIdx.x, li = lane >> 3, lj = lane & 7;
    if (state[12] == 0.f) return;                              // uniform: frozen estimate
    if (lane == 0) {
        int shift = 0;
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 7; ++j) {
                const double value = (double)sums[shift++];
                if (j == 6) M[i][6] = value; else { M[i][j] = value; M[j][i] = value; }
            }
        s_det = 1.0; s_flag[1] = 0;
    }
    __syncthreads();
    for (int c = 0; c < 6; ++c) {
        if (lane == 0) {
            int p = c;
            for (int i = c + 1; i < 6; ++i) if (fabs(M[i][c]) > fabs(M[p][c])) p = i;
            if (M[p][c] == 0.0 || M[p][c] != M[p][c]) { s_det = (M[p][c] != M[p][c]) ? M[p][c] : 0.0; s_flag[1] = 1; }
            s_flag[0] = p;
        }
        __syncthreads();
        if (s_flag[1]) break;                                  // uniform
        const int p = s_flag[0];
        double a_p = 0.0, a_c = 0.0;
        if (p != c && lane < 7) { a_p = M[p][lane]; a_c = M[c][lane]; }
        __syncthreads();
        if (p != c && lane < 7) { M[p][lane] = a_c; M[c][lane] = a_p; }
        __syncthreads();
        if (lane == 0) { if (p != c) s_det = -s_det; s_det *= M[c][c]; }
        const bool mine = li > c && li < 6 && lj >= c && lj < 7;
        double f = 0.0, mcj = 0.0, mij = 0.0;
        if (mine) { f = M[li][c] / M[c][c]; mcj = M[c][lj]; mij = M[li][lj]; }
        __syncthreads();
        if (mine) M[li][lj] = mij - f * mcj;
        __syncthreads();
    }
    if (lane != 0) return;
    const bool singular = s_flag[1] != 0;
    const double det = s_det;
    if (singular || fabs(det) < 1e-15 || det != det) { state[12] = 0.f; return; }
    float r[6];
    for (int i = 5; i >= 0; --i) {
        double sacc = M[i][6];
        for (int j = i + 1; j < 6; ++j) sacc -= M[i][j] * (double)r[j];
        r[i] = (float)(sacc / M[i][i]);
    }
    // Tinc = Affine3f(rvec = r[0..2], t = r[3..5]) (Rodrigues, host/include/kfusion/types.hpp), affine <- Tinc * affine
    float TR[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
    {
        const double rx = r[0], ry = r[1], rz = r[2];
        const double theta = sqrt(rx * rx + ry * ry + rz * rz);
        if (theta >= 2.220446049250313e-16) {
            const double c = cos(theta), sn = sin(theta), c1 = 1. - c, it = 1. / theta;
            const double k[3] = {rx * it, ry * it, rz * it};
            const double Kx[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    TR[3 * i + j] = (float)(c * (i == j ? 1. : 0.) + c1 * k[i] * k[j] + sn * Kx[3 * i + j]);
        }
    }
    float R[9], tt[3];
    for (int i = 0; i < 9; ++i) R[i] = state[i];
    for (int i = 0; i < 3; ++i) tt[i] = state[9 + i];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            state[3 * i + j] = (float)((double)TR[3 * i] * R[j] + (double)TR[3 * i + 1] * R[3 + j] + (double)TR[3 * i + 2] * R[6 + j]);
        state[9 + i] = (float)((double)TR[3 * i] * tt[0] + (double)TR[3 * i + 1] * tt[1] + (double)TR[3 * i + 2] * tt[2] + (double)r[3 + i]);
    }
}

__global__ void df_icp_state_init_kernel(float* state)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        for (int i = 0; i < 12; ++i) state[i] = (i == 0 || i == 4 || i == 8) ? 1.f : 0.f;
        state[12] = 1.f;
    }
}

extern "C" int dfusion_icp_estimate(const DfIcpLevel* levels, int n_levels, int depth_variant, const float intr[4], float dist2_thres,
                                    float min_cosine, float* workspace, float* state, dfStream stream)
{
    if (!levels || n_levels <= 0 || n_levels > 8 || !intr || !workspace || !state) return DF_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(df_icp_state_init_kernel, dim3(1), dim3(64), 0, st, state);
    DF_LAUNCH_CHECK();
    for (int level = n_levels - 1; level >= 0; --level) {                                // coarse to fine, :135
        const DfIcpLevel& L = levels[level];
        if (L.iters <= 0) continue;
        if (!L.curr || !L.ncurr || !L.prev || !L.nprev || L.cols <= 0 || L.rows <= 0) return DF_E_INVALID;
        const int div = 1 << level;                                                      // setLevelIntr, :17-23
        const float li[4] = {intr[0] / div, intr[1] / div, intr[2] / div, intr[3] / div};
        DfIcpArgs A;
        memset(&A, 0, sizeof(A));
        A.cols = L.cols; A.rows = L.rows;
        if (depth_variant) { A.dcurr = (const uint16_t*)L.curr; A.dcpitch = L.curr_pitch; A.dprev = (const uint16_t*)L.prev; A.dppitch = L.prev_pitch; }
        else { A.vcurr = (const float*)L.curr; A.vcpitch = L.curr_pitch; A.vprev = (const float*)L.prev; A.vppitch = L.prev_pitch; }
        A.ncurr = L.ncurr; A.ncpitch = L.ncurr_pitch; A.nprev = L.nprev; A.nppitch = L.nprev_pitch;
        A.I = fe_intr(li); A.min_cosine = min_cosine; A.dist2_thres = dist2_thres;
        const dim3 grid((A.cols + 31) / 32, (A.rows + 7) / 8);
        A.partials = (int)(grid.x * grid.y);
        A.partial = workspace; A.state = state;
        float* sums = workspace + (size_t)27 * A.partials;
        for (int it = 0; it < L.iters; ++it) {
            if (depth_variant) hipLaunchKernelGGL(df_icp_partial_kernel<true>, grid, dim3(256), 0, st, A);
            else hipLaunchKernelGGL(df_icp_partial_kernel<false>, grid, dim3(256), 0, st, A);
            hipLaunchKernelGGL(df_icp_final_kernel, dim3(27), dim3(256), 0, st, workspace, A.partials, sums);
            hipLaunchKernelGGL(df_icp_solve_kernel, dim3(1), dim3(64), 0, st, sums, state);
            DF_LAUNCH_CHECK();
        }
    }
    return DF_OK;
}
