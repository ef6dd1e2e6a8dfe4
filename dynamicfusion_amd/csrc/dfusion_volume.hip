// dfusion_volume.hip -- clear, compute_dists, rigid integrate (gfx950).
//
// Replaces /root/reference/kfusion/src/cuda/tsdf_volume.cu:15-41 (clear), :51-112,141-161
// (TsdfIntegrator / integrate) and kfusion/src/cuda/imgproc.cu:259-294 (compute_dists).
//
// MI355X design notes
//   * The reference runs one thread per (x,y) column marching all Z planes with 4-byte accesses
//     (a 32x8 block).  Here a lane owns FOUR x-adjacent columns, so a wave64 touches 1 KiB of
//     contiguous voxels per plane with one global_load_dwordx4 / global_store_dwordx4, and the
//     Z range is cut into chunks (blockIdx.y) so the 256 CUs see >= 4 waves per SIMD.
//   * Bit parity with the unsharded sweep: vc is the running sum vc += zstep from z = 0
//     (tsdf_volume.cu:75).  A chunk (or a Z-slab on another GPU) starting at plane z0 replays the
//     first z0 additions in registers - 3 adds per plane per column, no memory traffic.
//   * Voxels are read/written only when their update branch is taken (tsdf_volume.cu:91), exactly
//     the traffic the algorithmic-bytes figure 8*N_upd counts.
#include "dfusion_internal.h"
#include "dfusion_pyramid.h"
#include <math.h>
#include <stdlib.h>

// ------------------------------------------------------------------------------------------ clear
__global__ __launch_bounds__(256) void df_fill_zero_kernel(uint4* __restrict__ p, size_t n16)
{
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);       // pack_tsdf(0.f, 0) == 0 (device.hpp:53-54)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        p[i] = z;
}

extern "C" int dfusion_clear(DfVolume v, const DfSlab* slab, dfStream stream)
{
    if (!df_volume_valid(v)) return DF_E_INVALID;
    DfSlab s = df_slab_or_full(v, slab);
    if (!df_slab_valid(v, s)) return DF_E_INVALID;
    size_t n16 = (size_t)v.dims[0] * v.dims[1] * s.z_store_n / 4;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;         // grid-stride: 16 blocks per CU
    hipLaunchKernelGGL(df_fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)v.data, n16);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ copy probe
__global__ __launch_bounds__(256) void df_copy_kernel(uint4* __restrict__ d, const uint4* __restrict__ s, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        d[i] = s[i];
}

extern "C" int dfusion_copy_bandwidth_probe(void* dst, const void* src, size_t bytes, dfStream stream)
{
    if (!dst || !src || (bytes % 16)) return DF_E_INVALID;
    size_t n16 = bytes / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(df_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint4*)dst, (const uint4*)src, n16);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// read-only stream probe: the measured roofline denominator for scan kernels (extract)
__global__ __launch_bounds__(256) void df_read_kernel(const uint4* __restrict__ s, size_t n16, unsigned int* __restrict__ sink)
{
    unsigned int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = s[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;              // practically never true: keeps the loads alive without a store stream
}

extern "C" int dfusion_read_bandwidth_probe(const void* src, size_t bytes, void* sink4, dfStream stream)
{
    if (!src || !sink4 || (bytes % 16)) return DF_E_INVALID;
    size_t n16 = bytes / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(df_read_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, n16, (unsigned int*)sink4);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ compute_dists
// imgproc.cu:259-272.  The reference guard `x<cols || y<rows` is a bug (harmless when the grid
// divides evenly); && here.
__global__ __launch_bounds__(256) void df_compute_dists_kernel(const uint16_t* __restrict__ depth, size_t dpitch,
                                                               uint16_t* __restrict__ dists, size_t opitch, int cols,
                                                               int rows, float finvx, float finvy, float cx, float cy)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    float xl = ((float)x - cx) * finvx;
    float yl = ((float)y - cy) * finvy;
    float lambda = sqrtf(xl * xl + yl * yl + 1);
    uint16_t d = *(const uint16_t*)((const char*)depth + (size_t)y * dpitch + 2 * (size_t)x);
    *(uint16_t*)((char*)dists + (size_t)y * opitch + 2 * (size_t)x) = (uint16_t)f2h_bits((float)d * lambda * 0.001f);
}

extern "C" int dfusion_compute_dists(const uint16_t* depth, size_t depth_pitch, uint16_t* dists, size_t dists_pitch,
                                     int cols, int rows, const float intr[4], dfStream stream)
{
    if (!depth || !dists || !intr || cols <= 0 || rows <= 0) return DF_E_INVALID;
    dim3 grid((cols + 63) / 64, (rows + 3) / 4);
    hipLaunchKernelGGL(df_compute_dists_kernel, grid, dim3(256), 0, (hipStream_t)stream, depth, depth_pitch, dists,
                       dists_pitch, cols, rows, 1.f / intr[0], 1.f / intr[1], intr[2], intr[3]);   // imgproc.cu:292
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ project_and_remove / psdf
// project_kernel, tsdf_volume.cu:113-139, plus the per-point arithmetic of TsdfVolume::psdf (tsdf_volume.cpp:284-290):
// ro = (K^-1 * (coo.x*Dp, coo.y*Dp, Dp))[2] - point.z = b22*Dp - point.z with b22 from the closed-form 3x3 inverse.
// Reads the immutable dists_in, zeroes dists_out (the reference does both on one image: racy, see include/dfusion.h).
__global__ __launch_bounds__(256) void df_project_kernel(const uint16_t* __restrict__ din, size_t ipitch,
                                                         uint16_t* __restrict__ dout, size_t opitch, int cols, int rows,
                                                         float4* __restrict__ points, unsigned long long n, float fx, float fy,
                                                         float cx, float cy, float b22, float* __restrict__ ro,
                                                         unsigned long long* __restrict__ n_inside)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    bool inside = false;
    if (i < n) {
        const float4 p = points[i];
        const float qnan = __uint_as_float(0x7fffffffu);
        if (isnan(p.x) || isnan(p.y) || isnan(p.z)) {                       // :121
            if (ro) ro[i] = qnan;
        } else {
            const float u = fmaf(fx, p.x / p.z, cx);                        // device.hpp:35
            const float v = fmaf(fy, p.y / p.z, cy);                        // device.hpp:36
            if (!(u >= 0.f && v >= 0.f && v < (float)rows && u < (float)cols)) {   // :125 (+NaN => outside)
                points[i] = make_float4(qnan, qnan, qnan, 0.f);
                if (ro) ro[i] = qnan;
            } else {
                const float Dp = h2f_bits(*(const uint16_t*)((const char*)din + (size_t)(int)v * ipitch + 2 * (size_t)(int)u));
                if (dout) *(uint16_t*)((char*)dout + (size_t)(int)v * opitch + 2 * (size_t)(int)u) = 0;   // :132
                points[i] = make_float4(u * Dp, v * Dp, Dp, 0.f);           // :133
                if (ro) ro[i] = (0.f + b22 * Dp) - p.z;
                inside = true;
            }
        }
    }
    if (n_inside) {
        const unsigned long long m = __ballot(inside);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(n_inside, (unsigned long long)__popcll(m));
    }
}

extern "C" int dfusion_project_and_remove(const uint16_t* dists_in, size_t in_pitch, uint16_t* dists_out, size_t out_pitch,
                                          int cols, int rows, float* points, unsigned long long n, const float proj[4],
                                          float* ro, unsigned long long* n_inside, dfStream stream)
{
    if (!dists_in || !points || !proj || cols <= 0 || rows <= 0) return DF_E_INVALID;
    if (dists_out == dists_in) return DF_E_INVALID;
    if (n == 0) return DF_OK;
    const float P = proj[0] * proj[1];
    const float dinv = 1.f / P;
    const float b22 = P * dinv;                                             // Matx33f::inv(), third row (0, 0, b22)
    hipLaunchKernelGGL(df_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dists_in,
                       in_pitch, dists_out, out_pitch, cols, rows, (float4*)points, n, proj[0], proj[1], proj[2], proj[3], b22,
                       ro, n_inside);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ max-pyramid of the dists image (dfusion_pyramid.h)
// levels 1..5 of one 32 x 32 pixel tile per workgroup
__global__ __launch_bounds__(256) void df_pyramid_tiles_kernel(const DfDistsPyramid P, uint16_t* __restrict__ out)
{
    __shared__ uint16_t s[16 * 16];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int x0 = blockIdx.x * 32 + 2 * tx, y0 = blockIdx.y * 32 + 2 * ty;
    uint32_t m = 0;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const int x = x0 + dx, y = y0 + dy;
            if (x < P.cols && y < P.rows) m = max(m, (uint32_t)*(const uint16_t*)((const char*)P.dists + (size_t)y * P.pitch + 2 * (size_t)x));
        }
    s[ty * 16 + tx] = (uint16_t)m;
    {
        const int X = blockIdx.x * 16 + tx, Y = blockIdx.y * 16 + ty;
        if (P.top >= 1 && X < P.w[1] && Y < P.h[1]) out[P.off[1] + Y * P.w[1] + X] = (uint16_t)m;
    }
    __syncthreads();
#pragma unroll
    for (int l = 2; l <= 5; ++l) {
        const int n = 32 >> l;                 // tile edge at this level: 8, 4, 2, 1
        const int pn = n * 2;
        uint32_t v = 0;
        const bool act = t < n * n;
        const int cx = t % n, cy = t / n;
        if (act) v = max(max((uint32_t)s[(2 * cy) * pn + 2 * cx], (uint32_t)s[(2 * cy) * pn + 2 * cx + 1]),
                         max((uint32_t)s[(2 * cy + 1) * pn + 2 * cx], (uint32_t)s[(2 * cy + 1) * pn + 2 * cx + 1]));
        __syncthreads();
        if (act) {
            s[cy * n + cx] = (uint16_t)v;
            const int X = blockIdx.x * n + cx, Y = blockIdx.y * n + cy;
            if (l <= P.top && X < P.w[l] && Y < P.h[l]) out[P.off[l] + Y * P.w[l] + X] = (uint16_t)v;
        }
        __syncthreads();
    }
}
// levels 6..top from level 5, one workgroup, through LDS (level 5 of a 4096 x 4096 image is 128 x 128)
__global__ __launch_bounds__(256) void df_pyramid_top_kernel(const DfDistsPyramid P, uint16_t* __restrict__ out)
{
    extern __shared__ uint16_t sm[];
    const int n5 = P.w[5] * P.h[5];
    for (int i = threadIdx.x; i < n5; i += 256) sm[i] = out[P.off[5] + i];
    __syncthreads();
    int pw = P.w[5], ph = P.h[5];
    uint16_t* cur = sm;
    uint16_t* nxt = sm + n5;
    for (int l = 6; l <= P.top; ++l) {
        const int w = P.w[l], h = P.h[l];
        for (int i = threadIdx.x; i < w * h; i += 256) {
            const int x = i % w, y = i / w;
            uint32_t v = 0;
            for (int dy = 0; dy < 2; ++dy)
                for (int dx = 0; dx < 2; ++dx) {
                    const int sx = 2 * x + dx, sy = 2 * y + dy;
                    if (sx < pw && sy < ph) v = max(v, (uint32_t)cur[sy * pw + sx]);
                }
            nxt[i] = (uint16_t)v;
            out[P.off[l] + i] = (uint16_t)v;
        }
        __syncthreads();
        uint16_t* tmp = cur; cur = nxt; nxt = tmp;
        pw = w; ph = h;
    }
}
// ------------------------------------------------------------------------------------------ integrate (rigid)
static bool g_df_rigid_no_depth_cull = false, g_df_rigid_no_fast_forms = false;
// bit 0: behind-the-surface test, bit 1: short arithmetic forms (default 3 = both on; validation switches, results must not change)
extern "C" int dfusion_debug_rigid(int flags) { g_df_rigid_no_depth_cull = !(flags & 1); g_df_rigid_no_fast_forms = !(flags & 2); return DF_OK; }

struct DfRigidArgs {
    uint32_t* vol;            // first stored plane
    int X, Y;
    int z_store0, z_own0, z_own_n;
    int zc;                   // planes per chunk (blockIdx.y)
    DfAff vol2cam;
    float vsx, vsy, vsz;
    DfIntegrateParams P;
    unsigned long long* n_upd;
};

// Signed test "certainly outside one image-frustum side plane or behind the camera by more than m metres".
// The four side planes pass through the camera centre: u >= 0 <=> fx*x + cx*z >= 0, u < cols <=> -fx*x + (cols-cx)*z > 0,
// same for v.  A voxel with z <= 0 is skipped by the exact test anyway, so the plane tests are safe for any z.
struct DfFrustum { float nlx, nlz, nrx, nrz, nty, ntz, nby, nbz; };   // unit normals (pointing inside) of left/right/top/bottom
__device__ __forceinline__ unsigned df_outside_mask(const DfFrustum& F, f3 p, float m)
{
    unsigned o = 0;
    if (p.z < -m) o |= 1u;
    if (F.nlx * p.x + F.nlz * p.z < -m) o |= 2u;
    if (F.nrx * p.x + F.nrz * p.z < -m) o |= 4u;
    if (F.nty * p.y + F.ntz * p.z < -m) o |= 8u;
    if (F.nby * p.y + F.nbz * p.z < -m) o |= 16u;
    return o;
}

// Conservative, result-identical rejection of the voxels of one column on n consecutive planes, given its running camera-frame
// position a0 on the first of them.  All those voxels lie (up to the accumulated rounding of `vc += zstep`, < 1e-3 m over 1024
// planes) on the segment from a0 to a0 + (n-1) zstep.  None of them can take the update branch (tsdf_volume.cu:82-91) if
//   (a) both ends are outside the SAME frustum side plane (or behind the camera) by 5 mm, or
//   (b) every ray-length the segment can be compared with is too short: it projects (a segment in front of the camera onto the
//       segment between the projections) into a pixel rectangle whose largest dists value is Dmax, and its smallest distance
//       from the camera centre exceeds Dmax + trunc -- then sdf = Dp - |vc| < -trunc for all of them;
//       Dmax == 0 means no valid depth at all there (Dp == 0, :86).
// (b) is what skips the volume BEHIND the observed surface, most of what the frustum contains.
template <bool DEPTH>
__device__ __forceinline__ bool df_rigid_culled(const DfFrustum& F, const DfDistsPyramid& Py, const DfIntegrateParams& P, f3 a0, f3 zstep, int n)
{
    // one column: the hull is the segment a0 .. b0
    const f3 b0 = add3(a0, scale3(zstep, (float)(n - 1)));
    const float m = 5e-3f;
    if ((df_outside_mask(F, a0, m) & df_outside_mask(F, b0, m)) != 0u) return true;
    if (!DEPTH || Py.top == 0) return false;
    const float zmin = fminf(a0.z, b0.z);
    if (!(zmin > 0.05f)) return false;
    // pixel rectangle (approximate reciprocals are fine: 2 pixels of margin)
    const float r0 = __builtin_amdgcn_rcpf(a0.z), r2 = __builtin_amdgcn_rcpf(b0.z);
    const float u0 = P.fx * a0.x * r0, u2 = P.fx * b0.x * r2;
    const float v0 = P.fy * a0.y * r0, v2 = P.fy * b0.y * r2;
    const float ulo = fminf(u0, u2) + P.cx - 2.f, uhi = fmaxf(u0, u2) + P.cx + 2.f;
    const float vlo = fminf(v0, v2) + P.cy - 2.f, vhi = fmaxf(v0, v2) + P.cy + 2.f;
    if (!(ulo == ulo && uhi == uhi && vlo == vlo && vhi == vhi)) return false;
    if (uhi < 0.f || vhi < 0.f || ulo > (float)(P.cols - 1) || vlo > (float)(P.rows - 1)) return true;       // projects outside the image
    const int iu0 = (int)fmaxf(ulo, 0.f), iv0 = (int)fmaxf(vlo, 0.f);
    const int iu1 = (int)fminf(uhi, (float)(P.cols - 1)), iv1 = (int)fminf(vhi, (float)(P.rows - 1));
    const uint32_t dbits = df_pyramid_max(Py, iu0, iv0, iu1, iv1);
    if (dbits == 0u) return true;                                        // no valid depth anywhere it can project to
    const float dmax = h2f_bits((uint16_t)dbits);
    // smallest distance of the segment from the camera centre: per-axis smallest |coordinate|
    const float xl = fminf(a0.x, b0.x), xh = fmaxf(a0.x, b0.x), yl = fminf(a0.y, b0.y), yh = fmaxf(a0.y, b0.y);
    const float mx = xl > 0.f ? xl : (xh < 0.f ? -xh : 0.f), my = yl > 0.f ? yl : (yh < 0.f ? -yh : 0.f);
    const float rmin = sqrtf(mx * mx + my * my + zmin * zmin) - m;
    return (dbits < 0x7c00u) && (rmin > dmax * 1.001f + P.trunc);        // finite non-negative length only
}

// One COLUMN per lane, a wave = a 32(x) x 2(y) column patch (two 128-byte lines per plane), a Z chunk per blockIdx.y, walked in
// sub-chunks of DF_RIGID_SUB planes.  How it got here (512^3, MI355X): four columns per lane and half a row per wave took 0.24 ms
// although the update arithmetic of the 31 M voxels that update is ~45 us of VALU issue -- a wave costs what its busiest lane
// costs, nearly every wave cut the frustum, and the few thousand long-running waves landed unevenly on the 1024 SIMDs.  Now:
//   * compact patches and short chunks: 16 k waves, most of which reject their whole chunk at once (df_rigid_culled on the lane's
//     own column segment, verdict per wave by ballot) and leave the SIMD to the next one -- the dispatcher does the balancing;
//   * a rejected sub-chunk only advances the running position (the same `vc += zstep` additions, tsdf_volume.cu:75, 3 adds per
//     plane instead of ~100 instructions): everything outside the frustum AND everything more than trunc behind the surface;
//   * U planes per batch: U independent sample chains and U voxel loads in flight per lane (the chain divide -> dists fetch ->
//     sqrt / compare -> voxel load -> fuse -> store is what a lone voxel waits on).
// A chunk starting at plane zb replays the zb additions of :75 in registers, so every chunk -- and every Z-slab shard on another
// GPU -- produces the bits of the unsharded sweep.
// the voxels of one column on planes [zs, zse), U at a time
template <int U, bool FAST>
__device__ __forceinline__ void df_rigid_batches(const DfRigidArgs& a, f3& vc, f3 zstep, uint32_t*& p, size_t plane, int zs, int zse,
                                                 bool active, unsigned int& my_upd)
{
    for (int z = zs; z < zse; z += U) {
        float ts[U];
        bool up[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                   // stage 1: U branch-free sample chains
            const bool inr = z + u < zse;
            up[u] = (FAST ? tsdf_sample_fast(a.P, vc, &ts[u]) : tsdf_sample_nb(a.P, vc, &ts[u])) && inr && active;
            if (inr) vc = add3(vc, zstep);              // :75
        }
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)                     // stage 2: the voxel loads of the batch in flight together
            if (up[u]) v[u] = p[(size_t)u * plane];
#pragma unroll
        for (int u = 0; u < U; ++u)                     // stage 3: fuse (:97-103) and store
            if (up[u]) { p[(size_t)u * plane] = tsdf_fuse(v[u], ts[u], a.P.max_weight); ++my_upd; }
        p += (size_t)min(U, zse - z) * plane;
    }
}

#define DF_RIGID_SUB 16
template <int U, bool DEPTH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void df_integrate_rigid_kernel(const DfRigidArgs a, const DfFrustum F, const DfDistsPyramid Py, const bool FASTOK)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles_x = (a.X + 31) >> 5;
    const int tile = blockIdx.x * 4 + wv;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x = tx * 32 + (lane & 31), y = ty * 2 + (lane >> 5);
    const bool active = x < a.X && y < a.Y;
    unsigned int my_upd = 0;
    const int zb = a.z_own0 + blockIdx.y * a.zc;
    const int ze = min(zb + a.zc, a.z_own0 + a.z_own_n);
    const f3 zstep = scale3(mk3(a.vol2cam.R[2], a.vol2cam.R[5], a.vol2cam.R[8]), a.vsz);      // tsdf_volume.cu:69 (three separate multiplies)
    f3 vc = aff_mul(a.vol2cam, mk3((float)x * a.vsx, (float)y * a.vsy, 0.f));                  // :71-72
    // whole chunk first (position by multiplication: within the test's margin of the running sum): skips the replay too.
    // Verdicts are per WAVE: a lane that could have been skipped alone runs the exact test instead, which finds "no update" by
    // itself -- cheaper than a wave executing both branches.
    const bool culled = !active || df_rigid_culled<DEPTH>(F, Py, a.P, add3(vc, scale3(zstep, (float)zb)), zstep, ze - zb);
    if (__ballot(!culled) != 0ull) {
        for (int z = 0; z < zb; ++z) vc = add3(vc, zstep);          // replay of `vc += zstep` (:75) for planes [0, zb)
        const size_t plane = (size_t)a.X * a.Y;
        uint32_t* p = a.vol + (size_t)(zb - a.z_store0) * plane + (size_t)(active ? y : 0) * a.X + (active ? x : 0);
        for (int zs = zb; zs < ze; zs += DF_RIGID_SUB) {
            const int zse = min(zs + DF_RIGID_SUB, ze);
            const bool sub_culled = !active || df_rigid_culled<DEPTH>(F, Py, a.P, vc, zstep, zse - zs);
            if (__ballot(!sub_culled) == 0ull) {
                for (int z = zs; z < zse; ++z) vc = add3(vc, zstep);                            // :75, skipped voxels included
                p += (size_t)(zse - zs) * plane;
                continue;
            }
            // the short arithmetic forms of tsdf_sample_fast need their domain on every voxel of the run, for every lane
            const bool fast = FASTOK && df_wave_all(!active || tsdf_sample_domain_ok(vc, add3(vc, scale3(zstep, (float)(zse - zs)))));
            if (fast) df_rigid_batches<U, true>(a, vc, zstep, p, plane, zs, zse, active, my_upd);
            else df_rigid_batches<U, false>(a, vc, zstep, p, plane, zs, zse, active, my_upd);
        }
    }
    if (a.n_upd) {                                            // one atomic per wave
        unsigned int s = my_upd;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(a.n_upd, (unsigned long long)s);
    }
}

// the dists max-pyramid of one frame into `mem` (levels 1..top); returns the descriptor
int df_build_dists_pyramid(const uint16_t* dists, size_t pitch, int cols, int rows, uint16_t* mem, size_t mem_elems,
                                  DfDistsPyramid* out, hipStream_t st)
{
    DfDistsPyramid P;
    memset(&P, 0, sizeof(P));
    P.dists = dists; P.pitch = pitch; P.cols = cols; P.rows = rows; P.mem = mem;
    int off = 0, l = 1;
    for (; l < DF_PYR_MAX_LEVELS; ++l) {
        P.w[l] = (cols + (1 << l) - 1) >> l; P.h[l] = (rows + (1 << l) - 1) >> l; P.off[l] = off;
        off += P.w[l] * P.h[l];
        if (P.w[l] == 1 && P.h[l] == 1) break;
    }
    if (l >= DF_PYR_MAX_LEVELS || (size_t)off > mem_elems || l < 5) { out->top = 0; return DF_OK; }     // (images below 32 px: no test)
    P.top = l;
    hipLaunchKernelGGL(df_pyramid_tiles_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, P, mem);
    DF_LAUNCH_CHECK();
    if (P.top > 5) {
        const size_t lds = 2 * (size_t)P.w[5] * P.h[5] * sizeof(uint16_t);
        if (lds > 64 * 1024) { out->top = 0; return DF_OK; }
        hipLaunchKernelGGL(df_pyramid_top_kernel, dim3(1), dim3(256), lds, st, P, mem);
        DF_LAUNCH_CHECK();
    }
    *out = P;
    return DF_OK;
}
size_t df_pyramid_elems(int cols, int rows)
{
    size_t n = 0;
    for (int l = 1; l < DF_PYR_MAX_LEVELS; ++l) {
        const size_t w = (size_t)((cols + (1 << l) - 1) >> l), h = (size_t)((rows + (1 << l) - 1) >> l);
        n += w * h;
        if (w == 1 && h == 1) break;
    }
    return n;
}

// DFUSION_RIGID_NO_DEPTH_CULL=1 switches the behind-the-surface test off (validation: the volumes must be identical)
static bool df_rigid_depth_cull_disabled()
{
    static const bool off = [] { const char* e = getenv("DFUSION_RIGID_NO_DEPTH_CULL"); return e && e[0] == '1'; }();
    return off || g_df_rigid_no_depth_cull;
}

extern "C" int dfusion_integrate(const uint16_t* dists, size_t pitch, int cols, int rows, DfVolume v, const DfSlab* slab,
                                 const float vol2cam[12], const float proj[4], unsigned long long* n_updated,
                                 dfStream stream)
{
    if (!dists || !vol2cam || !proj || cols <= 0 || rows <= 0 || !df_volume_valid(v)) return DF_E_INVALID;
    DfSlab s = df_slab_or_full(v, slab);
    if (!df_slab_valid(v, s)) return DF_E_INVALID;
    if (s.z_own_n == 0) return DF_OK;

    DfRigidArgs a;
    a.vol = (uint32_t*)v.data; a.X = v.dims[0]; a.Y = v.dims[1];
    a.z_store0 = s.z_store0; a.z_own0 = s.z_own0; a.z_own_n = s.z_own_n;
    a.vol2cam = df_aff(vol2cam);
    a.vsx = v.voxel_size[0]; a.vsy = v.voxel_size[1]; a.vsz = v.voxel_size[2];
    a.P.dists = dists; a.P.pitch = pitch; a.P.cols = cols; a.P.rows = rows;
    a.P.fx = proj[0]; a.P.fy = proj[1]; a.P.cx = proj[2]; a.P.cy = proj[3];
    a.P.trunc = v.trunc_dist; a.P.trunc_inv = 1.f / v.trunc_dist;       // tsdf_volume.cu:147
    a.P.max_weight = v.max_weight;
    a.n_upd = n_updated;

    DfFrustum F;
    {
        const float fx = proj[0], fy = proj[1], cx = proj[2], cy = proj[3];
        const float cr = (float)cols - cx, cb = (float)rows - cy;
        const float nl = sqrtf(fx * fx + cx * cx), nr = sqrtf(fx * fx + cr * cr), nt = sqrtf(fy * fy + cy * cy), nb = sqrtf(fy * fy + cb * cb);
        F.nlx = fx / nl; F.nlz = cx / nl; F.nrx = -fx / nr; F.nrz = cr / nr;
        F.nty = fy / nt; F.ntz = cy / nt; F.nby = -fy / nb; F.nbz = cb / nb;
        if (!(fx > 0.f && fy > 0.f && nl > 0.f && nr > 0.f && nt > 0.f && nb > 0.f)) {      // degenerate intrinsics: disable the pre-tests
            F.nlx = F.nrx = F.nty = F.nby = 0.f; F.nlz = F.nrz = F.ntz = F.nbz = 0.f;
        }
    }
    const int tiles = ((a.X + 31) / 32) * ((a.Y + 1) / 2);         // 32 x 2 column patches, one per wave
    const int bx = (tiles + 3) / 4;
    // Z chunking: ~16 k waves (several rounds of the 8 k wave slots, so that the dispatcher can balance the few long-running
    // ones), chunks of whole sub-chunks and >= 32 planes (a chunk at plane zb replays zb additions: 3 per plane per lane)
    const long long want_blocks = 256LL * 16;
    int chunks = (int)((want_blocks + bx - 1) / bx);
    if (chunks < 1) chunks = 1;
    int zc = (s.z_own_n + chunks - 1) / chunks;
    zc = ((zc + DF_RIGID_SUB - 1) / DF_RIGID_SUB) * DF_RIGID_SUB;
    if (zc < 32) zc = 32;
    a.zc = zc;
    dim3 grid(bx, (s.z_own_n + zc - 1) / zc);
    hipStream_t st = (hipStream_t)stream;
    // behind-the-surface test: a max-pyramid of this frame's dists in stream-ordered scratch (no state is kept between calls)
    DfDistsPyramid Py;
    memset(&Py, 0, sizeof(Py));
    uint16_t* pyr_mem = nullptr;
    if (!df_rigid_depth_cull_disabled()) {
        const size_t elems = df_pyramid_elems(cols, rows);
        if (hipMallocAsync((void**)&pyr_mem, elems * sizeof(uint16_t), st) == hipSuccess) {
            int rc = df_build_dists_pyramid(dists, pitch, cols, rows, pyr_mem, elems, &Py, st);
            if (rc) { (void)hipFreeAsync(pyr_mem, st); return rc; }
        } else { (void)hipGetLastError(); pyr_mem = nullptr; }
    }
    // short arithmetic forms (tsdf_sample_fast): 32-bit dists offsets, sane intrinsics; the value domain is tested per run in the kernel
    const bool fast_ok = (unsigned long long)rows * pitch < (1ull << 31) && proj[0] == proj[0] && proj[1] == proj[1] && !g_df_rigid_no_fast_forms;
    if (Py.top) hipLaunchKernelGGL((df_integrate_rigid_kernel<4, true>), grid, dim3(256), 0, st, a, F, Py, fast_ok);
    else hipLaunchKernelGGL((df_integrate_rigid_kernel<4, false>), grid, dim3(256), 0, st, a, F, Py, fast_ok);
    DF_LAUNCH_CHECK();
    if (pyr_mem) DF_HIP(hipFreeAsync(pyr_mem, st));
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ misc
extern "C" int dfusion_abi_version(void) { return DFUSION_ABI_VERSION; }

extern "C" const char* dfusion_error_string(int err)
{
    switch (err) {
        case DF_OK: return "ok";
        case DF_E_INVALID: return "dfusion: invalid argument";
        case DF_E_NO_INDEX: return "dfusion: k-NN brick index missing or stale (call dfusion_warp_build_index)";
        case DF_E_NO_DEVICE: return "dfusion: no HIP device";
        default: return hipGetErrorString((hipError_t)err);
    }
}
