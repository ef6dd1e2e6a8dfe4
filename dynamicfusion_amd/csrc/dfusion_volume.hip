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
#include <atomic>
#include <list>
#include <mutex>
#include <vector>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>

// ------------------------------------------------------------------------------------------ clear
// (a workgroup zeroes CONTIGUOUS 16 KiB runs with non-temporal stores -- the copy probe's lesson below: a grid-stride loop of single
// 16-byte stores wrote 4.3 TB/s)
typedef unsigned int df_fill_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void df_fill_zero_kernel(df_fill_u4* __restrict__ p, size_t n16)
{
    const df_fill_u4 z = {0u, 0u, 0u, 0u};             // pack_tsdf(0.f, 0) == 0 (device.hpp:53-54)
    for (size_t c0 = (size_t)blockIdx.x * 1024; c0 < n16; c0 += (size_t)gridDim.x * 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = c0 + (size_t)u * 256 + threadIdx.x;
            if (i < n16) __builtin_nontemporal_store(z, p + i);
        }
    }
}

extern "C" int dfusion_clear(DfVolume v, const DfSlab* slab, dfStream stream)
{
    if (!df_volume_valid(v)) return DF_E_INVALID;
    DfSlab s = df_slab_or_full(v, slab);
    if (!df_slab_valid(v, s)) return DF_E_INVALID;
    size_t n16 = (size_t)v.dims[0] * v.dims[1] * s.z_store_n / 4;
    size_t blocks = (n16 + 1023) / 1024;
    if (blocks > 256 * 16) blocks = 256 * 16;         // 16 blocks per CU, each walking 16 KiB runs
    hipLaunchKernelGGL(df_fill_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (df_fill_u4*)v.data, n16);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ------------------------------------------------------------------------------------------ copy probe
// The MEASURED roofline denominator.  Which plain copy is fastest on this part was measured (tools/copy_probe.hip, round 3): a
// grid-stride loop of single 16-byte accesses -- the usual form, and round 2's probe -- reaches 4.4-4.9 TB/s, because the lines a
// workgroup touches at one time are scattered over the whole buffer; a workgroup that copies ONE CONTIGUOUS 16 KiB run, four
// non-temporal 16-byte loads in flight per lane and non-temporal stores, reaches 5.7-6.3 TB/s (the guide's 6.29 TB/s float4 copy).
typedef unsigned int df_probe_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void df_copy_kernel(df_probe_u4* __restrict__ d, const df_probe_u4* __restrict__ s, size_t n16)
{
    const size_t b = (size_t)blockIdx.x * 1024, e = b + 1024 < n16 ? b + 1024 : n16;
    size_t i = b + threadIdx.x;
    if (i + 768 < e) {
        df_probe_u4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(s + i + u * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], d + i + u * 256);
    } else {
        for (; i < e; i += 256) d[i] = s[i];
    }
}

extern "C" int dfusion_copy_bandwidth_probe(void* dst, const void* src, size_t bytes, dfStream stream)
{
    if (!dst || !src || (bytes % 16)) return DF_E_INVALID;
    const size_t n16 = bytes / 16;
    const size_t blocks = (n16 + 1023) / 1024;
    if (blocks == 0) return DF_OK;
    if (blocks > 0x7fffffffull) return DF_E_INVALID;
    hipLaunchKernelGGL(df_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (df_probe_u4*)dst, (const df_probe_u4*)src, n16);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// read-only stream probe: the measured roofline denominator for scan kernels (extract); four non-temporal loads in flight per lane
__global__ __launch_bounds__(256) void df_read_kernel(const df_probe_u4* __restrict__ s, size_t n16, unsigned int* __restrict__ sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned int acc = 0;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        df_probe_u4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(s + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const df_probe_u4 v = s[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9e3779b9u) *sink = acc;              // practically never true: keeps the loads alive without a store stream
}

extern "C" int dfusion_read_bandwidth_probe(const void* src, size_t bytes, void* sink4, dfStream stream)
{
    if (!src || !sink4 || (bytes % 16)) return DF_E_INVALID;
    size_t n16 = bytes / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (blocks == 0) return DF_OK;
    hipLaunchKernelGGL(df_read_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const df_probe_u4*)src, n16, (unsigned int*)sink4);
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
__global__ __launch_bounds__(256) void df_pyramid_tiles_kernel(const DfDistsPyramid P, uint16_t* __restrict__ out, unsigned int* __restrict__ zero16,
                                                               unsigned int* __restrict__ max_word)
{
    if (zero16 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) zero16[threadIdx.x] = 0u;     // (the rigid plan's 64 counters)
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
            if (l == 5 && max_word) atomicMax(max_word, v);                  // (t == 0: the tile's maximum; one atomic per workgroup)
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
// Validation switches and the swept-voxel counter are PER CALL (dfusion_integrate_ex's flags / n_swept_dev): nothing process-wide.
struct DfRigidArgs {
    uint32_t* vol;            // first stored plane
    int X, Y;
    int z_store0, z_own0, z_own_n;
    int zc;                   // planes per chunk (blockIdx.y)
    DfAff vol2cam;
    float vsx, vsy, vsz;
    DfIntegrateParams P;
    unsigned long long* n_upd;
    unsigned long long* n_swept;   // nullable: += voxels that went through the sample chain (alive sub-chunks x columns inside the volume)
    // launch plan (df_rigid_plan_kernel).  A patch = DF_RIGID_PX x DF_RIGID_PY columns (one wave), tile = ty * tiles_x + tx with
    // tiles_x a multiple of 4; a STRIP = 4 patches side by side in x (one workgroup: the four waves touch 4 * PX * 4 contiguous bytes
    // of every voxel row together); item = chunk * strips + strip, its mask = 4 x 8 bits (bit 8 k + s: sub-chunk s of patch k may
    // update, planes [zb + SUB s, zb + SUB s + SUB)); items with w > 0 bits are listed in bin w (plan_bins[w * plan_items ...],
    // plan_cnt[w] entries), their masks in plan_mask[item]
    const unsigned int* plan_bins; const unsigned int* plan_cnt; const unsigned int* plan_mask; unsigned int plan_items; int tiles, tiles_x;
    const float4* plan_starts;     // [item][lane]: the running position of lane's column at the item's first plane
#ifdef DF_TRACE_WG
    unsigned long long* trace;
#endif
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

// Rigid integrate: one COLUMN per lane, a wave = a 32(x) x 2(y) column patch (two 128-byte lines per plane) over one Z chunk, walked
// in sub-chunks of DF_RIGID_SUB planes.  How it got here (512^3, MI355X): four columns per lane and half a row per wave took 0.24 ms
// although the update arithmetic of the 31 M voxels that update is ~45 us of VALU issue -- a wave costs what its busiest lane
// costs, nearly every wave cut the frustum, and the few thousand long-running waves landed unevenly on the 1024 SIMDs.  Compact
// patches, short chunks whose dead sub-chunks only advance the running position (the same `vc += zstep` additions,
// tsdf_volume.cu:75, 3 adds per plane instead of ~100 instructions: everything outside the frustum AND everything more than trunc
// behind the surface) and U planes per batch (U independent sample chains and voxel loads in flight per lane: the chain divide ->
// dists fetch -> sqrt / compare -> voxel load -> fuse -> store is what a lone voxel waits on) made it 0.17 ms -- half of it a tail:
// a per-wave timeline (tools/trace_sweep.py) showed the SIMDs full for 80 us and then 80 us of ever fewer long waves, the dense far
// chunks being dispatched last.  Now the verdicts are made up front (df_rigid_plan_kernel: one LANE per sub-chunk, on the patch's
// box), the sweep's waves take the alive items most-work-first and carry no tests: full from start to end, 0.135 ms.
// A chunk starting at plane zb replays the zb additions of :75 in registers, so every chunk -- and every Z-slab shard on another
// GPU -- produces the bits of the unsharded sweep.
// the voxels of one column on planes [zs, zse), U at a time
template <int U, bool FAST, bool SAT, bool FULL, bool COUNT>
__device__ __forceinline__ void df_rigid_batches(const DfRigidArgs& a, f3& vc, f3 zstep, uint32_t*& p, size_t plane, int zs, int zse,
                                                 bool active, unsigned int& my_upd)
{
    // FULL: the run is a whole number of batches (every sub-chunk but a slab's last): no per-plane range tests
    const float sat_t = df_sat_threshold(a.P.trunc);
    for (int z = zs; z < zse; z += U) {
        float ts[U];
        bool up[U];
        bool sat = false;                               // every sample of the batch is decided without the exact square root
        if constexpr (FAST && SAT) {
            // stage 1 in two halves (dfusion_device.h, tsdf_sample_pre): the approximate |vc| decides wherever the voxel is not
            // within trunc of the surface -- the tsdf is then exactly 1.f or the voxel does not update
            DfSamplePre pre[U];
            bool undecided = false;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool inr = FULL || z + u < zse;
                pre[u] = tsdf_sample_pre(a.P, vc);
                ts[u] = pre[u].Dp - pre[u].s;           // approximate sdf
                up[u] = pre[u].ok && inr && active;
                undecided = undecided | (up[u] & (fabsf(ts[u]) < sat_t));
                if (inr) vc = add3(vc, zstep);          // :75
            }
            sat = __builtin_amdgcn_ballot_w64(undecided) == 0ull;
            if (sat) {
#pragma unroll
                for (int u = 0; u < U; ++u) { up[u] = up[u] & (ts[u] >= sat_t); ts[u] = 1.f; }
            } else {                                    // within trunc of the surface: the exact root and :89-93 as written
#pragma unroll
                for (int u = 0; u < U; ++u) up[u] = tsdf_sample_finish(a.P, pre[u], &ts[u]) & up[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {               // stage 1: U branch-free sample chains
                const bool inr = FULL || z + u < zse;
                up[u] = (FAST ? tsdf_sample_fast(a.P, vc, &ts[u]) : tsdf_sample_nb(a.P, vc, &ts[u])) && inr && active;
                if (inr) vc = add3(vc, zstep);          // :75
            }
        }
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)                     // stage 2: the voxel loads of the batch in flight together
            if (up[u]) v[u] = p[(size_t)u * plane];
        bool one = false;                               // saturated samples onto stored 1.0 / cleared voxels: the fuse is a weight increment
        if (FAST && SAT && sat) {
            one = true;
#pragma unroll
            for (int u = 0; u < U; ++u) one = one & (!up[u] | tsdf_fuse_one_ok(v[u]));
            one = df_wave_all(one);
        }
        if (one) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (up[u]) { p[(size_t)u * plane] = tsdf_fuse_one(v[u], a.P.max_weight); if (COUNT) ++my_upd; }
        } else {
            bool fin = FAST;                            // the fuse division's short form: finite stored values (a wave decides together)
            if (FAST) {
#pragma unroll
                for (int u = 0; u < U; ++u) fin = fin & (!up[u] | tsdf_fuse_short_ok(v[u]));
                fin = df_wave_all(fin);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)                 // stage 3: fuse (:97-103) and store
                if (up[u]) { p[(size_t)u * plane] = fin ? tsdf_fuse_short(v[u], ts[u], a.P.max_weight) : tsdf_fuse(v[u], ts[u], a.P.max_weight); if (COUNT) ++my_upd; }
        }
        p += (size_t)(FULL ? U : min(U, zse - z)) * plane;
    }
}

// The same run of voxels, software-pipelined (FAST + SAT, whole batches): the sweep is bound by what a voxel WAITS for -- the dists
// gather, then the voxel word, each a trip to L2 / HBM -- not by its instructions (PMC, round 3: a third fewer VALU instructions
// changed nothing while 2 x 8 voxels per SIMD were in flight).  So (a) the voxel word is requested together with the dists gather,
// for every column of the patch inside the volume, whatever the verdict will be (the sweep reads 1.26 x the words it writes), and
// (b) batch b + 1 is projected and its four requests issued BEFORE batch b is decided, fused and stored: 2 U voxels in flight per
// wave instead of U, and one wait per batch instead of two dependent ones.
// (Chunk starts are made once per column by the plan kernel -- every chunk replaying the additions from plane 0 was the first form --
// and the run is software-pipelined one batch ahead; both used to be compile-time switches.)
template <int U> struct DfRigidPend { float Dp[U], d2[U], s[U]; uint32_t v[U]; bool ok[U]; };
template <int U>
__device__ __forceinline__ void df_rigid_issue(const DfRigidArgs& a, f3& vc, f3 zstep, const uint32_t* p, size_t plane, bool active, DfRigidPend<U>& P)
{
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const DfSamplePre pre = tsdf_sample_pre(a.P, vc);                   // (issues the dists gather)
        P.Dp[u] = pre.Dp; P.d2[u] = pre.d2; P.s[u] = pre.s; P.ok[u] = pre.ok && active;
        P.v[u] = 0u;
        // the voxel word, unconditionally.  (Round 4 measured the obvious refinement -- only for voxels that project into the image, a
        // test on the geometry alone, so the load still goes out with the dists gather: 0.1123 against 0.1122 ms, same box,
        // volumes identical.  The words not read were never what the sweep waited for; left out.)
        if (active) P.v[u] = p[(size_t)u * plane];
        vc = add3(vc, zstep);                                               // :75
    }
}
template <int U, bool COUNT>
__device__ __forceinline__ void df_rigid_finish(const DfRigidArgs& a, uint32_t* p, size_t plane, const DfRigidPend<U>& P, float sat_t, unsigned int& my_upd)
{
    float ts[U]; bool up[U];
    bool undecided = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        ts[u] = P.Dp[u] - P.s[u];                                           // approximate sdf (see tsdf_sample_pre)
        undecided = undecided | (P.ok[u] & (fabsf(ts[u]) < sat_t));
    }
    const bool sat = __builtin_amdgcn_ballot_w64(undecided) == 0ull;
    if (sat) {
#pragma unroll
        for (int u = 0; u < U; ++u) { up[u] = P.ok[u] & (ts[u] >= sat_t); ts[u] = 1.f; }
    } else {                                                                // within trunc of the surface: the exact root, :89-93 as written
#pragma unroll
        for (int u = 0; u < U; ++u) {
            DfSamplePre pre; pre.Dp = P.Dp[u]; pre.d2 = P.d2[u]; pre.s = P.s[u]; pre.ok = P.ok[u];
            up[u] = tsdf_sample_finish(a.P, pre, &ts[u]);
        }
    }
    bool one = sat;                                                         // saturated samples onto stored 1.0 / cleared voxels: a weight increment
    if (sat) {
#pragma unroll
        for (int u = 0; u < U; ++u) one = one & (!up[u] | tsdf_fuse_one_ok(P.v[u]));
        one = df_wave_all(one);
    }
    if (one) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (up[u]) { p[(size_t)u * plane] = tsdf_fuse_one(P.v[u], a.P.max_weight); if (COUNT) ++my_upd; }
    } else {
        bool fin = true;
#pragma unroll
        for (int u = 0; u < U; ++u) fin = fin & (!up[u] | tsdf_fuse_short_ok(P.v[u]));
        fin = df_wave_all(fin);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (up[u]) { p[(size_t)u * plane] = fin ? tsdf_fuse_short(P.v[u], ts[u], a.P.max_weight) : tsdf_fuse(P.v[u], ts[u], a.P.max_weight); if (COUNT) ++my_upd; }
    }
}
// planes [zs, zs + n), n a multiple of U
template <int U, bool COUNT>
__device__ __forceinline__ void df_rigid_run_pipe(const DfRigidArgs& a, f3& vc, f3 zstep, uint32_t*& p, size_t plane, int n, bool active, unsigned int& my_upd)
{
    const float sat_t = df_sat_threshold(a.P.trunc);
    DfRigidPend<U> A, B;
    df_rigid_issue<U>(a, vc, zstep, p, plane, active, A);
    for (int z = U; z < n; z += 2 * U) {
        df_rigid_issue<U>(a, vc, zstep, p + (size_t)U * plane, plane, active, B);
        df_rigid_finish<U, COUNT>(a, p, plane, A, sat_t, my_upd);
        p += (size_t)U * plane;
        if (z + U < n) {
            df_rigid_issue<U>(a, vc, zstep, p + (size_t)U * plane, plane, active, A);
            df_rigid_finish<U, COUNT>(a, p, plane, B, sat_t, my_upd);
            p += (size_t)U * plane;
        } else { A = B; }
    }
    df_rigid_finish<U, COUNT>(a, p, plane, A, sat_t, my_upd);
    p += (size_t)U * plane;
}

#define DF_RIGID_PX 32           // columns of a wave's patch along x (a power of two <= 64); 64 / DF_RIGID_PX rows
#define DF_RIGID_PY (64 / DF_RIGID_PX)
#define DF_RIGID_SUB 8
#define DF_RIGID_U 2
#define DF_RIGID_U_GENERIC 2
#define DF_RIGID_WAVES 8
#define DF_RIGID_MAX_SUBS 8      // sub-chunks per chunk (the plan's masks are bytes)
#define DF_RIGID_STRIP 1         // patches (waves) side by side in x per plan item: 1, 2 or 4 (a sweep workgroup = 4 waves = 4 / STRIP items)
#define DF_RIGID_ZC 32           // planes per chunk wanted (a multiple of DF_RIGID_SUB, <= DF_RIGID_SUB * DF_RIGID_MAX_SUBS)
#define DF_RIGID_BINS (DF_RIGID_STRIP * DF_RIGID_MAX_SUBS + 1)
// (Round 5 built PAIRS -- two x-adjacent 32 x 2 patches whose alive sub-chunks coincide walked as two 64 x 1 rows, one 256-byte run per
// plane instead of two of 128: 0.1123 ms against 0.1121 without, same box, volumes identical, profiles/r05_ab_rigid_pairs.txt.  The sweep
// is not bound by how wide its row runs are; the variant is gone from the source.)
// Conservative, result-identical rejection of ALL the voxels of a 32 x 2 column patch on planes [zs, zs + n): the same two tests as
// df_rigid_culled, on the box the patch's voxels span (its 8 corners; positions by multiplication, within the tests' margin of the
// running sums the sweep carries): (a) all corners outside the same frustum side plane, (b) the box's least distance from the camera
// centre exceeds, by more than trunc, the largest dists value over the pixel rectangle that bounds the corners' projections.
template <bool DEPTH>
__device__ __forceinline__ bool df_rigid_box_culled(const DfRigidArgs& a, const DfFrustum& F, const DfDistsPyramid& Py, int x0, int y0, int zs, int n)
{
    const float m = 5e-3f;
    const int x1 = min(x0 + DF_RIGID_PX - 1, a.X - 1), y1 = min(y0 + DF_RIGID_PY - 1, a.Y - 1), z1 = zs + n - 1;
    unsigned out_all = 31u;
    float xl = 3.0e38f, xh = -3.0e38f, yl = 3.0e38f, yh = -3.0e38f, zl = 3.0e38f, zh = -3.0e38f;
    float ul = 3.0e38f, uh = -3.0e38f, vl = 3.0e38f, vh = -3.0e38f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f3 q = aff_mul(a.vol2cam, mk3((float)((c & 1) ? x1 : x0) * a.vsx, (float)((c & 2) ? y1 : y0) * a.vsy, (float)((c & 4) ? z1 : zs) * a.vsz));
        out_all &= df_outside_mask(F, q, m);
        xl = fminf(xl, q.x); xh = fmaxf(xh, q.x); yl = fminf(yl, q.y); yh = fmaxf(yh, q.y); zl = fminf(zl, q.z); zh = fmaxf(zh, q.z);
        const float r = __builtin_amdgcn_rcpf(q.z);                   // (only used when every corner has z > 0.05)
        const float u = a.P.fx * q.x * r, v = a.P.fy * q.y * r;
        ul = fminf(ul, u); uh = fmaxf(uh, u); vl = fminf(vl, v); vh = fmaxf(vh, v);
    }
    if (out_all != 0u) return true;
    if (!DEPTH || Py.top == 0) return false;
    if (!(zl > 0.05f)) return false;
    // the box is convex and in front of the camera: every voxel projects inside the bounding rectangle of the corners' projections
    const float ulo = ul + a.P.cx - 2.f, uhi = uh + a.P.cx + 2.f, vlo = vl + a.P.cy - 2.f, vhi = vh + a.P.cy + 2.f;
    if (!(ulo == ulo && uhi == uhi && vlo == vlo && vhi == vhi)) return false;
    if (uhi < 0.f || vhi < 0.f || ulo > (float)(a.P.cols - 1) || vlo > (float)(a.P.rows - 1)) return true;   // projects outside the image
    const int iu0 = (int)fmaxf(ulo, 0.f), iv0 = (int)fmaxf(vlo, 0.f);
    const int iu1 = (int)fminf(uhi, (float)(a.P.cols - 1)), iv1 = (int)fminf(vhi, (float)(a.P.rows - 1));
    const uint32_t dbits = df_pyramid_max_fine(Py, iu0, iv0, iu1, iv1, 2, 5);       // (levels <= 5: see the launcher)
    if (dbits == 0u) return true;                                        // no valid depth anywhere it can project to
    const float dmax = h2f_bits((uint16_t)dbits);
    const float mx = xl > 0.f ? xl : (xh < 0.f ? -xh : 0.f), my = yl > 0.f ? yl : (yh < 0.f ? -yh : 0.f);
    const float rmin = sqrtf(mx * mx + my * my + zl * zl) - m;
    return (dbits < 0x7c00u) && (rmin > dmax * 1.001f + a.P.trunc);      // finite non-negative length only
}

// The launch plan.  One wave per patch and group of 8 chunks: lane 8 c + s tests sub-chunk s of the group's c-th chunk (the box of
// the whole patch over the sub-chunk's planes), the ballot gives the patch's masks.  Four neighbouring waves are the four patches of a
// STRIP; their masks meet in LDS and make the strip items' 32-bit masks.  Alive items are binned by their number of alive
// (patch, sub-chunk) cells (slots taken per workgroup, one atomic per bin); the sweep takes the bins from the fullest down, so its
// workgroups are the long ones first and the launch does not end on a few of them.
// Why strips (round 3): what bounded the sweep was neither its instructions nor the voxels it had in flight but the WAY it touched
// HBM -- independent waves, each read-modify-writing 64 (16 x 4 patch) or 128 (32 x 2) contiguous bytes per voxel row at places
// unrelated to what every other wave was touching: tools/rmw_probe.hip measures 2.6 / 4.4 TB/s for exactly that traffic, and
// 6.3-6.6 TB/s as soon as >= 256 contiguous bytes of a row are touched together.
// The same wave then makes the CHUNK STARTS of its patch: a chunk at plane zb needs the running position after zb
// additions `vc += zstep` (tsdf_volume.cu:75) -- the sequence cannot be shortcut, its roundings are the result.  Lane l walks column l
// of the patch ONCE, up to the last alive chunk, and leaves the position at every alive chunk's first plane in `starts`.
template <bool DEPTH>
__global__ __launch_bounds__(1024) void df_rigid_plan_kernel(const DfRigidArgs a, const DfFrustum F, const DfDistsPyramid Py, unsigned n_items, int chunks,
                                                             bool keep_all, unsigned int* __restrict__ cnt, unsigned int* __restrict__ bins,
                                                             unsigned int* __restrict__ mask_out, float4* __restrict__ starts)
{
    __shared__ unsigned int s_cnt[DF_RIGID_BINS], s_base[DF_RIGID_BINS];
    __shared__ unsigned long long s_alive[16];
    if (threadIdx.x < DF_RIGID_BINS) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    // a wave = one patch and a group of cpw = 64 / subs chunks (subs = sub-chunks per chunk, a power of two <= 8): lane = cl * subs + sb
    const int subs = a.zc / DF_RIGID_SUB, sshift = __ffs(subs) - 1, cpw = 64 >> sshift;
    const int groups = (chunks + cpw - 1) / cpw;
    const unsigned wg = blockIdx.x * 16u + (threadIdx.x >> 6);            // (chunk group, tile): tile fastest, a.tiles a multiple of 4
    const int cg = (int)(wg / (unsigned)a.tiles), tile = (int)(wg % (unsigned)a.tiles);
    const int lane = threadIdx.x & 63, sb = lane & (subs - 1), chunk = cg * cpw + (lane >> sshift);
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    bool keep = false;
    if (cg < groups && chunk < chunks && tx * DF_RIGID_PX < a.X && ty * DF_RIGID_PY < a.Y) {
        const int zb = a.z_own0 + chunk * a.zc, ze = min(zb + a.zc, a.z_own0 + a.z_own_n);
        const int zs = zb + sb * DF_RIGID_SUB;
        if (zs < ze) keep = keep_all || !df_rigid_box_culled<DEPTH>(a, F, Py, tx * DF_RIGID_PX, ty * DF_RIGID_PY, zs, min(DF_RIGID_SUB, ze - zs));
    }
    const unsigned long long alive = __ballot(keep);                      // bits [cl * subs, (cl + 1) * subs): the patch's mask in chunk cg * cpw + cl
    if (lane == 0) s_alive[threadIdx.x >> 6] = alive;
    __syncthreads();
    // the workgroup's 16 waves are 16 / STRIP strips x cpw chunks of plan items: one thread per item
    const unsigned smask = (1u << subs) - 1u;
    unsigned m = 0, w = 0, slot = 0, item = 0;
    const int n_local = (16 / DF_RIGID_STRIP) * cpw;
    if ((int)threadIdx.x < n_local) {
        const int q = threadIdx.x / cpw, c = threadIdx.x % cpw;
#pragma unroll
        for (int k = 0; k < DF_RIGID_STRIP; ++k) m |= ((unsigned)(s_alive[DF_RIGID_STRIP * q + k] >> (c * subs)) & smask) << (8 * k);
        const unsigned wg0 = blockIdx.x * 16u + (unsigned)(DF_RIGID_STRIP * q);      // the strip's first wave
        const int scg = (int)(wg0 / (unsigned)a.tiles), strip = (int)(wg0 % (unsigned)a.tiles) / DF_RIGID_STRIP;
        item = (unsigned)(scg * cpw + c) * (unsigned)(a.tiles / DF_RIGID_STRIP) + (unsigned)strip;
        w = (unsigned)__popc(m);
        if (m) slot = atomicAdd(&s_cnt[w], 1u);
    }
    __syncthreads();
    if (threadIdx.x < DF_RIGID_BINS && s_cnt[threadIdx.x]) s_base[threadIdx.x] = atomicAdd(&cnt[threadIdx.x], s_cnt[threadIdx.x]);
    __syncthreads();
    if ((int)threadIdx.x < n_local && m) { bins[(size_t)w * n_items + s_base[w] + slot] = item; mask_out[item] = m; }
    if (alive) {                                                           // wave-uniform: the chunk starts of the patch's columns
        const int x = tx * DF_RIGID_PX + (lane & (DF_RIGID_PX - 1)), y = ty * DF_RIGID_PY + (lane / DF_RIGID_PX);
        const f3 zstep = scale3(mk3(a.vol2cam.R[2], a.vol2cam.R[5], a.vol2cam.R[8]), a.vsz);  // tsdf_volume.cu:69 (three separate multiplies)
        f3 vc = aff_mul(a.vol2cam, mk3((float)x * a.vsx, (float)y * a.vsy, 0.f));              // :71-72
        const int last = (63 - __clzll((long long)alive)) >> sshift;       // last alive chunk of the group
        int z = 0;
        for (int c = 0; c <= last; ++c) {
            const int zb = a.z_own0 + (cg * cpw + c) * a.zc;
            for (; z + 8 <= zb; z += 8) {                                  // :75, planes [0, zb)
#pragma unroll
                for (int i = 0; i < 8; ++i) vc = add3(vc, zstep);
            }
            for (; z < zb; ++z) vc = add3(vc, zstep);
            if ((alive >> (c * subs)) & (unsigned long long)smask)
                starts[((size_t)(cg * cpw + c) * a.tiles + tile) * 64 + lane] = make_float4(vc.x, vc.y, vc.z, 0.f);
        }
    }
}

// The sweep: wave e of the launch takes plan entry e (bins from the fullest down), replays `vc += zstep` up to its chunk, and walks
// the chunk's sub-chunks: alive ones U planes per batch, the others by the additions alone.
template <int U, bool SAT, bool COUNT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DF_RIGID_WAVES, DF_RIGID_WAVES))) void df_integrate_rigid_kernel(const DfRigidArgs a, const bool FASTOK)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // plan entry -> strip item: lane j < DF_RIGID_BINS - 1 holds the count of bin DF_RIGID_BINS - 1 - j and the running total
    const unsigned bin_cnt = lane < DF_RIGID_BINS - 1 ? a.plan_cnt[DF_RIGID_BINS - 1 - lane] : 0u;
    unsigned bin_end = bin_cnt;
#pragma unroll
    for (int o = 1; o < DF_RIGID_BINS; o <<= 1) { const unsigned t = __shfl_up(bin_end, o, 64); if (lane >= o) bin_end += t; }
    const unsigned n_alive = (unsigned)__builtin_amdgcn_readlane((int)bin_end, DF_RIGID_BINS - 2);
    const unsigned e = blockIdx.x * (4 / DF_RIGID_STRIP) + (unsigned)(wave / DF_RIGID_STRIP);      // plan entry: a strip of STRIP waves
    if (e >= n_alive) return;                                              // wave-uniform (no barrier below)
#ifdef DF_TRACE_WG
    const unsigned long long t_start = wall_clock64(); unsigned n_sub = 0;
#endif
    const int j = __ffsll((unsigned long long)__ballot(lane < DF_RIGID_BINS - 1 && e < bin_end)) - 1;
    const unsigned r = e - ((unsigned)__builtin_amdgcn_readlane((int)bin_end, j) - (unsigned)__builtin_amdgcn_readlane((int)bin_cnt, j));
    const unsigned sitem = (unsigned)__builtin_amdgcn_readfirstlane((int)a.plan_bins[(size_t)(DF_RIGID_BINS - 1 - j) * a.plan_items + r]);
    const unsigned mask = ((unsigned)__builtin_amdgcn_readfirstlane((int)a.plan_mask[sitem]) >> (8 * (wave % DF_RIGID_STRIP))) & 0xffu;
    if (mask == 0u) return;                                                // nothing alive in this wave's patch
    const int strips = a.tiles / DF_RIGID_STRIP;
    const int chunk = (int)(sitem / (unsigned)strips), tile = (int)(sitem % (unsigned)strips) * DF_RIGID_STRIP + (wave % DF_RIGID_STRIP);
    const unsigned item = (unsigned)chunk * (unsigned)a.tiles + (unsigned)tile;     // (patch item: indexes plan_starts)
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int x = tx * DF_RIGID_PX + (lane & (DF_RIGID_PX - 1)), y = ty * DF_RIGID_PY + (lane / DF_RIGID_PX);
    const bool active = x < a.X && y < a.Y;
    unsigned int my_upd = 0, my_swept = 0;
    const int zb = a.z_own0 + chunk * a.zc;
    const int ze = min(zb + a.zc, a.z_own0 + a.z_own_n);
    const f3 zstep = scale3(mk3(a.vol2cam.R[2], a.vol2cam.R[5], a.vol2cam.R[8]), a.vsz);      // tsdf_volume.cu:69 (three separate multiplies)
    // the column's running position at plane zb: vol2cam * (x, y, 0) + zb additions of zstep (:71-75), made once per column by
    // the plan kernel
    const float4 st4 = a.plan_starts[(size_t)item * 64 + lane];
    f3 vc = mk3(st4.x, st4.y, st4.z);
    const size_t plane = (size_t)a.X * a.Y;
    uint32_t* p = a.vol + (size_t)(zb - a.z_store0) * plane + (size_t)(active ? y : 0) * a.X + (active ? x : 0);
    int sb = 0;
    for (int zs = zb; zs < ze; zs += DF_RIGID_SUB, ++sb) {
        const int zse = min(zs + DF_RIGID_SUB, ze);
        if (!((mask >> sb) & 1u)) {
            if ((mask >> sb) == 0u) break;                                                      // nothing alive further on
            for (int z = zs; z < zse; ++z) vc = add3(vc, zstep);                                // :75, skipped voxels included
            p += (size_t)(zse - zs) * plane;
            continue;
        }
#ifdef DF_TRACE_WG
        ++n_sub;
#endif
        // the short arithmetic forms of tsdf_sample_fast need their domain on every voxel of the run, for every lane
        const f3 vc_end = add3(vc, scale3(zstep, (float)(zse - zs)));
        // (SAT: the saturated-sample shortcuts also want every coordinate within 32 m; the generic forms otherwise)
        const bool fast = FASTOK && df_wave_all(!active || (tsdf_sample_domain_ok(vc, vc_end) && (!SAT || tsdf_sat_domain_ok(vc, vc_end))));
        my_swept += (unsigned)(zse - zs);               // (wave-uniform; times the wave's columns inside the volume at the end)
        if (fast && SAT && (zse - zs) % U == 0) df_rigid_run_pipe<U, COUNT>(a, vc, zstep, p, plane, zse - zs, active, my_upd);
        else if (fast && zse - zs == DF_RIGID_SUB) df_rigid_batches<DF_RIGID_U_GENERIC, true, SAT, true, COUNT>(a, vc, zstep, p, plane, zs, zse, active, my_upd);
        else if (fast) df_rigid_batches<DF_RIGID_U_GENERIC, true, SAT, false, COUNT>(a, vc, zstep, p, plane, zs, zse, active, my_upd);
        else df_rigid_batches<DF_RIGID_U_GENERIC, false, false, false, COUNT>(a, vc, zstep, p, plane, zs, zse, active, my_upd);   // (rare: fewer chains in flight, fewer registers)
    }
#ifdef DF_TRACE_WG
    if (lane == 0) {
        unsigned long long* t = a.trace + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
        t[0] = t_start; t[1] = wall_clock64(); t[2] = 0; t[3] = n_sub;
    }
#endif
    if (COUNT && a.n_upd) {                                   // one atomic per wave
        unsigned int s = my_upd;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(a.n_upd, (unsigned long long)s);
    }
    if (COUNT && a.n_swept) {                                          // (validation / measurement hook: dfusion_integrate_ex n_swept_dev)
        const unsigned long long s = (unsigned long long)my_swept * (unsigned)__popcll(__builtin_amdgcn_ballot_w64(active));
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(a.n_swept, s);
    }
}

// the dists max-pyramid of one frame into `mem` (levels 1..top); returns the descriptor
int df_build_dists_pyramid(const uint16_t* dists, size_t pitch, int cols, int rows, uint16_t* mem, size_t mem_elems,
                           DfDistsPyramid* out, hipStream_t st, bool levels_to_5, unsigned int* zero16, unsigned int* max_word)
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
    if (levels_to_5 && max_word) { P.max_bits = max_word; P.capped = 1; }
    hipLaunchKernelGGL(df_pyramid_tiles_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, P, mem, zero16, P.capped ? max_word : nullptr);
    DF_LAUNCH_CHECK();
    if (P.top > 5 && !levels_to_5) {
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


// The rigid integrate's scratch, cached per (device, stream) and kept until dfusion_release_scratch(): at most DF_SCRATCH_MAX entries PER
// DEVICE, the least recently used idle one is evicted (a host that makes a stream per frame would otherwise leave ~84 MB behind per
// stream at 512^3).  A cached stream handle may have been destroyed by the host since: entries are freed after a DEVICE synchronise,
// never through the stored handle.
// Round 5: an entry is HELD for the whole of the call that uses it -- `busy` is locked from the lookup until the call has
// enqueued its last kernel -- so another host thread can neither evict nor regrow it in the window between the lookup and the launches
// (a device synchronise sees nothing of kernels that are not enqueued yet).  Eviction only considers entries it can try_lock; when every
// entry of the device is in use the cache grows past the cap instead.  Lock order: the table's mutex, then an entry's; nothing takes the
// table's mutex while holding an entry.
#define DF_SCRATCH_MAX 8
struct DfScratchEntry { int device; hipStream_t stream; char* mem; size_t cap; unsigned long long used; std::mutex busy; };
static std::mutex g_df_scratch_mutex;
static std::list<DfScratchEntry> g_df_scratch;                            // (a list: entries never move while a call holds one)
static unsigned long long g_df_scratch_clock = 0;
static void df_scratch_free_entry(DfScratchEntry& c)                       // caller holds c.busy
{
    if (!c.mem) return;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    if (hipSetDevice(c.device) == hipSuccess) { (void)hipDeviceSynchronize(); (void)hipFree(c.mem); }
    (void)hipGetLastError();
    if (have_cur) (void)hipSetDevice(cur);
    c.mem = nullptr; c.cap = 0;
}
// the entry of (current device, st), locked; nullptr on failure.  *mem_out = at least `bytes` of device memory.
static DfScratchEntry* df_rigid_scratch_acquire(hipStream_t st, size_t bytes, char** mem_out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    DfScratchEntry* e = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_df_scratch_mutex);
        for (DfScratchEntry& c : g_df_scratch) if (c.device == dev && c.stream == st) { e = &c; break; }
        if (!e) {
            size_t n_dev = 0;
            for (const DfScratchEntry& c : g_df_scratch) n_dev += c.device == dev;
            while (n_dev >= DF_SCRATCH_MAX) {                              // evict this device's least recently used IDLE entry
                std::list<DfScratchEntry>::iterator lru = g_df_scratch.end();
                for (auto it = g_df_scratch.begin(); it != g_df_scratch.end(); ++it)
                    if (it->device == dev && (lru == g_df_scratch.end() || it->used < lru->used) && it->busy.try_lock()) {
                        if (lru != g_df_scratch.end()) lru->busy.unlock();
                        lru = it;
                    }
                if (lru == g_df_scratch.end()) break;                      // every entry is in use: grow past the cap
                df_scratch_free_entry(*lru);
                lru->busy.unlock();
                g_df_scratch.erase(lru);
                --n_dev;
            }
            g_df_scratch.emplace_back();
            e = &g_df_scratch.back();
            e->device = dev; e->stream = st; e->mem = nullptr; e->cap = 0; e->used = 0;
        }
        e->used = ++g_df_scratch_clock;
        e->busy.lock();                                                    // (a second thread on the SAME stream waits here for the first call's enqueue)
    }
    if (bytes > e->cap) {
        if (e->mem) { (void)hipStreamSynchronize(st); (void)hipFree(e->mem); e->mem = nullptr; e->cap = 0; }
        const size_t cap = bytes + bytes / 4;
        if (hipMalloc((void**)&e->mem, cap) != hipSuccess) { (void)hipGetLastError(); e->mem = nullptr; e->busy.unlock(); return nullptr; }
        e->cap = cap;
    }
    *mem_out = e->mem;
    return e;
}
extern "C" int dfusion_release_scratch(void)
{
    std::lock_guard<std::mutex> lock(g_df_scratch_mutex);
    for (DfScratchEntry& c : g_df_scratch) { c.busy.lock(); df_scratch_free_entry(c); c.busy.unlock(); }      // (waits for calls in flight; restores the caller's device)
    g_df_scratch.clear();
    return DF_OK;
}

extern "C" int dfusion_integrate(const uint16_t* dists, size_t pitch, int cols, int rows, DfVolume v, const DfSlab* slab,
                                 const float vol2cam[12], const float proj[4], unsigned long long* n_updated,
                                 dfStream stream)
{
    return dfusion_integrate_ex(dists, pitch, cols, rows, v, slab, vol2cam, proj, 0u, n_updated, nullptr, stream);
}

extern "C" int dfusion_integrate_ex(const uint16_t* dists, size_t pitch, int cols, int rows, DfVolume v, const DfSlab* slab,
                                    const float vol2cam[12], const float proj[4], unsigned flags, unsigned long long* n_updated,
                                    unsigned long long* n_swept, dfStream stream)
{
    const bool no_depth_cull = (flags & DF_RIGID_NO_DEPTH_CULL) != 0, no_fast_forms = (flags & DF_RIGID_NO_SHORT_FORMS) != 0;
    const bool keep_all = (flags & DF_RIGID_KEEP_ALL) != 0, no_sat = (flags & DF_RIGID_NO_SAT) != 0;
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
    a.n_swept = n_swept;

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
    // column patches (one per wave), 4 side by side per strip (one per workgroup): the patch grid's x extent is padded to whole strips
    constexpr int tx_pad = DF_RIGID_STRIP;
    const int tiles_x = ((a.X + DF_RIGID_PX - 1) / DF_RIGID_PX + tx_pad - 1) / tx_pad * tx_pad;
    const int tiles = tiles_x * ((a.Y + DF_RIGID_PY - 1) / DF_RIGID_PY);
    // Z chunking: chunks of DF_RIGID_ZC planes -- 1, 2, 4 or 8 sub-chunks.  Short enough that the longest wave is a fraction of the
    // launch: a per-wave timeline (tools/trace_sweep.py, round 3) showed 64-plane items taking up to 93 us of a 115 us launch whose
    // waves summed to 71 us per slot -- the launch ended on a few long waves.  (Chunk starts come from the plan kernel: a short
    // chunk does not pay a long replay.)
    int zc = DF_RIGID_ZC;
    while ((s.z_own_n + zc - 1) / zc > 4096 && zc < DF_RIGID_SUB * DF_RIGID_MAX_SUBS) zc *= 2;      // (very deep slabs)
    a.zc = zc;
    const int chunks = (s.z_own_n + zc - 1) / zc;
    const unsigned n_pitems = (unsigned)tiles * (unsigned)chunks;        // patch items (chunk starts)
    const unsigned n_items = n_pitems / DF_RIGID_STRIP;                  // strip items (the plan's entries)
    a.tiles = tiles; a.tiles_x = tiles_x; a.plan_items = n_items;
    hipStream_t st = (hipStream_t)stream;
    // scratch: the launch plan, and for the behind-the-surface test a max-pyramid of this frame's dists
    const size_t pyr_elems = no_depth_cull ? 0 : df_pyramid_elems(cols, rows);
    if (n_pitems >= (1u << 30)) return DF_E_INVALID;
    const size_t off_cnt = 0, off_bins = 256, off_mask = off_bins + (size_t)DF_RIGID_BINS * n_items * 4;
    const size_t off_pyr = (off_mask + (size_t)n_items * 4 + 15) / 16 * 16;
    const size_t off_starts = (off_pyr + pyr_elems * sizeof(uint16_t) + 255) / 256 * 256;
    const size_t bytes = off_starts + (size_t)n_pitems * 64 * sizeof(float4);
    // One scratch buffer per (device, stream), grown on demand and kept: calls on a stream are ordered, so the next call's kernels
    // cannot start before this call's have finished with it.  (Round 3 tried the runtime's stream-ordered allocator here,
    // hipMallocAsync / hipFreeAsync per call: in a process that also allocates and frees with hipMalloc / hipFree between the calls --
    // the host mirror's reference-shaped flow -- one integrate in ~30 then updated a different set of voxels from identical inputs;
    // with a plain or a kept allocation never, 200 runs each.)
    char* scratch = nullptr;
    DfScratchEntry* scratch_entry = df_rigid_scratch_acquire(st, bytes, &scratch);       // held until the last launch below is enqueued
    if (!scratch_entry) return (int)hipErrorOutOfMemory;
    struct ScratchHold { DfScratchEntry* e; ~ScratchHold() { e->busy.unlock(); } } scratch_hold{scratch_entry};
    // validation: the kept buffer still holds the previous call's plan -- exactly what would hide a read of plan data this call did
    // not write.  Poisoned (every byte 0xFF: NaN starts, out-of-range items, full masks), such a read cannot go unnoticed.
    if (flags & DF_RIGID_POISON_SCRATCH) DF_HIP(hipMemsetAsync(scratch, 0xFF, bytes, st));
    auto release_scratch = [&]() {};                                       // (scratch_hold's destructor: every return path)
    DfDistsPyramid Py;
    memset(&Py, 0, sizeof(Py));
    int rc = DF_OK;
    // (levels 1..5 only: the plan reads none above; the pyramid's first workgroup zeroes the plan's counters)
    if (pyr_elems) rc = df_build_dists_pyramid(dists, pitch, cols, rows, (uint16_t*)(scratch + off_pyr), pyr_elems, &Py, st, true, (unsigned int*)(scratch + off_cnt));
    if (rc == DF_OK && Py.top == 0 && hipMemsetAsync(scratch + off_cnt, 0, 256, st) != hipSuccess) rc = (int)hipGetLastError();
    if (rc != DF_OK) { release_scratch(); return rc; }
    unsigned int* cnt = (unsigned int*)(scratch + off_cnt);
    unsigned int* bins = (unsigned int*)(scratch + off_bins);
    unsigned int* pmask = (unsigned int*)(scratch + off_mask);
    float4* starts = (float4*)(scratch + off_starts);
    const int cpw = 64 / (zc / DF_RIGID_SUB);
    const unsigned plan_waves = (unsigned)tiles * (unsigned)((chunks + cpw - 1) / cpw);
    if (Py.top) hipLaunchKernelGGL((df_rigid_plan_kernel<true>), dim3((plan_waves + 15) / 16), dim3(1024), 0, st, a, F, Py, n_items, chunks, keep_all, cnt, bins, pmask, starts);
    else hipLaunchKernelGGL((df_rigid_plan_kernel<false>), dim3((plan_waves + 15) / 16), dim3(1024), 0, st, a, F, Py, n_items, chunks, keep_all, cnt, bins, pmask, starts);
    a.plan_bins = bins; a.plan_cnt = cnt; a.plan_mask = pmask; a.plan_starts = starts;
    // short arithmetic forms (tsdf_sample_fast): 32-bit dists offsets, sane intrinsics; the value domain is tested per run in the kernel
    const bool fast_ok = (unsigned long long)rows * pitch < (1ull << 31) && proj[0] == proj[0] && proj[1] == proj[1] && proj[2] > 0.f && proj[3] > 0.f &&
                         !no_fast_forms;          // (cx, cy > 0: the one-compare pixel range test of tsdf_sample_fast)
    const dim3 grid((n_items * DF_RIGID_STRIP + 3) / 4);             // 4 waves per workgroup; those past the plan's end return at once
#ifdef DF_TRACE_WG
    static unsigned long long* trace_dev = nullptr; static size_t trace_cap = 0;
    const size_t trace_n = (size_t)grid.x * 4 * 4;
    if (trace_n > trace_cap) { (void)hipFree(trace_dev); DF_HIP(hipMalloc((void**)&trace_dev, trace_n * 8)); trace_cap = trace_n; }
    DF_HIP(hipMemsetAsync(trace_dev, 0, trace_n * 8, st));
    a.trace = trace_dev;
#endif
    const bool sat_ok = fast_ok && !no_sat && df_sat_trunc_ok(v.trunc_dist);
    const bool count = a.n_upd || a.n_swept;
    if (sat_ok && !count) hipLaunchKernelGGL((df_integrate_rigid_kernel<DF_RIGID_U, true, false>), grid, dim3(256), 0, st, a, fast_ok);
    else if (sat_ok) hipLaunchKernelGGL((df_integrate_rigid_kernel<DF_RIGID_U, true, true>), grid, dim3(256), 0, st, a, fast_ok);
    else if (!count) hipLaunchKernelGGL((df_integrate_rigid_kernel<DF_RIGID_U, false, false>), grid, dim3(256), 0, st, a, fast_ok);
    else hipLaunchKernelGGL((df_integrate_rigid_kernel<DF_RIGID_U, false, true>), grid, dim3(256), 0, st, a, fast_ok);
    rc = (int)hipGetLastError();
#ifdef DF_TRACE_WG
    if (getenv("DF_TRACE_FILE")) {
        DF_HIP(hipStreamSynchronize(st));
        unsigned long long* h = (unsigned long long*)malloc(trace_n * 8);
        DF_HIP(hipMemcpy(h, trace_dev, trace_n * 8, hipMemcpyDeviceToHost));
        FILE* f = fopen(getenv("DF_TRACE_FILE"), "wb");
        if (f) { unsigned long long hdr[4] = {grid.x, 1, 4, 0}; fwrite(hdr, 8, 4, f); fwrite(h, 8, trace_n, f); fclose(f); }
        free(h);
    }
#endif
    release_scratch();
    return rc;
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
