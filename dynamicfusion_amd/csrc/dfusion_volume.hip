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
#include <math.h>

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

// ------------------------------------------------------------------------------------------ integrate (rigid)
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

// U = planes handled per batch.  Measured on MI355X (512^3): the sweep is bound by DEPENDENT LATENCY, not by VALU
// throughput or HBM bandwidth -- each plane is a chain  divide -> dists fetch (L2) -> sqrt/compare -> voxel load (HBM)
// -> fuse -> store,  and all 8192 waves of the launch are resident at once (8 per SIMD), so there is nothing else to
// switch to.  Hence the straight-line, branch-free sample (4 columns x U planes = 4U independent chains per lane) and
// the batched voxel loads: U dists-fetch groups and U 16-byte voxel loads are in flight per lane instead of one.
template <int U>
__global__ __launch_bounds__(256) void df_integrate_rigid_kernel(const DfRigidArgs a, const DfFrustum F)
{
    const int xgroups = a.X >> 2;
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const bool active = gid < xgroups * a.Y;
    unsigned int my_upd = 0;
    if (active) {
        const int y = gid / xgroups;
        const int x0 = (gid - y * xgroups) << 2;
        const int zb = a.z_own0 + blockIdx.y * a.zc;
        const int ze = min(zb + a.zc, a.z_own0 + a.z_own_n);

        // tsdf_volume.cu:69 (three separate multiplies)
        const f3 zstep = scale3(mk3(a.vol2cam.R[2], a.vol2cam.R[5], a.vol2cam.R[8]), a.vsz);
        f3 vc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            vc[i] = aff_mul(a.vol2cam, mk3((float)(x0 + i) * a.vsx, (float)y * a.vsy, 0.f));     // :71-72
        // Conservative, result-identical chunk rejection: the chunk's voxels of columns x0..x0+3 lie (up to the
        // accumulated rounding of `vc += zstep`, < 1e-3 m over 1024 planes) in the convex hull of the four points
        // {column 0, column 3} x {plane zb, plane ze-1}.  If all four are outside the SAME frustum plane by 5 mm,
        // no voxel of the chunk can pass the exact test (:82,:86): skip it, including the replay.
        bool culled;
        {
            const f3 a0 = add3(vc[0], scale3(zstep, (float)zb)), a3 = add3(vc[3], scale3(zstep, (float)zb));
            const f3 span = scale3(zstep, (float)(ze - 1 - zb));
            const float m = 5e-3f;
            culled = (df_outside_mask(F, a0, m) & df_outside_mask(F, a3, m) & df_outside_mask(F, add3(a0, span), m) &
                      df_outside_mask(F, add3(a3, span), m)) != 0u;
        }
        if (!culled) {
            for (int z = 0; z < zb; ++z) {                  // replay of `vc += zstep` (:75) for planes [0, zb)
#pragma unroll
                for (int i = 0; i < 4; ++i) vc[i] = add3(vc[i], zstep);
            }
            const size_t plane = (size_t)a.X * a.Y;
            uint32_t* p = a.vol + (size_t)(zb - a.z_store0) * plane + (size_t)y * a.X + x0;
            for (int z = zb; z < ze; z += U, p += U * plane) {
                float ts[U][4];
                bool up[U][4], any[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {               // stage 1: 4U branch-free sample chains
                    const bool inr = z + u < ze;
                    any[u] = false;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        up[u][i] = tsdf_sample_nb(a.P, vc[i], &ts[u][i]) && inr;
                        any[u] |= up[u][i];
                        vc[i] = add3(vc[i], zstep);          // :75
                    }
                }
                uint4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)                 // stage 2: all voxel loads of the batch in flight together
                    if (any[u]) v[u] = *reinterpret_cast<const uint4*>(p + u * plane);
#pragma unroll
                for (int u = 0; u < U; ++u) {               // stage 3: fuse (:97-103) and store
                    if (any[u]) {
                        if (up[u][0]) { v[u].x = tsdf_fuse(v[u].x, ts[u][0], a.P.max_weight); ++my_upd; }
                        if (up[u][1]) { v[u].y = tsdf_fuse(v[u].y, ts[u][1], a.P.max_weight); ++my_upd; }
                        if (up[u][2]) { v[u].z = tsdf_fuse(v[u].z, ts[u][2], a.P.max_weight); ++my_upd; }
                        if (up[u][3]) { v[u].w = tsdf_fuse(v[u].w, ts[u][3], a.P.max_weight); ++my_upd; }
                        *reinterpret_cast<uint4*>(p + u * plane) = v[u];
                    }
                }
            }
        }
    }
    if (a.n_upd) {                                            // one atomic per wave
        unsigned int s = my_upd;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(a.n_upd, (unsigned long long)s);
    }
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
    const int groups = (a.X / 4) * a.Y;
    const int bx = (groups + 255) / 256;
    // Z chunking: enough chunks for >= ~8 waves per SIMD over 256 CUs, chunks of >= 16 planes.
    const long long want_blocks = 256LL * 8;                // 8 blocks of 4 waves per CU
    int chunks = (int)((want_blocks + bx - 1) / bx);
    if (chunks < 1) chunks = 1;
    int zc = (s.z_own_n + chunks - 1) / chunks;
    if (zc < 16) zc = 16;
    a.zc = zc;
    dim3 grid(bx, (s.z_own_n + zc - 1) / zc);
    hipLaunchKernelGGL(df_integrate_rigid_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a, F);   // 2 planes per load batch (measured best of 1/2/4)
    DF_LAUNCH_CHECK();
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
