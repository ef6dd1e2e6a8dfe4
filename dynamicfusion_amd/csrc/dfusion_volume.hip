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

template <int UNROLL>
__global__ __launch_bounds__(256) void df_integrate_rigid_kernel(const DfRigidArgs a)
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
        for (int z = 0; z < zb; ++z) {                      // replay of `vc += zstep` (:75) for planes [0, zb)
#pragma unroll
            for (int i = 0; i < 4; ++i) vc[i] = add3(vc[i], zstep);
        }

        const size_t plane = (size_t)a.X * a.Y;
        uint32_t* p = a.vol + (size_t)(zb - a.z_store0) * plane + (size_t)y * a.X + x0;
#pragma unroll UNROLL
        for (int z = zb; z < ze; ++z, p += plane) {
            float ts[4];
            bool up[4];
            bool any = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) { up[i] = tsdf_sample(a.P, vc[i], &ts[i]); any |= up[i]; }
            if (any) {
                uint4 v = *reinterpret_cast<const uint4*>(p);
                if (up[0]) { v.x = tsdf_fuse(v.x, ts[0], a.P.max_weight); ++my_upd; }
                if (up[1]) { v.y = tsdf_fuse(v.y, ts[1], a.P.max_weight); ++my_upd; }
                if (up[2]) { v.z = tsdf_fuse(v.z, ts[2], a.P.max_weight); ++my_upd; }
                if (up[3]) { v.w = tsdf_fuse(v.w, ts[3], a.P.max_weight); ++my_upd; }
                *reinterpret_cast<uint4*>(p) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) vc[i] = add3(vc[i], zstep);
        }
    }
    if (a.n_upd) {                                            // one atomic per wave
        unsigned int s = my_upd;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(a.n_upd, (unsigned long long)s);
    }
}

static int df_env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
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

    const int groups = (a.X / 4) * a.Y;
    const int bx = (groups + 255) / 256;
    // Z chunking: enough chunks for >= ~8 waves per SIMD over 256 CUs, chunks of >= 16 planes.
    int zc = df_env_int("DFUSION_RIGID_ZCHUNK", 0);
    if (zc <= 0) {
        const long long want_blocks = 256LL * 8;            // 8 blocks of 4 waves per CU
        int chunks = (int)((want_blocks + bx - 1) / bx);
        if (chunks < 1) chunks = 1;
        zc = (s.z_own_n + chunks - 1) / chunks;
        if (zc < 16) zc = 16;
    }
    a.zc = zc;
    dim3 grid(bx, (s.z_own_n + zc - 1) / zc);
    const int unroll = df_env_int("DFUSION_RIGID_UNROLL", 2);
    if (unroll >= 4)
        hipLaunchKernelGGL(df_integrate_rigid_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (unroll >= 2)
        hipLaunchKernelGGL(df_integrate_rigid_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(df_integrate_rigid_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
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
