// dfusion_selftest.hip -- exhaustive / randomised device checks of the "same bits, fewer instructions" building blocks of the warped
// sweep (dfusion_device.h) against the generic forms they stand in for.  Not on any product path; the parity tests call it so that
// the exactness claims rest on every input of the domains, not only on the voxels a synthetic scene happens to produce.
#include "dfusion_internal.h"

#pragma clang fp contract(off)

__device__ __forceinline__ bool st_same(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }

// [0] df_sqrt_short vs sqrtf over EVERY f32 in its domain (finite x >= 2^-96)
// [1] df_rcp_short((double)n) vs 1.0 / (double)n over EVERY positive normal finite f32 n
__global__ __launch_bounds__(256) void df_selftest_scan_kernel(unsigned long long* __restrict__ counts)
{
    unsigned long long bad0 = 0, bad1 = 0;
    const unsigned stride = gridDim.x * 256u;
    for (unsigned long long b = blockIdx.x * 256u + threadIdx.x; b < 0x7f800000ull; b += stride) {
        const float x = __uint_as_float((unsigned)b);
        if (df_sqrt_short_ok(x)) bad0 += !st_same(df_sqrt_short(x), sqrtf(x));
        if (b >= 0x00800000ull) {
            const double d = (double)x;
            bad1 += __double_as_longlong(df_rcp_short(d)) != __double_as_longlong(1.0 / d);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { bad0 += __shfl_xor(bad0, o, 64); bad1 += __shfl_xor(bad1, o, 64); }
    if ((threadIdx.x & 63) == 0) { if (bad0) atomicAdd(&counts[0], bad0); if (bad1) atomicAdd(&counts[1], bad1); }
}

__device__ __forceinline__ unsigned st_rng(unsigned long long& s)
{
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (unsigned)(s >> 32);
}
// a float with a random sign, an exponent drawn around 1 (or anywhere, or a special value every so often) and random mantissa
__device__ __forceinline__ float st_float(unsigned long long& s)
{
    const unsigned r = st_rng(s), m = st_rng(s);
    const unsigned kind = r & 31u;
    if (kind == 0) return 0.f;
    if (kind == 1) return -0.f;
    if (kind == 2) return __uint_as_float(0x7f800000u | ((r >> 8) & 0x80000000u));            // +-inf
    if (kind == 3) return __uint_as_float(0x7fc00000u);                                         // NaN
    if (kind == 4) return __uint_as_float((m & 0x807fffffu));                                   // denormal
    if (kind < 10) return __uint_as_float(m);                                                   // anything
    const unsigned e = 120u + ((r >> 8) % 14u);                                                 // 2^-7 .. 2^6
    return __uint_as_float((m & 0x807fffffu) | (e << 23));
}

// [2] q_mul_pk / q_mul_conj_pk vs q_mul / q_mul(a, q_conj(b)) on `n` pseudo-random quaternion pairs (specials included)
// [3] q_normalize_near_unit vs q_normalize on quaternions normalised once already (the sweep's use), every s it accepts
// [4] number of near-unit cases actually compared in [3] (so that the test can see the domain was exercised)
__global__ __launch_bounds__(256) void df_selftest_quat_kernel(unsigned long long n, unsigned long long* __restrict__ counts)
{
    unsigned long long bad2 = 0, bad3 = 0, seen = 0;
    unsigned long long seed = 0x9e3779b97f4a7c15ull * (blockIdx.x * 256ull + threadIdx.x + 1ull);
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256ull) {
        quat a, b;
        a.w = st_float(seed); a.x = st_float(seed); a.y = st_float(seed); a.z = st_float(seed);
        b.w = st_float(seed); b.x = st_float(seed); b.y = st_float(seed); b.z = st_float(seed);
        const quat r0 = q_mul(a, b), r1 = q_unpair(q_mul_pk(q_pairs(a), q_pairs(b)));
        const quat c0 = q_mul(a, q_conj(b)), c1 = q_unpair(q_mul_conj_pk(q_pairs(a), q_pairs(b)));
        bad2 += !(st_same(r0.w, r1.w) && st_same(r0.x, r1.x) && st_same(r0.y, r1.y) && st_same(r0.z, r1.z));
        bad2 += !(st_same(c0.w, c1.w) && st_same(c0.x, c1.x) && st_same(c0.y, c1.y) && st_same(c0.z, c1.z));
        const quat u = q_normalize(a);
        const float s = q_sumsq(u);
        if (q_near_unit_ok(s)) {
            ++seen;
            const quat g = q_normalize(u), f = q_normalize_near_unit(u, s);
            bad3 += !(st_same(g.w, f.w) && st_same(g.x, f.x) && st_same(g.y, f.y) && st_same(g.z, f.z));
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { bad2 += __shfl_xor(bad2, o, 64); bad3 += __shfl_xor(bad3, o, 64); seen += __shfl_xor(seen, o, 64); }
    if ((threadIdx.x & 63) == 0) {
        if (bad2) atomicAdd(&counts[2], bad2);
        if (bad3) atomicAdd(&counts[3], bad3);
        atomicAdd(&counts[4], seen);
    }
}

// [5] tsdf_fuse_short vs tsdf_fuse on EVERY finite stored half x weights 0..64 and a spread up to 65535 x `per` tsdf values each
//     (the values a sample can produce: min(1, sdf / trunc) -- random in [-2, 1], tiny ones of both signs, zeros, 1)
__global__ __launch_bounds__(256) void df_selftest_fuse_kernel(unsigned per, unsigned long long* __restrict__ counts)
{
    unsigned long long bad = 0;
    unsigned long long seed = 0xd1b54a32d192ed03ull * (blockIdx.x * 256ull + threadIdx.x + 1ull);
    const unsigned n_w = 65 + 32;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < 65536ull * n_w; i += (unsigned long long)gridDim.x * 256ull) {
        const unsigned h = (unsigned)(i & 0xffffu), wi = (unsigned)(i >> 16);
        const unsigned w = wi < 65 ? wi : 65u + (st_rng(seed) % 65471u);
        const uint32_t vox = h | (w << 16);
        if (!tsdf_fuse_short_ok(vox)) continue;
        for (unsigned j = 0; j < per; ++j) {
            const unsigned r = st_rng(seed), m = st_rng(seed);
            float t;
            switch (r & 7u) {
                case 0: t = 0.f; break;
                case 1: t = 1.f; break;
                case 2: t = __uint_as_float((m & 0x807fffffu) | ((90u + (r >> 8) % 37u) << 23)); break;      // 2^-37 .. 2^-1, both signs
                case 3: t = -h2f_bits((uint16_t)h) * (float)w; break;                                      // cancels the stored sum exactly
                case 4: t = -h2f_bits((uint16_t)h) * (float)w * (1.f + (float)((int)(m & 15u) - 8) * 0x1p-23f); break;   // ... nearly
                default: t = fminf(1.f, (float)(int)(m >> 8) * 0x1p-23f * 1.5f - 2.f + (float)(m & 255u) * 0x1p-31f); break;   // [-2, 1]
            }
            bad += tsdf_fuse_short(vox, t, 64) != tsdf_fuse(vox, t, 64);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) bad += __shfl_xor(bad, o, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(&counts[5], bad);
}

extern "C" int dfusion_selftest_exact_forms(unsigned long long n_random, unsigned long long* counts_dev, dfStream stream)
{
    if (!counts_dev) return DF_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    DF_HIP(hipMemsetAsync(counts_dev, 0, 6 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(df_selftest_scan_kernel, dim3(4096), dim3(256), 0, st, counts_dev);
    DF_LAUNCH_CHECK();
    hipLaunchKernelGGL(df_selftest_quat_kernel, dim3(2048), dim3(256), 0, st, n_random, counts_dev);
    DF_LAUNCH_CHECK();
    hipLaunchKernelGGL(df_selftest_fuse_kernel, dim3(2048), dim3(256), 0, st, (unsigned)(n_random >> 21 ? n_random >> 21 : 1), counts_dev);
    DF_LAUNCH_CHECK();
    return DF_OK;
}
