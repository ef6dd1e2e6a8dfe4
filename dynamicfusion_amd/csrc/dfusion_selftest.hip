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

// [6] the projective sample in its short forms vs tsdf_sample_nb (the reference's statements with the generic division / sqrtf):
//     tsdf_sample_fast (shared refined reciprocal, short sqrtf, one-compare pixel test) and the two-stage form of the rigid sweep
//     (tsdf_sample_pre + the saturation decision on the approximate root, tsdf_sample_finish otherwise) -- verdict AND tsdf bits.
//     Positions are random inside the forms' domain and ON ITS EDGES (|x|, |y| = 2^-20 and 2^30, z = 0.05 and 2^30, the 32 m limit of
//     the saturation shortcut), signs both ways, against a synthetic dists image holding every kind of half (zeros, tiny, finite,
//     65504, inf, NaN); trunc random in [2^-10, 2^10].   [7] = how many of the compared samples took the update branch.
__global__ __launch_bounds__(256) void df_selftest_sample_kernel(unsigned long long n, const uint16_t* __restrict__ img, unsigned long long* __restrict__ counts)
{
    unsigned long long bad = 0, upd = 0;
    unsigned long long seed = 0xa0761d6478bd642full * (blockIdx.x * 256ull + threadIdx.x + 1ull);
    DfIntegrateParams P;
    P.dists = img; P.pitch = 64 * 2; P.cols = 64; P.rows = 48; P.fx = 57.0342f; P.fy = 57.0342f; P.cx = 32.f; P.cy = 24.f; P.max_weight = 64;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256ull) {
        const unsigned r = st_rng(seed);
        P.trunc = ldexpf(1.f + (float)(st_rng(seed) >> 9) * 0x1p-23f, (int)(r % 20u) - 10);   // [2^-10, 2^10)
        if ((r >> 8) % 7u == 0) P.trunc = (r & 1u) ? 0x1p-10f : 0x1p10f;
        P.trunc_inv = 1.f / P.trunc;
        auto coord = [&](bool is_z) {
            const unsigned k = st_rng(seed), m = st_rng(seed);
            float v;
            switch (k & 7u) {
                case 0: v = is_z ? 0.05f : 0x1p-20f; break;                                         // lower edge
                case 1: v = 0x1p30f; break;                                                         // upper edge
                case 2: v = 32.f; break;                                                            // the saturation shortcut's limit
                case 3: v = __uint_as_float((m & 0x007fffffu) | ((127u + (k >> 8) % 6u) << 23)); break;          // 1 .. 64 m
                default: v = __uint_as_float((m & 0x007fffffu) | ((120u + (k >> 8) % 9u) << 23)); break;         // 2^-7 .. 4 m
            }
            if (is_z && v < 0.05f) v = 0.05f + v;
            return (!is_z && (k & 0x80000000u)) ? -v : v;
        };
        const f3 vc = mk3(coord(false), coord(false), coord(true));
        if (!tsdf_sample_domain_ok(vc, vc)) continue;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        const bool u0 = tsdf_sample_nb(P, vc, &t0);
        const bool u1 = tsdf_sample_fast(P, vc, &t1);
        bad += (u0 != u1) || (u0 && !st_same(t0, t1));
        if (df_sat_trunc_ok(P.trunc) && tsdf_sat_domain_ok(vc, vc)) {
            const DfSamplePre pre = tsdf_sample_pre(P, vc);
            const float sdf_a = pre.Dp - pre.s, T = df_sat_threshold(P.trunc);
            bool u2;
            if (!(pre.ok & (fabsf(sdf_a) < T))) { u2 = pre.ok & (sdf_a >= T); t2 = 1.f; }         // decided on the approximate root
            else u2 = tsdf_sample_finish(P, pre, &t2);
            bad += (u0 != u2) || (u0 && !st_same(t0, t2));
        }
        upd += u0;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { bad += __shfl_xor(bad, o, 64); upd += __shfl_xor(upd, o, 64); }
    if ((threadIdx.x & 63) == 0) { if (bad) atomicAdd(&counts[6], bad); atomicAdd(&counts[7], upd); }
}
// [8] the first normalisation as an f32 division (q_div_f32) vs the reference's (float)((1.0 / (double)n) * (double)c), on every
//     quaternion q_div_f32_ok accepts out of (i) random quaternions normalised the way the sweep does it (n = sqrtf(sumsq)), specials,
//     zeros of both signs and denormal components included, (ii) components drawn at and around the domain's edges: |c| = 2^-100
//     exactly and one ulp either side, n^2 at 2^-96 and 2^40, quotients down to 2^-120, components that are exact multiples of n
//     (quotients with no rounding at all) and one ulp off them.   [9] = how many quaternions were compared.
__global__ __launch_bounds__(256) void df_selftest_normdiv_kernel(unsigned long long n, unsigned long long* __restrict__ counts)
{
    unsigned long long bad = 0, seen = 0;
    unsigned long long seed = 0x94d049bb133111ebull * (blockIdx.x * 256ull + threadIdx.x + 1ull);
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256ull) {
        quat a;
        const unsigned kind = st_rng(seed);
        if ((kind & 3u) != 0u) {
            a.w = st_float(seed); a.x = st_float(seed); a.y = st_float(seed); a.z = st_float(seed);
            if ((kind & 12u) == 0u) {                                       // the scale of blend sums far from the nodes: 2^-47 .. 2^-2
                const float sc = __uint_as_float((80u + (kind >> 8) % 46u) << 23);
                a.w *= sc; a.x *= sc; a.y *= sc; a.z *= sc;
            }
        } else {
            auto edge = [&]() {
                const unsigned k = st_rng(seed), m = st_rng(seed);
                float v;
                switch (k & 7u) {
                    case 0: v = 0x1p-100f; break;
                    case 1: v = __uint_as_float(0x0d800000u - 1u); break;                                   // one ulp below the edge (must be refused)
                    case 2: v = __uint_as_float(0x0d800000u + 1u); break;
                    case 3: v = 0.f; break;
                    case 4: v = __uint_as_float((m & 0x007fffffu) | ((27u + (k >> 8) % 8u) << 23)); break;  // 2^-100 .. 2^-93
                    case 5: v = 0x1p-48f; break;                                                             // n^2 = 2^-96 when alone
                    case 6: v = 0x1p20f * (1.f - (float)(m & 3u) * 0x1p-24f); break;                        // n^2 ~ 2^40
                    default: v = __uint_as_float((m & 0x007fffffu) | ((100u + (k >> 8) % 40u) << 23)); break;
                }
                return (k & 0x80000000u) ? -v : v;
            };
            a.w = edge(); a.x = edge(); a.y = edge(); a.z = edge();
            if ((kind & 0x30u) == 0u) {                                     // exact multiples of the norm: w = 3 t, x = 4 t => n = 5 t (and off by one ulp)
                const float t = __uint_as_float((st_rng(seed) & 0x007fffffu) | ((110u + (kind >> 8) % 30u) << 23));
                a.w = 3.f * t; a.x = 4.f * t; a.y = 0.f; a.z = (kind & 0x40u) ? 0.f : -0.f;
                if (kind & 0x80u) a.x = __uint_as_float(__float_as_uint(a.x) + 1u);
            }
        }
        // (-0 is outside the form's domain and cannot reach it in the sweep: the blend sums start at +0 and an IEEE sum is -0 only when
        // both terms are -- q_div_f32's comment.  With a -0 component the two forms differ in the sign of that zero and nowhere else.)
        if ((__float_as_uint(a.w) == 0x80000000u) | (__float_as_uint(a.x) == 0x80000000u) | (__float_as_uint(a.y) == 0x80000000u) |
            (__float_as_uint(a.z) == 0x80000000u)) continue;
        const float s = q_sumsq(a);
        if (!q_div_f32_ok(a, s)) continue;
        ++seen;
        const float nrm = sqrtf(s);
        const quat g = q_scale_f64(1.0 / (double)nrm, a), f = q_div_f32(a, nrm);
        bad += !(st_same(g.w, f.w) && st_same(g.x, f.x) && st_same(g.y, f.y) && st_same(g.z, f.z));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { bad += __shfl_xor(bad, o, 64); seen += __shfl_xor(seen, o, 64); }
    if ((threadIdx.x & 63) == 0) { if (bad) atomicAdd(&counts[8], bad); atomicAdd(&counts[9], seen); }
}
__global__ __launch_bounds__(256) void df_selftest_sample_image_kernel(uint16_t* __restrict__ img)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= 64u * 48u) return;
    unsigned long long seed = 0x2545f4914f6cdd1dull * (i + 1ull);
    const unsigned r = st_rng(seed);
    uint16_t h;
    switch (r & 15u) {
        case 0: h = 0; break;                                  // invalid depth
        case 1: h = 0x7bff; break;                             // 65504
        case 2: h = 0x7c00; break;                             // +inf
        case 3: h = 0x7e00; break;                             // NaN
        case 4: h = (uint16_t)(r >> 16) & 0x03ff; break;       // subnormal halves
        case 5: h = 0x8000 | ((uint16_t)(r >> 16) & 0x7bff); break;   // negative lengths (never produced by compute_dists, legal input)
        default: h = (uint16_t)(0x3000 + ((r >> 8) % 0x2400u)); break;        // 0.125 .. 64 m
    }
    img[i] = h;
}

extern "C" int dfusion_selftest_exact_forms(unsigned long long n_random, unsigned long long* counts_dev, dfStream stream)
{
    if (!counts_dev) return DF_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    DF_HIP(hipMemsetAsync(counts_dev, 0, 10 * sizeof(unsigned long long), st));
    {
        uint16_t* img = nullptr;
        DF_HIP(hipMalloc((void**)&img, 64 * 48 * sizeof(uint16_t)));
        hipLaunchKernelGGL(df_selftest_sample_image_kernel, dim3(12), dim3(256), 0, st, img);
        hipLaunchKernelGGL(df_selftest_sample_kernel, dim3(2048), dim3(256), 0, st, n_random, (const uint16_t*)img, counts_dev);
        const hipError_t e = hipGetLastError();
        (void)hipStreamSynchronize(st); (void)hipFree(img);                // (a test entry point: the wait costs nothing that matters)
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(df_selftest_scan_kernel, dim3(4096), dim3(256), 0, st, counts_dev);
    DF_LAUNCH_CHECK();
    hipLaunchKernelGGL(df_selftest_quat_kernel, dim3(2048), dim3(256), 0, st, n_random, counts_dev);
    DF_LAUNCH_CHECK();
    hipLaunchKernelGGL(df_selftest_fuse_kernel, dim3(2048), dim3(256), 0, st, (unsigned)(n_random >> 21 ? n_random >> 21 : 1), counts_dev);
    DF_LAUNCH_CHECK();
    hipLaunchKernelGGL(df_selftest_normdiv_kernel, dim3(2048), dim3(256), 0, st, n_random, counts_dev);
    DF_LAUNCH_CHECK();
    return DF_OK;
}
