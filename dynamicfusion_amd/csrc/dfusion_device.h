// dfusion_device.h -- device-side arithmetic shared by the gfx950 kernels.
//
// Every routine states the reference line it reproduces.  The whole library is compiled with
// -ffp-contract=off: a fused multiply-add appears ONLY where the reference writes __fmaf_rn / dot()
// (explicit fmaf below); '/' and sqrtf are hipcc's correctly rounded fp32 forms; f32 denormals are
// preserved (hipcc default on gfx950).  That makes the kernels bit-compatible with the IEEE CPU
// restatement in oracle/dfusion_oracle.c, which is how parity is tested.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dfusion.h"

#pragma clang fp contract(off)

#define DF_WAVE 64

struct f3 { float x, y, z; };
struct quat { float w, x, y, z; };      // utils::Quaternion<float> (w_, x_, y_, z_)

struct DfAff { float R[9]; float t[3]; };   // device::Aff3f, internal.hpp:26-27

__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
// temp_utils.hpp:27-30
__device__ __forceinline__ float dot3(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
// device.hpp:71-72
__device__ __forceinline__ f3 mat3_mul(const float* R, f3 v)
{
    return mk3(dot3(mk3(R[0], R[1], R[2]), v), dot3(mk3(R[3], R[4], R[5]), v), dot3(mk3(R[6], R[7], R[8]), v));
}
// device.hpp:74
__device__ __forceinline__ f3 aff_mul(const DfAff& A, f3 v) { return add3(mat3_mul(A.R, v), mk3(A.t[0], A.t[1], A.t[2])); }
// temp_utils.hpp:97-100 (rsqrt -> IEEE 1/sqrt)
__device__ __forceinline__ f3 normalized3(f3 v) { float r = 1.0f / sqrtf(dot3(v, v)); return scale3(v, r); }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ float qnanf_() { return __uint_as_float(0x7fffffffu); }   // temp_utils.hpp:16

// device.hpp:53-61 : __float2half_rn / __half2float.
// The empty asm pins `f` as an already-rounded f32 value: without it the gfx950 backend folds
// fptrunc(fmul a, b) into v_fma_mixlo_f16, which rounds the EXACT product once to f16 -- a different
// result from __float2half_rn(a * b) (double rounding) on ties (measured: 8 of 307200 dists pixels).
__device__ __forceinline__ uint32_t f2h_bits(float f)
{
    asm volatile("" : "+v"(f));
    return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f);
}
__device__ __forceinline__ float h2f_bits(uint32_t h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(h & 0xffffu)); }

// ---------------------------------------------------------------- quaternion.hpp / dual_quaternion.hpp
// quaternion.hpp:186-194
__device__ __forceinline__ quat q_mul(quat a, quat b)
{
    quat r;
    r.w = ((a.w * b.w) - (a.x * b.x) - (a.y * b.y) - (a.z * b.z));
    r.x = ((a.w * b.x) + (a.x * b.w) + (a.y * b.z) - (a.z * b.y));
    r.y = ((a.w * b.y) - (a.x * b.z) + (a.y * b.w) + (a.z * b.x));
    r.z = ((a.w * b.z) + (a.x * b.y) - (a.y * b.x) + (a.z * b.w));
    return r;
}
__device__ __forceinline__ quat q_conj(quat a) { quat r; r.w = a.w; r.x = -a.x; r.y = -a.y; r.z = -a.z; return r; }
// quaternion.hpp:211-228 : (1.0/norm) * q with a double scalar, rounded per component
__device__ __forceinline__ quat q_normalize(quat a)
{
    float n = sqrtf((a.w * a.w) + (a.x * a.x) + (a.y * a.y) + (a.z * a.z));
    double inv = 1.0 / (double)n;
    quat r;
    r.w = (float)(inv * (double)a.w); r.x = (float)(inv * (double)a.x);
    r.y = (float)(inv * (double)a.y); r.z = (float)(inv * (double)a.z);
    return r;
}
__device__ __forceinline__ quat q_scale(float s, quat a) { quat r; r.w = s * a.w; r.x = s * a.x; r.y = s * a.y; r.z = s * a.z; return r; }
__device__ __forceinline__ quat q_add(quat a, quat b) { quat r; r.w = a.w + b.w; r.x = a.x + b.x; r.y = a.y + b.y; r.z = a.z + b.z; return r; }
// dual_quaternion.hpp:120-125
__device__ __forceinline__ quat dq_get_translation(quat rot, quat dual) { return q_mul(q_scale(2.f, dual), q_conj(q_normalize(rot))); }
// dual_quaternion.hpp:204-210 + quaternion.hpp:124-130, given rn = normalize(real)
__device__ __forceinline__ f3 dq_transform_rn(quat rn, quat dual, f3 p)
{
    quat t = q_mul(q_scale(2.f, dual), q_conj(rn));
    f3 qv = mk3(rn.x, rn.y, rn.z);
    f3 inner = add3(cross3(qv, p), scale3(p, rn.w));
    p = add3(p, cross3(scale3(qv, 2.f), inner));
    return add3(p, mk3(t.x, t.y, t.z));
}
__device__ __forceinline__ f3 dq_transform(quat rot, quat dual, f3 p)
{
    return dq_transform_rn(q_normalize(rot), dual, p);          // one normalize shared by getTranslation() and rotate()
}

// ---------------------------------------------------------------- short forms of the correctly rounded operations
// The warped sweep is VALU-issue bound, and a third of its instructions are the generic expansions of sqrtf, the f64
// division and the f64 scaling inside q_normalize.  On a restricted domain each has a shorter sequence with THE SAME
// result bits; callers test the domain wave-wide (df_wave_all) and fall back to the generic form otherwise.
//
// sqrtf for finite x >= 2^-96: the compiler's own expansion (v_sqrt_f32, then step to the neighbour below / above when
// the residual says so) without its input scaling for tiny x and its zero / inf / NaN pass-through.
// (2^-96 <= x < inf as ONE unsigned compare of the bits: negative floats and NaNs have bits above every positive finite float's)
__device__ __forceinline__ bool df_sqrt_short_ok(float x) { return (__float_as_uint(x) - 0x0f800000u) < 0x70000000u; }
__device__ __forceinline__ float df_sqrt_short(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    float r = (rd <= 0.f) ? sd : s;
    r = (ru > 0.f) ? su : r;
    return r;
}
// 1.0 / (double)n for a normal, finite f32 n > 0: the f64 division's Newton-Raphson core (v_rcp_f64, two refinements, one
// correction of the quotient) without v_div_scale / v_div_fixup, which only act on operands near the ends of the f64
// exponent range -- unreachable from an f32.
__device__ __forceinline__ double df_rcp_short(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double rem = __builtin_fma(-d, r, 1.0);     // quotient q = 1.0 * r
    return __builtin_fma(rem, r, r);
}
__device__ __forceinline__ float q_sumsq(quat a) { return (a.w * a.w) + (a.x * a.x) + (a.y * a.y) + (a.z * a.z); }
__device__ __forceinline__ quat q_scale_f64(double inv, quat a)
{
    quat r;
    r.w = (float)(inv * (double)a.w); r.x = (float)(inv * (double)a.x);
    r.y = (float)(inv * (double)a.y); r.z = (float)(inv * (double)a.z);
    return r;
}
// q_normalize with the short reciprocal.  Domain: everything.  For a normal, finite norm it is the same division; for norm 0, inf
// or NaN both forms make every component of the result (and, through the second normalize and the transform, of the warped
// position) non-finite -- the generic one via 1.0 / 0 = inf, (float)(inf * c) = inf or NaN, the short one via rcp(0) = inf,
// fma(-0, inf, 1) = NaN -- and a non-finite position fails the reference's vc.z > 0 test either way: the voxel is not updated.
// (The norm is never denormal: sqrtf of the smallest positive f32 is 2^-74.5.)
__device__ __forceinline__ quat q_normalize_rcp_short(quat a) { return q_scale_f64(df_rcp_short((double)sqrtf(q_sumsq(a))), a); }
// q_normalize of an (almost) unit quaternion: s = 1 + d with |d| <= 2^-20 (d = s - 1 is exact).  In f64,
//   sqrt(s) = 1 + d/2 - d^2/8 + O(d^3):  1 + d/2 is a multiple of 2^-25 and d^2/8 one of 2^-51, so the first three terms are exact
//   in f64, lie >= 2^-51 off every f32 rounding boundary when d != 0 while the dropped terms are < 2^-63  =>  the f32 rounding of
//   that f64 value IS the correctly rounded sqrtf(s);
//   1/n = 1 - e + e^2 - O(e^3) for n = 1 + e (e = n - 1 exact, a multiple of 2^-24): 1 - e + e^2 is exact in f64 and the dropped
//   terms are < 2^-63, far inside half an f64 ulp  =>  it IS the correctly rounded 1.0 / (double)n.
__device__ __forceinline__ bool q_near_unit_ok(float s) { return fabsf(s - 1.f) <= 0x1p-20f; }
__device__ __forceinline__ quat q_normalize_near_unit(quat a, float s)
{
    const double d = (double)(s - 1.f);
    const float n = (float)__builtin_fma(-0.125 * d, d, __builtin_fma(0.5, d, 1.0));
    const double e = (double)(n - 1.f);
    return q_scale_f64(__builtin_fma(e, e, 1.0 - e), a);
}
// ---- the FIRST normalisation, (float)((1.0 / (double)n) * (double)c) per component (quaternion.hpp:220-228), as an f32 division (round 5).
// Claim: for f32 c, n with a NORMAL (or zero) f32 quotient, the reference's double-rounded value equals the correctly rounded f32
// quotient RN32(c / n).  Proof.  inv = RN64(1 / n) = (1 / n)(1 + e1), p = RN64(inv * c) = (c / n)(1 + e), |e| < 2^-52 + 2^-105.  Write
// c = C 2^a, n = N 2^b with integers 2^23 <= C, N < 2^24 and let the quotient Q = c / n lie in [2^k, 2^(k+1)).  An f32 rounding boundary
// there is M 2^(k-24) with M odd, 2^24 < M < 2^25, and |Q - M 2^(k-24)| = 2^(k-24) |C 2^s - M N| / N with s = a - b - k + 24 in {24, 25}.
// C 2^s = M N is impossible (M odd => 2^s | N, but N < 2^24), so the integer |C 2^s - M N| >= 1 and Q is more than 2^(k-48) away from
// every boundary -- while |p - Q| < 2^(k+1) 2^-52 (1 + 2^-53) < 2^(k-50).  p and Q therefore round to the same f32, never on a tie.
// (A zero component gives +-0 * inv = +-0 either way; the blend sums are never -0: they start at +0 and a sum is -0 only if both
// terms are.)  RN32(c / n) itself is hipcc's division sequence minus v_div_scale / v_div_fixup (df_div_shared), which act on
// numerators below 2^-103, on denominators and quotients near the ends of the exponent range: with 2^-100 <= |c| (or c = 0) and
// 2^-48 <= n <= 2^20 -- q_div_f32_ok, tested wave-wide -- no operand, quotient (>= 2^-120) or residual (a multiple of 2^-147) comes
// near them.  dfusion_selftest_exact_forms [8] compares the two on random and edge-of-domain inputs.
__device__ __forceinline__ bool q_div_f32_ok(quat a, float s)       // s = q_sumsq(a)
{
    // every component zero or >= 2^-100 in magnitude: (bits << 1) - 1 maps +-0 to 0xffffffff and orders the rest by magnitude
    const unsigned tw = (__float_as_uint(a.w) << 1) - 1u, tx = (__float_as_uint(a.x) << 1) - 1u;
    const unsigned ty = (__float_as_uint(a.y) << 1) - 1u, tz = (__float_as_uint(a.z) << 1) - 1u;
    return (min(min(tw, tx), min(ty, tz)) >= ((0x0d800000u << 1) - 1u)) & (s >= 0x1p-96f) & (s <= 0x1p40f);
}
__device__ __forceinline__ float df_rcp_refined(float d);
__device__ __forceinline__ float df_div_shared(float n, float d, float r);
__device__ __forceinline__ quat q_div_f32(quat a, float n)
{
    const float r = df_rcp_refined(n);
    quat q;
    q.w = df_div_shared(a.w, n, r); q.x = df_div_shared(a.x, n, r); q.y = df_div_shared(a.y, n, r); q.z = df_div_shared(a.z, n, r);
    return q;
}
// ---- quaternion products on (w,x),(y,z) register pairs: 8 v_pk_mul_f32 + 6 v_pk_add_f32.  Every product and every sum of
// q_mul above, in the same association -- ((p0 +- p1) +- p2) +- p3 per component -- with the operand halves picked by op_sel and
// the signs by neg_lo / neg_hi (a sign flip of an operand: exact).  The compiler's own pairing of the scalar form needs 23
// instructions, a third of them moves.  CONJ_B: b is used as conj(b) (quaternion.hpp:124, every x/y/z operand of b negated).
typedef float df_v2f __attribute__((ext_vector_type(2)));
struct quat2 { df_v2f wx, yz; };
__device__ __forceinline__ quat2 q_pairs(quat a) { quat2 r; r.wx = df_v2f{a.w, a.x}; r.yz = df_v2f{a.y, a.z}; return r; }
__device__ __forceinline__ quat q_unpair(quat2 a) { quat r; r.w = a.wx.x; r.x = a.wx.y; r.y = a.yz.x; r.z = a.yz.y; return r; }
__device__ __forceinline__ quat2 q_mul_pk(quat2 a, quat2 b)
{
    quat2 r; df_v2f p1, p2, p3, q1, q2, q3;
    asm("v_pk_mul_f32 %[rwx], %[awx], %[bwx] op_sel_hi:[0,1]\n\t"                 // (aw bw, aw bx)
        "v_pk_mul_f32 %[p1], %[awx], %[bwx] op_sel:[1,1] op_sel_hi:[1,0]\n\t"     // (ax bx, ax bw)
        "v_pk_mul_f32 %[p2], %[ayz], %[byz] op_sel_hi:[0,1]\n\t"                  // (ay by, ay bz)
        "v_pk_mul_f32 %[p3], %[ayz], %[byz] op_sel:[1,1] op_sel_hi:[1,0]\n\t"     // (az bz, az by)
        "v_pk_mul_f32 %[ryz], %[awx], %[byz] op_sel_hi:[0,1]\n\t"                 // (aw by, aw bz)
        "v_pk_mul_f32 %[q1], %[awx], %[byz] op_sel:[1,1] op_sel_hi:[1,0]\n\t"     // (ax bz, ax by)
        "v_pk_mul_f32 %[q2], %[ayz], %[bwx] op_sel_hi:[0,1]\n\t"                  // (ay bw, ay bx)
        "v_pk_mul_f32 %[q3], %[ayz], %[bwx] op_sel:[1,1] op_sel_hi:[1,0]\n\t"     // (az bx, az bw)
        "v_pk_add_f32 %[rwx], %[rwx], %[p1] neg_lo:[0,1]\n\t"                     // w: - ax bx   x: + ax bw
        "v_pk_add_f32 %[ryz], %[ryz], %[q1] neg_lo:[0,1]\n\t"                     // y: - ax bz   z: + ax by
        "v_pk_add_f32 %[rwx], %[rwx], %[p2] neg_lo:[0,1]\n\t"                     // w: - ay by   x: + ay bz
        "v_pk_add_f32 %[ryz], %[ryz], %[q2] neg_hi:[0,1]\n\t"                     // y: + ay bw   z: - ay bx
        "v_pk_add_f32 %[rwx], %[rwx], %[p3] neg_lo:[0,1] neg_hi:[0,1]\n\t"        // w: - az bz   x: - az by
        "v_pk_add_f32 %[ryz], %[ryz], %[q3]"                                       // y: + az bx   z: + az bw
        : [rwx] "=&v"(r.wx), [ryz] "=&v"(r.yz), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3)
        : [awx] "v"(a.wx), [ayz] "v"(a.yz), [bwx] "v"(b.wx), [byz] "v"(b.yz));
    return r;
}
__device__ __forceinline__ quat2 q_mul_conj_pk(quat2 a, quat2 b)
{
    quat2 r; df_v2f p1, p2, p3, q1, q2, q3;
    asm("v_pk_mul_f32 %[rwx], %[awx], %[bwx] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"                          // (aw bw, aw (-bx))
        "v_pk_mul_f32 %[p1], %[awx], %[bwx] op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"              // (ax (-bx), ax bw)
        "v_pk_mul_f32 %[p2], %[ayz], %[byz] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"              // (ay (-by), ay (-bz))
        "v_pk_mul_f32 %[p3], %[ayz], %[byz] op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" // (az (-bz), az (-by))
        "v_pk_mul_f32 %[ryz], %[awx], %[byz] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"             // (aw (-by), aw (-bz))
        "v_pk_mul_f32 %[q1], %[awx], %[byz] op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" // (ax (-bz), ax (-by))
        "v_pk_mul_f32 %[q2], %[ayz], %[bwx] op_sel_hi:[0,1] neg_hi:[0,1]\n\t"                           // (ay bw, ay (-bx))
        "v_pk_mul_f32 %[q3], %[ayz], %[bwx] op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"              // (az (-bx), az bw)
        "v_pk_add_f32 %[rwx], %[rwx], %[p1] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[ryz], %[ryz], %[q1] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[rwx], %[rwx], %[p2] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %[ryz], %[ryz], %[q2] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[rwx], %[rwx], %[p3] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[ryz], %[ryz], %[q3]"
        : [rwx] "=&v"(r.wx), [ryz] "=&v"(r.yz), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3)
        : [awx] "v"(a.wx), [ayz] "v"(a.yz), [bwx] "v"(b.wx), [byz] "v"(b.yz));
    return r;
}
// dq_transform_rn with the two quaternion products on register pairs (dual = 0.5 * tsum * rot is formed by the caller the same way)
__device__ __forceinline__ f3 dq_transform_rn_pk(quat rn, quat2 dual, f3 p)
{
    quat2 d2; d2.wx = dual.wx * 2.f; d2.yz = dual.yz * 2.f;                              // q_scale(2, dual)
    const quat2 t = q_mul_conj_pk(d2, q_pairs(rn));
    f3 qv = mk3(rn.x, rn.y, rn.z);
    f3 inner = add3(cross3(qv, p), scale3(p, rn.w));
    p = add3(p, cross3(scale3(qv, 2.f), inner));
    return add3(p, mk3(t.wx.y, t.yz.x, t.yz.y));
}
__device__ __forceinline__ bool df_wave_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }
// warp_field.cpp:238-241 (double exp overload, see oracle header)
__device__ __forceinline__ float dqb_weight(float d2, float sigma) { return (float)exp((double)(-d2 / (2 * sigma * sigma))); }
// knn_point_cloud.hpp:25-31
__device__ __forceinline__ float knn_dist2(f3 q, float px, float py, float pz)
{
    const float d0 = q.x - px, d1 = q.y - py, d2 = q.z - pz;
    return d0 * d0 + d1 * d1 + d2 * d2;
}

// ---------------------------------------------------------------- projective TSDF update
// tsdf_volume.cu:77-104 from the projection on.  Returns the new packed voxel in `vox` and true
// if the update branch (:91) was taken.  `vox_in` is only meaningful when it is.
struct DfIntegrateParams {
    const uint16_t* dists; size_t pitch; int cols, rows;
    float fx, fy, cx, cy;
    float trunc, trunc_inv; int max_weight;
};

// Decide whether voxel at camera-frame vc updates; returns tsdf sample in *tsdf_out.
__device__ __forceinline__ bool tsdf_sample(const DfIntegrateParams& P, f3 vc, float* tsdf_out)
{
    if (!(vc.z > 0.f)) return false;                      // :86 (second half); order-independent skip
    float u = fmaf(P.fx, vc.x / vc.z, P.cx);              // device.hpp:35
    float v = fmaf(P.fy, vc.y / vc.z, P.cy);              // device.hpp:36
    if (!(u >= 0.f && v >= 0.f && u < (float)P.cols && v < (float)P.rows)) return false;   // :82 (+NaN => skip)
    const uint16_t* row = (const uint16_t*)((const char*)P.dists + (size_t)(int)v * P.pitch);
    float Dp = h2f_bits(row[(int)u]);                     // :85
    if (Dp == 0.f) return false;                          // :86
    float sdf = Dp - sqrtf(dot3(vc, vc));                 // :89
    if (!(sdf >= -P.trunc)) return false;                 // :91
    *tsdf_out = fminf(1.f, sdf * P.trunc_inv);            // :93
    return true;
}
// Branch-free form of tsdf_sample for the rigid sweep: every test is evaluated, the dists fetch uses a clamped
// (always valid) address, and the verdict is the AND of the reference's conditions -- the same result for every
// voxel (for a voxel that passes :82, (int)u is unchanged by the clamp), but four voxels per lane run as
// independent straight-line chains instead of four nested divergent branches.
__device__ __forceinline__ bool tsdf_sample_nb(const DfIntegrateParams& P, f3 vc, float* tsdf_out)
{
    const float u = fmaf(P.fx, vc.x / vc.z, P.cx);        // device.hpp:35
    const float v = fmaf(P.fy, vc.y / vc.z, P.cy);        // device.hpp:36
    bool ok = (vc.z > 0.f) & (u >= 0.f) & (v >= 0.f) & (u < (float)P.cols) & (v < (float)P.rows);   // :82, :86 (+NaN => skip)
    const int ui = (int)fminf(fmaxf(u, 0.f), (float)(P.cols - 1));
    const int vi = (int)fminf(fmaxf(v, 0.f), (float)(P.rows - 1));
    const uint16_t* row = (const uint16_t*)((const char*)P.dists + (size_t)vi * P.pitch);
    const float Dp = h2f_bits(row[ui]);                   // :85
    const float sdf = Dp - sqrtf(dot3(vc, vc));           // :89
    ok = ok & (Dp != 0.f) & (sdf >= -P.trunc);            // :86, :91
    *tsdf_out = fminf(1.f, sdf * P.trunc_inv);            // :93
    return ok;
}
// The same sample on a restricted domain, for sweeps that have established it for a whole run of voxels (the rigid sweep does, per
// sub-chunk and per wave): 2^-20 <= |vc.x|, |vc.y| <= 2^30 and 0.05 <= vc.z <= 2^30, dists image below 2 GiB (32-bit offsets).
// There the correctly rounded operations have shorter sequences WITH THE SAME BITS:
//   * x / z and y / z share the refined reciprocal.  hipcc's f32 division is v_div_scale (x2), v_rcp, one Newton step on the
//     reciprocal, quotient, two residual corrections (the last inside v_div_fmas), v_div_fixup.  With every exponent this far from
//     the ends of the range v_div_scale returns its operand unchanged (it acts on denormal / huge denominators, on quotients near
//     the denormal range and on numerators below 2^-103), v_div_fmas is a plain fma and v_div_fixup passes the finite, normal
//     quotient through -- what remains is the sequence below, 13 instructions for both quotients instead of 22;
//   * sqrtf as df_sqrt_short (argument >= vc.z^2 >= 2^-9);
//   * the pixel clamp by v_med3_f32 (no NaN can reach it), the dists address in 32 bits.
// dfusion_selftest_exact_forms (counts[6]) compares it, and the two-stage form below, with tsdf_sample_nb on random in-domain
// positions and on the domain's edges.  (A numerator of -0 gives +0 here where the IEEE division gives -0; both make the same
// u = fma(fx, q, cx) because cx > 0, which the callers of this form require.)
__device__ __forceinline__ float df_div_shared(float n, float d, float r)       // n / d given r = df_rcp_refined(d)
{
    float q = n * r;
    q = fmaf(fmaf(-d, q, n), r, q);
    return fmaf(fmaf(-d, q, n), r, q);
}
__device__ __forceinline__ float df_rcp_refined(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ bool tsdf_sample_domain_ok(f3 a, f3 b)               // both ends of a run along which vc moves monotonically
{
    const float lo = 0x1p-20f, hi = 0x1p30f;
    const bool x_ok = (a.x * b.x > 0.f) & (fminf(fabsf(a.x), fabsf(b.x)) >= lo) & (fmaxf(fabsf(a.x), fabsf(b.x)) <= hi);
    const bool y_ok = (a.y * b.y > 0.f) & (fminf(fabsf(a.y), fabsf(b.y)) >= lo) & (fmaxf(fabsf(a.y), fabsf(b.y)) <= hi);
    const bool z_ok = (fminf(a.z, b.z) >= 0.05f) & (fmaxf(a.z, b.z) <= hi);
    return x_ok & y_ok & z_ok;
}
__device__ __forceinline__ bool tsdf_sample_fast(const DfIntegrateParams& P, f3 vc, float* tsdf_out)
{
    const float r = df_rcp_refined(vc.z);
    const float u = fmaf(P.fx, df_div_shared(vc.x, vc.z, r), P.cx);        // device.hpp:35
    const float v = fmaf(P.fy, df_div_shared(vc.y, vc.z, r), P.cy);        // device.hpp:36
    // :82 (:86's vc.z > 0 holds on the domain).  0 <= u < cols as ONE unsigned compare of the float bits: non-negative floats order
    // like their bits, negative ones and NaNs have bits above every positive finite float's.  (-0.f would differ -- it is >= 0 --
    // but u = fma(fx, q, cx) with cx > 0, which the callers of this form require, is never -0.)
    bool ok = (__float_as_uint(u) < __float_as_uint((float)P.cols)) & (__float_as_uint(v) < __float_as_uint((float)P.rows));
    const uint32_t ui = (uint32_t)(int)__builtin_amdgcn_fmed3f(u, 0.f, (float)(P.cols - 1));
    const uint32_t vi = (uint32_t)(int)__builtin_amdgcn_fmed3f(v, 0.f, (float)(P.rows - 1));
    const uint32_t off = vi * (uint32_t)P.pitch + 2u * ui;
    const float Dp = h2f_bits(*(const uint16_t*)((const char*)P.dists + off));       // :85
    const float sdf = Dp - df_sqrt_short(dot3(vc, vc));    // :89
    ok = ok & (Dp != 0.f) & (sdf >= -P.trunc);             // :86, :91
    *tsdf_out = fminf(1.f, sdf * P.trunc_inv);             // :93
    return ok;
}
// ---- the sample in two stages, for sweeps that decide wave-wide whether the exact square root is needed at all.
// Most voxels that update lie in observed free space, far in front of the surface: their tsdf saturates at exactly 1.f
// (:93), whatever the last bits of |vc| are.  tsdf_sample_pre does everything of tsdf_sample_fast up to the hardware
// approximation s ~ sqrt(|vc|^2) (v_sqrt_f32, 1 ulp); with sdf_a = Dp - s, the EXACT sdf = Dp - sqrtf(|vc|^2) differs from
// sdf_a by at most 1.5 ulp(s) + two roundings of the difference, < 2^-17 + 2^-23 |sdf_a| for |vc| < 64 m.  With
//     T = trunc * (1 + 2^-10) + 2^-16            (df_sat_threshold; 2^-10 <= trunc <= 2^10)
//   sdf_a >=  T  =>  sdf >= trunc (1 + 2^-20)  =>  fl(sdf * fl(1 / trunc)) >= 1  =>  tsdf = fminf(1, .) = 1.f exactly;
//   sdf_a <= -T  =>  sdf < -trunc: the update branch (:91) is not taken;
//   |sdf_a| < T  =>  undecided: tsdf_sample_finish corrects s to the correctly rounded root (df_sqrt_short's tail) and
//                    evaluates :89-93 as written.  NaN / inf dists values fall out right: NaN compares false everywhere
//                    (no update, as in `sdf >= -trunc`), +inf saturates, -inf is below -T.
// A wave takes the finish only when one of its voxels is undecided (a few planes either side of the surface).
struct DfSamplePre { float Dp, d2, s; bool ok; };
__device__ __forceinline__ DfSamplePre tsdf_sample_pre(const DfIntegrateParams& P, f3 vc)
{
    DfSamplePre r;
    const float rc = df_rcp_refined(vc.z);
    const float u = fmaf(P.fx, df_div_shared(vc.x, vc.z, rc), P.cx);       // device.hpp:35
    const float v = fmaf(P.fy, df_div_shared(vc.y, vc.z, rc), P.cy);       // device.hpp:36
    r.ok = (__float_as_uint(u) < __float_as_uint((float)P.cols)) & (__float_as_uint(v) < __float_as_uint((float)P.rows));   // :82 (see tsdf_sample_fast)
    const uint32_t ui = (uint32_t)(int)__builtin_amdgcn_fmed3f(u, 0.f, (float)(P.cols - 1));
    const uint32_t vi = (uint32_t)(int)__builtin_amdgcn_fmed3f(v, 0.f, (float)(P.rows - 1));
    const uint32_t off = vi * (uint32_t)P.pitch + 2u * ui;
    r.Dp = h2f_bits(*(const uint16_t*)((const char*)P.dists + off));      // :85
    r.ok = r.ok & (r.Dp != 0.f);                                           // :86
    r.d2 = dot3(vc, vc);
    r.s = __builtin_amdgcn_sqrtf(r.d2);
    return r;
}
__device__ __forceinline__ float df_sat_threshold(float trunc) { return trunc * (1.f + 0x1p-10f) + 0x1p-16f; }
__host__ __device__ __forceinline__ bool df_sat_trunc_ok(float trunc) { return (trunc >= 0x1p-10f) & (trunc <= 0x1p10f); }
// both ends of a run: every coordinate within 32 m (|vc| < 64: the error budget above)
__device__ __forceinline__ bool tsdf_sat_domain_ok(f3 a, f3 b)
{
    return fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(b.x)), fmaxf(fabsf(a.y), fabsf(b.y))), fmaxf(fabsf(a.z), fabsf(b.z))) <= 32.f;
}
__device__ __forceinline__ bool tsdf_sample_finish(const DfIntegrateParams& P, const DfSamplePre& r, float* tsdf_out)
{
    const float s = r.s, x = r.d2;                                          // df_sqrt_short from its second line on
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = fmaf(-sd, s, x), ru = fmaf(-su, s, x);
    float n = (rd <= 0.f) ? sd : s;
    n = (ru > 0.f) ? su : n;
    const float sdf = r.Dp - n;                                             // :89
    *tsdf_out = fminf(1.f, sdf * P.trunc_inv);                              // :93
    return r.ok & (sdf >= -P.trunc);                                        // :91
}
// :97-103 for a saturated sample (tsdf = 1.f) on a voxel whose stored value is 1.0 or that is still cleared: fma(1, w, 1) = w + 1
// exactly (w <= 65535), (w + 1) / (w + 1) = 1; fma(0, 0, 1) / 1 = 1 -- the new value is 1.0 (0x3c00) without any arithmetic.
__device__ __forceinline__ bool tsdf_fuse_one_ok(uint32_t vox) { return (vox == 0u) | ((vox & 0xffffu) == 0x3c00u); }
__device__ __forceinline__ uint32_t tsdf_fuse_one(uint32_t vox, int max_weight)
{
    const int weight_new = min((int)(vox >> 16) + 1, max_weight);
    return 0x3c00u | (((uint32_t)weight_new & 0xffffu) << 16);
}
// :97-103
__device__ __forceinline__ uint32_t tsdf_fuse(uint32_t vox, float tsdf, int max_weight)
{
    int weight_prev = (int)(vox >> 16);
    float tsdf_prev = h2f_bits(vox);
    float tsdf_new = fmaf(tsdf_prev, (float)weight_prev, tsdf) / (float)(weight_prev + 1);
    int weight_new = min(weight_prev + 1, max_weight);
    return f2h_bits(tsdf_new) | (((uint32_t)weight_new & 0xffffu) << 16);
}

// The same with the division in its short form (df_div_shared: hipcc's own sequence minus v_div_scale / v_div_fixup).  The
// denominator is an integer in [1, 65536]; the numerator fma(prev, w, tsdf) is 0 or at least ~2^-50 in magnitude (prev is a half,
// w an integer, |tsdf| a difference of a half-derived and an f32 length times 1 / trunc): no operand, quotient or residual comes near
// the denormal range, where v_div_scale would have acted.  Valid for FINITE prev (tsdf_fuse_short_ok); the generic form otherwise.
// dfusion_selftest_exact_forms compares the two on every half x every weight x a spread of tsdf values.
__device__ __forceinline__ bool tsdf_fuse_short_ok(uint32_t vox) { return (vox & 0x7c00u) != 0x7c00u; }
__device__ __forceinline__ uint32_t tsdf_fuse_short(uint32_t vox, float tsdf, int max_weight)
{
    const int weight_prev = (int)(vox >> 16);
    const float tsdf_prev = h2f_bits(vox);
    const float d = (float)(weight_prev + 1);
    const float tsdf_new = df_div_shared(fmaf(tsdf_prev, (float)weight_prev, tsdf), d, df_rcp_refined(d));
    const int weight_new = min(weight_prev + 1, max_weight);
    return f2h_bits(tsdf_new) | (((uint32_t)weight_new & 0xffffu) << 16);
}

// ---------------------------------------------------------------- wave64 helpers
__device__ __forceinline__ float wave_min_f32(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned long long lane_mask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }
