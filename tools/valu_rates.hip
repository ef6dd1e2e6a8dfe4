// Issue cost of the VALU instruction classes the sweep kernels are made of (cycles per wave64 instruction per SIMD), measured with
// s_memtime inside one wave while W waves per SIMD run the same loop.  hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o build/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
#define K_BODY(NAME, ASM)                                                                                              \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* out, int iters, float seed)                        \
    {                                                                                                                  \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; \
        double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3;                                                  \
        typedef float v2 __attribute__((ext_vector_type(2)));                                                          \
        v2 p0 = {seed, seed}, p1 = {seed + 1, seed}, p2 = {seed + 2, seed}, p3 = {seed + 3, seed};                      \
        unsigned long long t0 = __builtin_readcyclecounter();                                                          \
        for (int i = 0; i < iters; ++i) {                                                                              \
            _Pragma("unroll") for (int r = 0; r < REP / 4; ++r) { ASM }                                               \
        }                                                                                                              \
        unsigned long long t1 = __builtin_readcyclecounter();                                                          \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + p0.x + p1.x + p2.x + p3.x + p0.y + p1.y + p2.y + p3.y == 12345.f) out[1] = 1; \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                     \
    }
K_BODY(k_fma, asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_mul, asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_pkfma, asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
K_BODY(k_pkmul, asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
K_BODY(k_pkadd, asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
K_BODY(k_pkmul_opsel, asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[0,1]\n v_pk_mul_f32 %1, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]\n v_pk_mul_f32 %2, %2, %3 op_sel_hi:[0,1]\n v_pk_mul_f32 %3, %3, %0 op_sel:[1,0] op_sel_hi:[1,1]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
K_BODY(k_fma64, asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
K_BODY(k_mul64, asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
K_BODY(k_cvt_f64_f32, asm volatile("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %6\n v_cvt_f64_f32 %3, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
K_BODY(k_cvt_f32_f64, asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));)
K_BODY(k_sqrt, asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_rcp, asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_rcp64, asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
K_BODY(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
K_BODY(k_cndmask_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_cndmask_b32_e64 %1, %1, %2, s[10:11]\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n v_cndmask_b32_e64 %3, %3, %0, s[10:11]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s10", "s11");)
K_BODY(k_cmp_cndmask, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
K_BODY(k_cndmask_indep, asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %6, vcc\n v_cndmask_b32 %2, %6, %7, vcc\n v_cndmask_b32 %3, %7, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");)
K_BODY(k_bfi, asm volatile("v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %1, %1, %2, %3\n v_bfi_b32 %2, %2, %3, %0\n v_bfi_b32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_add_f32, asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_add_u32, asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_readfirstlane, asm volatile("v_readfirstlane_b32 s10, %0\n v_readfirstlane_b32 s11, %1\n v_readfirstlane_b32 s10, %2\n v_readfirstlane_b32 s11, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "s10", "s11");)
K_BODY(k_mov, asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_sdwa, asm volatile("v_lshlrev_b32_sdwa %0, 5, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_lshlrev_b32_sdwa %1, 5, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_lshlrev_b32_sdwa %2, 5, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_lshlrev_b32_sdwa %3, 5, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_divscale, asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0\n v_div_scale_f32 %1, vcc, %1, %2, %1\n v_div_scale_f32 %2, vcc, %2, %3, %2\n v_div_scale_f32 %3, vcc, %3, %0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
K_BODY(k_divfmas, asm volatile("v_div_fmas_f32 %0, %0, %1, %2\n v_div_fmas_f32 %1, %1, %2, %3\n v_div_fmas_f32 %2, %2, %3, %0\n v_div_fmas_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
K_BODY(k_divfixup, asm volatile("v_div_fixup_f32 %0, %0, %1, %2\n v_div_fixup_f32 %1, %1, %2, %3\n v_div_fixup_f32 %2, %2, %3, %0\n v_div_fixup_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_cmp, asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
K_BODY(k_med3, asm volatile("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %0\n v_med3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_cvt_i32, asm volatile("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
K_BODY(k_fma_mix, asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %2, %3 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %3, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %0, %1 op_sel_hi:[1,0,0]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)

typedef void (*kern_t)(unsigned long long*, int, float);
int main()
{
    unsigned long long* out; hipMalloc(&out, 16); hipMemset(out, 0, 16);
    int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    struct { const char* n; kern_t k; } ks[] = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_pk_fma_f32", k_pkfma}, {"v_pk_mul_f32", k_pkmul},
        {"v_pk_add_f32", k_pkadd}, {"v_pk_mul_f32 op_sel", k_pkmul_opsel}, {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_cvt_f64_f32", k_cvt_f64_f32},
        {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_sqrt_f32", k_sqrt}, {"v_rcp_f32", k_rcp}, {"v_rcp_f64", k_rcp64}, {"v_cndmask_b32", k_cndmask},
        {"v_cndmask_b32_e64 sgpr", k_cndmask_e64}, {"v_cmp + v_cndmask (x2)", k_cmp_cndmask}, {"v_cndmask indep dst", k_cndmask_indep}, {"v_bfi_b32", k_bfi}, {"v_add_f32", k_add_f32}, {"v_add_u32", k_add_u32}, {"v_readfirstlane_b32", k_readfirstlane},
        {"v_mov_b32", k_mov}, {"v_lshlrev_b32_sdwa", k_sdwa}, {"v_div_scale_f32", k_divscale}, {"v_div_fmas_f32", k_divfmas},
        {"v_div_fixup_f32", k_divfixup}, {"v_cmp_lt_f32", k_cmp}, {"v_med3_f32", k_med3}, {"v_cvt_i32_f32", k_cvt_i32}, {"v_fma_mix_f32", k_fma_mix}};
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-22s %9s %9s %9s   ns per wave64 instruction per SIMD (wall clock over the whole launch) at 1, 4, 8 waves per SIMD; [s_memtime ticks in one wave]\n", "instruction", "1 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    for (auto& e : ks) {
        double r[3], tk[3];
        for (int m = 0; m < 3; ++m) {
            const int wg_per_cu = m == 0 ? 1 : m == 1 ? 4 : 8;            // 256 threads = 4 waves = 1 per SIMD
            e.k<<<ncu * wg_per_cu, 256>>>(out, iters, 1.5f); hipDeviceSynchronize();
            hipEventRecord(e0); e.k<<<ncu * wg_per_cu, 256>>>(out, iters, 1.5f); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
            r[m] = (double)ms * 1e6 / ((double)iters * REP * wg_per_cu);
            tk[m] = (double)t / ((double)iters * REP);
        }
        printf("%-22s %9.3f %9.3f %9.3f   [%.2f %.2f %.2f]\n", e.n, r[0], r[1], r[2], tk[0], tk[1], tk[2]);
    }
    // clock: s_memtime ticks per microsecond
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); k_fma<<<ncu, 256>>>(out, 200000, 1.5f); hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long t; hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
    printf("s_memtime: %.1f ticks/us over a %.2f ms v_fma_f32 kernel (%d CUs)\n", (double)t / (ms * 1000.0), ms, ncu);
    return 0;
}
