// Max-pyramid of a frame's dists image: what the integrate kernels' behind-the-surface tests query.  Built by
// df_build_dists_pyramid (dfusion_volume.hip) on the caller's stream; queried on the device with df_pyramid_max.
#pragma once
#include "dfusion_internal.h"

// Per frame, for the sweeps' behind-the-surface tests: level l holds, per 2^l x 2^l pixel block, the maximum of the
// dists HALF BITS taken as unsigned integers.  For non-negative finite halves that is the maximum ray length; negative values,
// infinities and NaNs order ABOVE every finite length (sign / exponent bits), so a block containing one can never be culled --
// the conservative direction.  Level 0 is the image itself.
#define DF_PYR_MAX_LEVELS 14
struct DfDistsPyramid {
    const uint16_t* dists; size_t pitch; int cols, rows;
    const uint16_t* mem;                     // levels 1..top, dense, level l at off[l] with width w[l]
    int off[DF_PYR_MAX_LEVELS], w[DF_PYR_MAX_LEVELS], h[DF_PYR_MAX_LEVELS];
    int top;                                 // coarsest level (1 x 1); 0 = no pyramid (test disabled)
    // levels above 5 NOT built (one launch instead of two): the image-wide maximum (what the 1 x 1 level holds) is then *max_bits,
    // accumulated by the tile kernel's workgroups with one atomicMax each; df_pyramid_max_fine callers cap the level at 5 (`capped`)
    const uint32_t* max_bits; int capped;
};
__device__ __forceinline__ uint32_t df_pyramid_image_max(const DfDistsPyramid& P) { return P.capped ? *P.max_bits : (uint32_t)P.mem[P.off[P.top]]; }


// Builds levels 1..top into `mem` (df_pyramid_elems(cols, rows) uint16 entries) on `st`; out->top == 0 when the image is too small
// (< 32 px) or too large for a pyramid: the callers then run without the test.
// `levels_to_5`: build levels 1..5 only (one launch; the descriptor still names `top`, whose levels above 5 are then NOT valid);
// `zero16`: 16 32-bit words the first workgroup also zeroes (a caller's counters, saving it a memset), or null.
// `max_word` (with levels_to_5): a device word that is 0 on entry and receives the image-wide maximum of the half bits (out->max_bits;
// the descriptor is then `capped`); the caller zeroes it again once its readers are done (df_sweep_plan_kernel does).
int df_build_dists_pyramid(const uint16_t* dists, size_t pitch, int cols, int rows, uint16_t* mem, size_t mem_elems,
                           DfDistsPyramid* out, hipStream_t st, bool levels_to_5 = false, unsigned int* zero16 = nullptr, unsigned int* max_word = nullptr);
size_t df_pyramid_elems(int cols, int rows);

// max of the dists half bits over the pixel rectangle [u0, u1] x [v0, v1] (inclusive, inside the image) or a superset of it: the texels
// of the level `shift` below the coarsest one at which the rectangle spans at most 2 x 2 texels -- at most (2^(shift+1) + 1)^2 of
// them (the 2 x 2 cover itself takes in up to three times the rectangle's extent on each axis).  `max_level` caps the level read (more texels
// instead): a caller that only ever passes max_level <= 5 does not need the levels above 5 built.
__device__ __forceinline__ uint32_t df_pyramid_max_fine(const DfDistsPyramid& P, int u0, int v0, int u1, int v1, int shift, int max_level = DF_PYR_MAX_LEVELS)
{
    const int ext = max(u1 - u0, v1 - v0);
    int L = ext == 0 ? 0 : 32 - __clz(ext);
    if (P.capped) {
        max_level = min(max_level, 5);
        // a rectangle that would span more than 5 x 5 texels of level 5: the image-wide maximum instead (a superset of the rectangle)
        if ((u1 >> 5) - (u0 >> 5) >= 5 || (v1 >> 5) - (v0 >> 5) >= 5) return *P.max_bits;
    }
    L = min(min(max(L - shift, 0), P.top), max_level);
    const int a0 = u0 >> L, a1 = u1 >> L, b0 = v0 >> L, b1 = v1 >> L;
    uint32_t m = 0;
    if (L == 0) {
        for (int b = b0; b <= b1; ++b) {
            const uint16_t* r = (const uint16_t*)((const char*)P.dists + (size_t)b * P.pitch);
            for (int a = a0; a <= a1; ++a) m = max(m, (uint32_t)r[a]);
        }
        return m;
    }
    const uint16_t* lv = P.mem + P.off[L];
    const int w = P.w[L];
    if (a1 - a0 < 5 && b1 - b0 < 5) {
        // the usual case (shift <= 1 always, shift = 2 unless max_level cut in): 25 clamped, unconditional loads, all in flight together
        // (the loops below wait for each texel in turn -- 25 dependent round trips in a kernel that does little else)
        uint32_t t[25];
#pragma unroll
        for (int i = 0; i < 25; ++i) t[i] = (uint32_t)lv[min(b0 + i / 5, b1) * w + min(a0 + i % 5, a1)];
#pragma unroll
        for (int i = 0; i < 25; ++i) m = max(m, t[i]);
        return m;
    }
    for (int b = b0; b <= b1; ++b)
        for (int a = a0; a <= a1; ++a) m = max(m, (uint32_t)lv[b * w + a]);
    return m;
}
