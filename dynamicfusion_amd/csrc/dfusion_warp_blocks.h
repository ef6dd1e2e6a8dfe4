// Block blend models: a second, much tighter conservative verdict for the warped sweep's launch plan (included by dfusion_warp.hip
// after DfWarpedArgs / df_tab_index / the pyramid).
//
// df_tile_culled() bounds a block's warped positions by a ball around its UNWARPED centre whose radius is the largest motion any
// node could cause (max angle x lever arm + k max|t|).  On the headline scene that ball keeps 57 % of the 8 x 8 x 8 blocks alive
// where 27 % hold a voxel that updates (swept / updated = 2.4): the nodes turn by up to 5 degrees about the world origin, metres away,
// but a voxel's blend AVERAGES its k nodes, and the bound cannot know that.  The per-voxel weight tables can: they are frame-invariant,
// so what the blend of a block can be, as a function of the frame's node transforms, is frame-invariant too.
//
// The reference's blend (warp_field.cpp:203-217, dual_quaternion.hpp:59-63, 204-210) of a voxel q with neighbours i, weights w_i:
//     rotation    rn = normalize(sum w_i r_i)            position = R(rn) q + sum w_i t_i      (t_i = the node's translation)
// With lambda_i = w_i / sum w (normalised weights; the rotation does not see the scale) and r_i = (s_i, v_i), s_i > 0:
//     the rotation's Gibbs vector  u = N / D,   N = sum lambda_i v_i,  D = sum lambda_i s_i   -- LINEAR in lambda over LINEAR in lambda,
//     the translation              T = sum w_i t_i                                             -- linear in w.
// MODEL of a block (built once per weight table, for the blocks a sweep finds alive, by df_block_model_kernel): the union of its
// 512 voxels' neighbour sets (<= 16 nodes, mean 11; larger unions -> no model, the ball test alone decides) and for every node of it
// an interval [mid - hw, mid + hw] that holds lambda_i(q) of every voxel of the block (0 where the node is not a neighbour), and one
// for w_i(q).  10 bytes per entry.
// VERDICT per frame (df_block_verdict_kernel, one lane per block, after the ball test): with the frame's node transforms,
//     N_c in  sum mid_i v_ic + (1 - sum mid_i) v*_c  +-  sum hw_i |v_ic - v*_c|          (sum lambda_i = 1; v* = entry 0's value)
//     D   in  [min s_i, max s_i]                                                          (a convex combination)
//     T_c in  sum wmid_i t_ic  +-  sum whw_i |t_ic|
// then interval arithmetic through  R(u) q = q + 2 / (1 + |u|^2) (u x q + u x (u x q))  over the block's box of q, through world2cam,
// and the tests of df_tile_culled on the resulting camera-frame BOX: behind the camera, projecting outside the image, no valid depth
// where it projects, or farther than (largest dists value where it projects) + trunc.  Measured on the headline scene: 35 % of the
// blocks stay alive (the exact bounding boxes of the warped voxels would keep 29 %).
// Everything is evaluated in f32 without directed rounding; the box is inflated by 1 mm + 1e-5 of the coordinates' magnitude, three
// orders of magnitude above the rounding of either side.  Any comparison that involves a NaN / inf keeps the block alive.
// (The sweep forms the translation as (0.5 tsum rn) * 2 conj(rn): its vector part is tsum's whatever tsum's scalar part, up to
// rounding relative to |tsum| -- inside the inflation for any translation the volume could hold.)
#pragma once

#define DF_BM_NU 16                 // entries per block model
#define DF_BM_NONE 0xffu            // count byte: no model for this block

struct DfIv { float lo, hi; };
__device__ __forceinline__ DfIv iv(float lo, float hi) { DfIv r; r.lo = lo; r.hi = hi; return r; }
__device__ __forceinline__ DfIv iv_add(DfIv a, DfIv b) { return iv(a.lo + b.lo, a.hi + b.hi); }
__device__ __forceinline__ DfIv iv_sub(DfIv a, DfIv b) { return iv(a.lo - b.hi, a.hi - b.lo); }
__device__ __forceinline__ DfIv iv_mul(DfIv a, DfIv b)
{
    const float p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    return iv(fminf(fminf(p0, p1), fminf(p2, p3)), fmaxf(fmaxf(p0, p1), fmaxf(p2, p3)));
}
__device__ __forceinline__ DfIv iv_sq(DfIv a)                               // [min x^2, max x^2]
{
    const float l2 = a.lo * a.lo, h2 = a.hi * a.hi;
    return iv((a.lo <= 0.f && a.hi >= 0.f) ? 0.f : fminf(l2, h2), fmaxf(l2, h2));
}
struct DfIv3 { DfIv x, y, z; };
__device__ __forceinline__ DfIv3 iv_cross(const DfIv3& a, const DfIv3& b)
{
    DfIv3 r;
    r.x = iv_sub(iv_mul(a.y, b.z), iv_mul(a.z, b.y));
    r.y = iv_sub(iv_mul(a.z, b.x), iv_mul(a.x, b.z));
    r.z = iv_sub(iv_mul(a.x, b.y), iv_mul(a.y, b.x));
    return r;
}

__device__ __forceinline__ uint32_t df_f2h_up(float x)                     // smallest half >= x (x >= 0, finite, < 65504)
{
    _Float16 h = (_Float16)x;
    unsigned short b = __builtin_bit_cast(unsigned short, h);
    if ((float)h < x) b = (unsigned short)(b + 1u);
    return b;
}
__device__ __forceinline__ uint32_t df_bm_pack(float lo, float hi)          // {mid, hw} as halves, [mid - hw, mid + hw] contains [lo, hi]
{
    const _Float16 m = (_Float16)(0.5f * (lo + hi));
    const float mf = (float)m;
    const float hw = fmaxf(hi - mf, mf - lo) * 1.00390625f + 1e-7f;
    return (uint32_t)__builtin_bit_cast(unsigned short, m) | (df_f2h_up(hw) << 16);
}

// ---- the model build: one wave per 8 x 8 x 8 block (lane = column (x, y), 8 voxels each), four blocks per workgroup round; the
// blocks come from a work list (`list`, `*count` entries: the blocks the verdict pass found alive and built but without a model).
// The union of the neighbour sets comes out in ascending node order: each round takes the smallest id above the last one over all
// 512 x K table entries (a wave minimum), then the range of that node's lambda and w over the voxels.
template <int K>
__global__ __launch_bounds__(256, K == 8 ? 3 : 4) void df_block_model_kernel(const DfWarpedArgs a, int nbx, int nby, int nbz, const uint32_t* __restrict__ list,
                                                             const uint32_t* __restrict__ count, uint16_t* __restrict__ bm_idx,
                                                             uint32_t* __restrict__ bm_lam, uint32_t* __restrict__ bm_w,
                                                             uint8_t* __restrict__ bm_cnt, uint8_t* __restrict__ blk_state)
{
    __shared__ uint32_t s_idx[4][DF_BM_NU], s_lam[4][DF_BM_NU], s_w[4][DF_BM_NU];
    __shared__ float s_hw[4][DF_BM_NU];
    const int wave = threadIdx.x >> 6, ln = threadIdx.x & 63;
    const size_t nblk = (size_t)nbx * nby * nbz;
    const uint32_t n_work = *count;
    for (uint32_t it = blockIdx.x * 4 + wave; it < n_work; it += gridDim.x * 4) {    // (waves are independent: no workgroup barrier below)
    const size_t blk = list[it];
    const int bx = (int)(blk % (size_t)nbx), by = (int)((blk / (size_t)nbx) % (size_t)nby), bz = (int)(blk / ((size_t)nbx * nby));
    const int x = bx * 8 + (ln & 7), y = by * 8 + (ln >> 3), z0 = a.tab_z0 + bz * 8;
    const bool col_in = x < a.X && y < a.Y;
    int ids[8][K];
    float wv[8][K], inv[8];
    bool bad = false;
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bool in = col_in && z0 + j < a.Z;
        // (padded table entries exist but hold nothing; a lane outside the volume reads voxel (0, 0, z0) instead and ignores it)
        const size_t tv = df_tab_index(a, in ? x : 0, in ? y : 0, in ? z0 + j : z0);
        knn_tab_load<K>(a.knn_tab, tv, ids[j]);
        w_tab_load<K>(a.w_tab, a.tab_nvox, tv, wv[j]);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) s += wv[j][i];
        inv[j] = 1.f / s;
        if (in) { valid |= 1u << j; if (!(s > 1e-30f && s < 3.0e38f)) bad = true; }
    }
    bad = __builtin_amdgcn_ballot_w64(bad) != 0ull;
    const bool any_valid = __builtin_amdgcn_ballot_w64(valid != 0u) != 0ull;
    int n = 0, last = -1;
    bool overflow = false;
    if (!bad && any_valid) {
        for (;;) {
            int cand = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (valid & (1u << j)) {
#pragma unroll
                    for (int i = 0; i < K; ++i) { const int id = ids[j][i]; cand = (id > last && id < cand) ? id : cand; }
                }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
            if (cand == 0x7fffffff) break;
            if (n == DF_BM_NU) { overflow = true; break; }
            float lmin = 3.0e38f, lmax = 0.f, wmin = 3.0e38f, wmax = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (valid & (1u << j)) {
                    float w = 0.f;
#pragma unroll
                    for (int i = 0; i < K; ++i) w = ids[j][i] == cand ? wv[j][i] : w;
                    const float lam = w * inv[j];
                    lmin = fminf(lmin, lam); lmax = fmaxf(lmax, lam); wmin = fminf(wmin, w); wmax = fmaxf(wmax, w);
                }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                lmin = fminf(lmin, __shfl_xor(lmin, o, 64)); lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
                wmin = fminf(wmin, __shfl_xor(wmin, o, 64)); wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
            }
            if (ln == 0) {
                // lambda as the sweep's blend sees it differs from w * (1 / s) by a few ulps: a relative 2^-18 either way
                const uint32_t lp = df_bm_pack(lmin * (1.f - 0x1p-18f), lmax * (1.f + 0x1p-18f));
                s_idx[wave][n] = (uint32_t)cand; s_lam[wave][n] = lp; s_w[wave][n] = df_bm_pack(wmin, wmax);
                s_hw[wave][n] = h2f_bits(lp >> 16);
            }
            last = cand; ++n;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- 4-bit neighbour codes (k = 8; round 6: per 4 x 4 x 4 SUB-block).  A voxel's k neighbours as positions in the union of the
    // neighbour sets of its sub-block (ascending node ids; <= 16 of them: 99 % of the sub-blocks at four times the headline's node
    // density, where a whole block's union fits 16 for one block in four), neighbour i in bits [4 i, 4 i + 4) -- the order of the table
    // record, i.e. of the blend's sums.  Sub-block (h, q): planes 4 h .. 4 h + 3 of column quadrant q = (x >> 2 & 1) | (y >> 2 & 1) << 1,
    // i.e. the 16 lanes that agree in lane bits 2 and 5.  Every round each sub-block takes its smallest id above its last one (a minimum
    // over its 16 lanes: xor 1, 2, 8, 16); a sub-block's e-th round is its union's entry e.  The union lists go out as the dword a sweep lane
    // loads (dfusion_internal.h: bm_ids) -- padded with node 0, which exists.  Independent of the blend model: a block without one (union
    // of the whole block above 16, weights too small to normalise) still gets codes.
    if constexpr (K == 8) {
        if (a.code_tab && a.bm_ids && !any_valid) { if (ln == 0) a.bm_coded[blk] = 0; }
        if (a.code_tab && a.bm_ids && any_valid) {
            unsigned code[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) code[j] = 0u;
            int last_h[2] = {-1, -1};
            bool sub_over = false;
            const bool writer = (ln & 0x1b) == 0;                              // one lane per quadrant: lanes 0, 4, 32, 36
            const unsigned qd = ((unsigned)(ln >> 2) & 1u) | (((unsigned)(ln >> 5) & 1u) << 1);
            uint32_t* ids_out = a.bm_ids + blk * 64;
            ids_out[ln] = 0u;
            __builtin_amdgcn_wave_barrier();
            for (int e = 0;; ++e) {
                int cand_h[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int c = 0x7fffffff;
#pragma unroll
                    for (int j = 4 * h; j < 4 * h + 4; ++j)
                        if (valid & (1u << j)) {
#pragma unroll
                            for (int i = 0; i < K; ++i) { const int id = ids[j][i]; c = (id > last_h[h] && id < c) ? id : c; }
                        }
                    c = min(c, __shfl_xor(c, 1, 64)); c = min(c, __shfl_xor(c, 2, 64)); c = min(c, __shfl_xor(c, 8, 64)); c = min(c, __shfl_xor(c, 16, 64));
                    cand_h[h] = c;
                }
                if (__builtin_amdgcn_ballot_w64(cand_h[0] != 0x7fffffff || cand_h[1] != 0x7fffffff) == 0ull) break;
                if (e == DF_BM_NU) { sub_over = true; break; }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (cand_h[h] == 0x7fffffff) continue;
#pragma unroll
                    for (int j = 4 * h; j < 4 * h + 4; ++j)
                        if (valid & (1u << j)) {
                            unsigned hit = 0u;
#pragma unroll
                            for (int i = 0; i < K; ++i) hit = ids[j][i] == cand_h[h] ? (unsigned)e << (4 * i) : hit;
                            code[j] |= hit;
                        }
                    if (writer) ((uint16_t*)(ids_out + qd * 16 + (unsigned)e))[h] = (uint16_t)cand_h[h];
                    last_h[h] = cand_h[h];
                }
            }
            if (!sub_over) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (col_in && z0 + j < a.Z) a.code_tab[df_code_index(a, x, y, z0 + j)] = code[j];
            }
            __builtin_amdgcn_wave_barrier();
            if (ln == 0) a.bm_coded[blk] = sub_over ? (uint8_t)0 : (uint8_t)1;
        }
    }
    if (ln == 0) blk_state[blk] = 2;                                       // a model record exists (possibly "none")
    if (bad || overflow || n == 0) { if (ln == 0) bm_cnt[blk] = DF_BM_NONE; continue; }
    // entry 0 = the node with the widest lambda interval: its value is the reference v* of the verdict, its own error term vanishes
    float h = ln < n ? s_hw[wave][ln] : -1.f;
    int e0 = ln;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ho = __shfl_xor(h, o, 64); const int eo = __shfl_xor(e0, o, 64);
        if (ho > h || (ho == h && eo < e0)) { h = ho; e0 = eo; }
    }
    if (ln < n) {
        const int dst = ln == e0 ? 0 : ln == 0 ? e0 : ln;
        bm_idx[(size_t)dst * nblk + blk] = (uint16_t)s_idx[wave][ln];
        bm_lam[(size_t)dst * nblk + blk] = s_lam[wave][ln];
        bm_w[(size_t)dst * nblk + blk] = s_w[wave][ln];
    }
    if (ln == 0) bm_cnt[blk] = (uint8_t)n;
    __builtin_amdgcn_wave_barrier();                                       // (the next round reuses the wave's LDS rows)
    }
}

// ---- the box test of one block against its model (n entries): true = no voxel of the block can update this frame
__device__ __forceinline__ bool df_block_box_dead(const DfWarpedArgs& a, const float4* __restrict__ rot, const float4* __restrict__ node_t,
                                                  int nbx, int nby, size_t nblk, size_t blk, unsigned n, const uint16_t* __restrict__ bm_idx,
                                                  const uint32_t* __restrict__ bm_lam, const uint32_t* __restrict__ bm_w)
{
    float slm = 0.f, Nx = 0.f, Ny = 0.f, Nz = 0.f, Ex = 0.f, Ey = 0.f, Ez = 0.f, Dmin = 3.0e38f, Dmax = 0.f;
    float Tx = 0.f, Ty = 0.f, Tz = 0.f, Fx = 0.f, Fy = 0.f, Fz = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    {   const float4 r0 = rot[bm_idx[blk]]; sx = r0.y; sy = r0.z; sz = r0.w; }         // entry 0's value is the reference v*
    // four entries at a time: their records, then their nodes, are requested together (a lane per block walks a chain of dependent
    // loads otherwise -- 35 us for the pass at 512^3); entries past n are clamped to the last one and given zero weight
    for (unsigned e0 = 0; e0 < n; e0 += 4) {
        unsigned id[4]; uint32_t lp[4], wp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const size_t at = (size_t)min(e0 + (unsigned)i, n - 1u) * nblk + blk;
            id[i] = bm_idx[at]; lp[i] = bm_lam[at]; wp[i] = bm_w[at];
        }
        float4 r4[4], t4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { r4[i] = rot[id[i]]; t4[i] = node_t[id[i]]; }       // float4 .x = the quaternion's scalar part
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool on = e0 + (unsigned)i < n;
            const float lm = on ? h2f_bits(lp[i]) : 0.f, lh = on ? h2f_bits(lp[i] >> 16) : 0.f, wm = on ? h2f_bits(wp[i]) : 0.f, wh = on ? h2f_bits(wp[i] >> 16) : 0.f;
            slm += lm;
            Nx += lm * r4[i].y; Ny += lm * r4[i].z; Nz += lm * r4[i].w;
            Ex += lh * fabsf(r4[i].y - sx); Ey += lh * fabsf(r4[i].z - sy); Ez += lh * fabsf(r4[i].w - sz);
            Dmin = fminf(Dmin, r4[i].x); Dmax = fmaxf(Dmax, r4[i].x);                    // (a clamped entry repeats a real one: harmless)
            Tx += wm * t4[i].y; Ty += wm * t4[i].z; Tz += wm * t4[i].w;
            Fx += wh * fabsf(t4[i].y); Fy += wh * fabsf(t4[i].z); Fz += wh * fabsf(t4[i].w);
        }
    }
    const float rest = 1.f - slm;                                          // sum lambda_i = 1, the mids need not
    Nx += rest * sx; Ny += rest * sy; Nz += rest * sz;
    bool dead = false;
    if (Dmin > 0.f) {
        const float id0 = 1.f / Dmin, id1 = 1.f / Dmax;
        DfIv3 U, Q;
        U.x = iv(fminf((Nx - Ex) * id0, (Nx - Ex) * id1), fmaxf((Nx + Ex) * id0, (Nx + Ex) * id1));
        U.y = iv(fminf((Ny - Ey) * id0, (Ny - Ey) * id1), fmaxf((Ny + Ey) * id0, (Ny + Ey) * id1));
        U.z = iv(fminf((Nz - Ez) * id0, (Nz - Ez) * id1), fmaxf((Nz + Ez) * id0, (Nz + Ez) * id1));
        const float umax = fmaxf(fmaxf(fmaxf(fabsf(U.x.lo), fabsf(U.x.hi)), fmaxf(fabsf(U.y.lo), fabsf(U.y.hi))), fmaxf(fabsf(U.z.lo), fabsf(U.z.hi)));
        const bool u_ok = umax < 1.0e3f;                                   // (false for NaN / inf: no 0 * inf inside the interval products below)
        // the block's voxels: canonical positions vol2world * ((x, y, z) * vs), x in [8 bx, 8 bx + 7] (clipped to the volume), ...
        const int bx = (int)(blk % (size_t)nbx), by = (int)((blk / (size_t)nbx) % (size_t)nby), bz = (int)(blk / ((size_t)nbx * nby));
        const int x0 = bx * 8, y0 = by * 8, z0 = a.tab_z0 + bz * 8;
        const int x1 = min(x0 + 7, a.X - 1), y1 = min(y0 + 7, a.Y - 1), z1 = min(z0 + 7, a.Z - 1);
        const float cxv = 0.5f * (float)(x0 + x1) * a.vsx, cyv = 0.5f * (float)(y0 + y1) * a.vsy, czv = 0.5f * (float)(z0 + z1) * a.vsz;
        const float hxv = 0.5f * (float)(x1 - x0) * a.vsx, hyv = 0.5f * (float)(y1 - y0) * a.vsy, hzv = 0.5f * (float)(z1 - z0) * a.vsz;
        const f3 c = aff_mul(a.vol2world, mk3(cxv, cyv, czv));
        const float* M = a.vol2world.R;
        const float hx = fabsf(M[0]) * hxv + fabsf(M[1]) * hyv + fabsf(M[2]) * hzv, hy = fabsf(M[3]) * hxv + fabsf(M[4]) * hyv + fabsf(M[5]) * hzv,
                    hz = fabsf(M[6]) * hxv + fabsf(M[7]) * hyv + fabsf(M[8]) * hzv;
        Q.x = iv(c.x - hx, c.x + hx); Q.y = iv(c.y - hy, c.y + hy); Q.z = iv(c.z - hz, c.z + hz);
        const DfIv3 c1 = iv_cross(U, Q), c2 = iv_cross(U, c1);
        const DfIv ux2 = iv_sq(U.x), uy2 = iv_sq(U.y), uz2 = iv_sq(U.z);
        const DfIv S = iv(2.f / (1.f + (ux2.hi + uy2.hi + uz2.hi)), 2.f / (1.f + (ux2.lo + uy2.lo + uz2.lo)));
        DfIv3 P;
        P.x = iv_add(iv_add(Q.x, iv_mul(S, iv_add(c1.x, c2.x))), iv(Tx - Fx, Tx + Fx));
        P.y = iv_add(iv_add(Q.y, iv_mul(S, iv_add(c1.y, c2.y))), iv(Ty - Fy, Ty + Fy));
        P.z = iv_add(iv_add(Q.z, iv_mul(S, iv_add(c1.z, c2.z))), iv(Tz - Fz, Tz + Fz));
        // world2cam in centre / radius form, then the inflation
        const f3 pc = mk3(0.5f * (P.x.lo + P.x.hi), 0.5f * (P.y.lo + P.y.hi), 0.5f * (P.z.lo + P.z.hi));
        const float prx = 0.5f * (P.x.hi - P.x.lo), pry = 0.5f * (P.y.hi - P.y.lo), prz = 0.5f * (P.z.hi - P.z.lo);
        const f3 cc = aff_mul(a.world2cam, pc);
        const float* R = a.world2cam.R;
        const float mag = fabsf(pc.x) + fabsf(pc.y) + fabsf(pc.z) + fabsf(cc.x) + fabsf(cc.y) + fabsf(cc.z) + prx + pry + prz;
        const float infl = 1e-3f + 1e-5f * mag;
        const float rx = fabsf(R[0]) * prx + fabsf(R[1]) * pry + fabsf(R[2]) * prz + infl, ry = fabsf(R[3]) * prx + fabsf(R[4]) * pry + fabsf(R[5]) * prz + infl,
                    rz = fabsf(R[6]) * prx + fabsf(R[7]) * pry + fabsf(R[8]) * prz + infl;
        const float xl = cc.x - rx, xh = cc.x + rx, yl = cc.y - ry, yh = cc.y + ry, zl = cc.z - rz, zh = cc.z + rz;
        const bool ok = u_ok && mag < 1.0e15f;                             // every bound finite; otherwise the block stays alive
        if (zh <= 0.f) dead = true;                                        // behind the camera (tsdf_volume.cu:86)
        // the nearest point of the box to the camera centre
        const float nx = (xl <= 0.f && xh >= 0.f) ? 0.f : fminf(fabsf(xl), fabsf(xh)), ny = (yl <= 0.f && yh >= 0.f) ? 0.f : fminf(fabsf(yl), fabsf(yh)),
                    nz = (zl <= 0.f && zh >= 0.f) ? 0.f : fminf(fabsf(zl), fabsf(zh));
        const float rmin = sqrtf(nx * nx + ny * ny + nz * nz) * 0.9999f;
        float max_dist;
        if (a.py.top != 0) { const uint32_t tb = df_pyramid_image_max(a.py); max_dist = tb < 0x7c00u ? h2f_bits((uint16_t)tb) : 3.0e38f; }
        else max_dist = a.cull[2];
        if (rmin > max_dist * 1.002f + a.P.trunc) dead = true;             // sdf < -trunc whatever pixel it meets (:91)
        if (!dead && ok && zl > 0.05f) {
            const float il = 1.f / zl, ih = 1.f / zh;
            const float ulo = a.P.fx * fminf(xl * il, xl * ih) + a.P.cx - 2.f, uhi = a.P.fx * fmaxf(xh * il, xh * ih) + a.P.cx + 2.f;
            const float vlo = a.P.fy * fminf(yl * il, yl * ih) + a.P.cy - 2.f, vhi = a.P.fy * fmaxf(yh * il, yh * ih) + a.P.cy + 2.f;
            if (ulo == ulo && uhi == uhi && vlo == vlo && vhi == vhi) {
                if (uhi < 0.f || vhi < 0.f || ulo > (float)(a.P.cols - 1) || vlo > (float)(a.P.rows - 1)) dead = true;      // projects outside the image (:82)
                else if (a.py.top != 0) {
                    const int iu0 = (int)fmaxf(ulo, 0.f), iv0 = (int)fmaxf(vlo, 0.f);
                    const int iu1 = (int)fminf(uhi, (float)(a.P.cols - 1)), iv1 = (int)fminf(vhi, (float)(a.P.rows - 1));
                    const uint32_t dbits = df_pyramid_max_fine(a.py, iu0, iv0, iu1, iv1, 2);
                    if (dbits == 0u) dead = true;                          // no valid depth anywhere it can project to (:86)
                    else if (dbits < 0x7c00u && rmin > h2f_bits((uint16_t)dbits) * 1.002f + a.P.trunc) dead = true;
                }
            }
        }
        dead = dead && ok;
    }
    return dead;
}

// ---- the per-frame verdict pass: one lane per 8 x 8 x 8 block of the table's planes (x fastest: every array below is read coalesced).
//   alive[blk] = 0 where no voxel of the block can update this frame: outside this launch's planes, zero-weight (see DF_ZERO_WEIGHT),
//   culled by the ball test (df_tile_culled, with the block's own bound on sum w_i), or by the box of its blend model.
// (bx < vbx, by < vby: the brick grid of the index.)  It also feeds the ON-DEMAND work of the frame (dfusion_warp.hip, df_block_verdicts):
//   list 0 (URGENT builds)   alive blocks whose tables are not built: built on the launch stream before the sweep;
//   list 1 (LOOK-AHEAD builds, a.pf_margin > 0) blocks that are not alive now but pass the ball test with every radius widened by
//          a.pf_margin -- what a moving camera / a changing warp brings in over the next few frames: built on the handle's SIDE stream
//          while this frame's sweep runs, so that a block is usually built (and modelled) before it is first swept;
//   list 2 (models, when `want_models`) built blocks without a model record that are alive or near: also side-stream work (a model
//          only serves from the next frame on).  With a.pf_margin == 0 list 1 stays empty and everything runs on the launch stream.
// Builds are packed brick coordinates, models block indices.  Counter set `cnt`: [0] urgent builds, [1] models, [2] the urgent build
// pass's cursor, [3] look-ahead builds, [4] their pass's cursor; this pass zeroes the other set, `cnt_next`.
#ifdef DF_TRACE_VERDICT          // per-wave timeline of the verdict pass (tools/trace_verdict.py): start, after the ball test, after the box test, end
__device__ unsigned long long g_df_vtrace[8192 * 4];
#define DF_VT(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x * 4 + (threadIdx.x >> 6) < 8192) g_df_vtrace[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (i)] = wall_clock64(); } while (0)
#else
#define DF_VT(i) do { } while (0)
#endif
__global__ __launch_bounds__(256) void df_block_verdict_kernel(const DfWarpedArgs a, const float4* __restrict__ rot, const float4* __restrict__ node_t,
                                                               int nbx, int nby, int nbz, uint8_t* __restrict__ blk_state,
                                                               const float* __restrict__ blk_wmax, const float* __restrict__ brick_d1, int vbx, int vby, int zero_skip, int use_models, int want_models,
                                                               int build_on_demand, const uint16_t* __restrict__ bm_idx,
                                                               const uint32_t* __restrict__ bm_lam, const uint32_t* __restrict__ bm_w,
                                                               const uint8_t* __restrict__ bm_cnt, const uint8_t* __restrict__ bm_coded, uint8_t* __restrict__ alive,
                                                               uint32_t* __restrict__ build_list, uint32_t* __restrict__ ahead_list, uint32_t* __restrict__ model_list,
                                                               uint32_t* __restrict__ cnt, uint32_t* __restrict__ cnt_next)
{
    const size_t nblk = (size_t)nbx * nby * nbz;
    const size_t blk = (size_t)blockIdx.x * 256 + threadIdx.x;
    DF_VT(0);
    if (blockIdx.x == 0 && threadIdx.x < 8) cnt_next[threadIdx.x] = 0u;
    bool keep = false, near = false, need_build = false, need_ahead = false, need_model = false;
    int bx = 0, by = 0, bz = 0;
    if (blk < nblk) {
        bx = (int)(blk % (size_t)nbx); by = (int)((blk / (size_t)nbx) % (size_t)nby); bz = (int)(blk / ((size_t)nbx * nby));
        const int x0 = bx * 8, y0 = by * 8, z0 = a.tab_z0 + bz * 8;
        const int own1 = min(a.z_own0 + a.z_own_n, a.Z);
        keep = x0 < a.X && y0 < a.Y && max(z0, a.z_own0) < min(z0 + 8, own1);
        const unsigned st = keep ? blk_state[blk] : 0u;
        if (keep) {
            float wk = a.kf;
            if (zero_skip && st >= 1u) {
                const float wmax = blk_wmax[blk];
                keep = !(wmax * a.cull[3] < DF_ZERO_WEIGHT);               // nothing in the block can update
                wk = fminf(wk, wmax * 1.0001f);                            // |sum w_i t_i| <= (sum w_i) max |t_i|
            } else if (zero_skip && a.cull[4] < 1.0e30f && a.cull[4] > 0.f) {
                // tables not built: every weight of the block is exp(-d^2 / (2 dg_w^2)) with d >= (distance of the nearest node from
                // the brick's centre) - (the lattice's half diagonal) and dg_w <= the node set's largest -- a bound on sum w_i that
                // needs no table.  A third of the volume is zero-weight (farther than ~10 sigma from every node): never built.
                const float dmin = fmaxf(brick_d1[((size_t)(a.tab_z0 / 8 + bz) * vby + by) * vbx + bx] - a.tile_r, 0.f) * 0.999f;
                const float sg = a.cull[4];
                const float wmax = a.kf * 1.001f * __expf(-(dmin * dmin) / (2.f * sg * sg) * 0.998f);
                keep = !(wmax * a.cull[3] < DF_ZERO_WEIGHT);
                wk = fminf(wk, wmax);
            }
            if (keep) {
                const f3 c = aff_mul(a.vol2world, mk3(((float)x0 + 3.5f) * a.vsx, ((float)y0 + 3.5f) * a.vsy, ((float)z0 + 3.5f) * a.vsz));
                if (a.pf_margin > 0.f && st < 2u) {
                    // (only blocks that still lack something can be look-ahead work: the widened test first, the frame's own
                    // test for those that pass it)
                    near = !df_tile_culled(a, c, wk, a.pf_margin);
                    keep = near && !df_tile_culled(a, c, wk);
                } else keep = !df_tile_culled(a, c, wk);
            }
        }
        DF_VT(1);
        if (keep && st == 2u && (use_models & 1)) {
            const unsigned n = bm_cnt[blk];
            if (n != DF_BM_NONE && a.cull[1] <= 1.0f) keep = !df_block_box_dead(a, rot, node_t, nbx, nby, nblk, blk, n, bm_idx, bm_lam, bm_w);
        }
        DF_VT(2);
        // bit 1: the model pass has been over the block -- it has union lists and 4-bit neighbour codes -- and was COMPLETE before this pass
        // started (the state byte was read above, after the previous frame's side-stream work was joined).  The plan kernel takes "coded"
        // from here and not from the state bytes: it runs beside THIS frame's model builds, which set them before their codes are all written.
        alive[blk] = keep ? (uint8_t)(1u | ((st == 2u && (use_models & 2) && bm_coded[blk] != 0) ? 2u : 0u)) : (uint8_t)0;
        need_build = keep && build_on_demand && st == 0u;
        need_ahead = !keep && near && build_on_demand && st == 0u;
        need_model = (keep || near) && want_models && (st == 1u || (need_build && want_models > 1));
        if (need_build) blk_state[blk] = 1;                                // (look-ahead blocks: only those that get a slot below)
    }
    // one counter bump per wave and list
    const unsigned long long mb = __builtin_amdgcn_ballot_w64(need_build), ma = __builtin_amdgcn_ballot_w64(need_ahead), mm = __builtin_amdgcn_ballot_w64(need_model);
    const unsigned long long below = ((unsigned long long)1 << (threadIdx.x & 63)) - 1ull;
    const unsigned code = (unsigned)bx | ((unsigned)by << 10) | ((unsigned)(a.tab_z0 / 8 + bz) << 20);
    if (mb) {
        unsigned base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&cnt[0], (unsigned)__popcll(mb));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (need_build) build_list[base + (unsigned)__popcll(mb & below)] = code;
    }
    if (ma) {
        // at most a.pf_cap look-ahead builds per frame (the pass that makes them takes min(count, cap)): the shell around a NEW alive
        // set is thousands of blocks, and built all at once the side stream would still be busy when the next frame has to wait for
        // it.  A block that gets no slot stays unbuilt and is listed again next frame.
        unsigned base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&cnt[3], (unsigned)__popcll(ma));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        const unsigned slot = base + (unsigned)__popcll(ma & below);
        if (need_ahead && slot < a.pf_cap) { ahead_list[slot] = code; blk_state[blk] = 1; }
    }
    if (mm) {
        unsigned base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&cnt[1], (unsigned)__popcll(mm));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (need_model) model_list[base + (unsigned)__popcll(mm & below)] = (unsigned)blk;
    }
    DF_VT(3);
}

// ---- every block whose tables are not built yet, onto the build list (completing on-demand tables for a sweep that has no verdicts)
__global__ __launch_bounds__(256) void df_blocks_unbuilt_kernel(int nbx, int nby, int nbz, int vbx, int vby, int tab_bz0, uint8_t* __restrict__ blk_state,
                                                                uint32_t* __restrict__ build_list, uint32_t* __restrict__ cnt,
                                                                uint32_t* __restrict__ cnt_next)
{
    const size_t nblk = (size_t)nbx * nby * nbz;
    const size_t blk = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 8) cnt_next[threadIdx.x] = 0u;
    bool need = false;
    int bx = 0, by = 0, bz = 0;
    if (blk < nblk) {
        bx = (int)(blk % (size_t)nbx); by = (int)((blk / (size_t)nbx) % (size_t)nby); bz = (int)(blk / ((size_t)nbx * nby));
        need = blk_state[blk] == 0 && bx < vbx && by < vby;               // (blocks of the table's padding hold no voxels)
        if (blk_state[blk] == 0) blk_state[blk] = 1;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(need);
    if (m) {
        unsigned base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(&cnt[0], (unsigned)__popcll(m));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (need) build_list[base + (unsigned)__popcll(m & (((unsigned long long)1 << (threadIdx.x & 63)) - 1ull))] =
                      (unsigned)bx | ((unsigned)by << 10) | ((unsigned)(tab_bz0 + bz) << 20);
    }
}
