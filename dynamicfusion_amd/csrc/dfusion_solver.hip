// dfusion_solver.hip -- the warp-field data term on the GPU (gfx950), SURVEY.md 8(f) "next" #4.
//
// Replaces WarpFieldOptimiser::optimiseWarpData -> CombinedSolver (Opt, kfusion/solvers/dynamicfusion.t:26-52) and
// WarpField::energy_data (Ceres, kfusion/src/warp_field.cpp:117-163 with the functor of
// kfusion/include/kfusion/optimisation.hpp:36-71).  Both hand the SAME energy to a third-party non-linear solver:
//     E(T) = sum_v | (live_v - canonical_v) - sum_{i<k} w_vi * T_{n_vi} |^2
// with n_vi / w_vi the k nearest nodes of canonical_v and their weights exp(-d^2 / 2 sigma^2) (getWeightsAndUpdateKNN); only
// the node TRANSLATIONS T are unknowns (RotationDeform is declared but unused in dynamicfusion.t; the Ceres functor reads only
// the translation slots), and there is no regularisation term in the reference (optimisation.hpp:125-157 is never added).
// E is linear least squares, so Gauss-Newton is one linear solve: here `iters` steps of conjugate gradients on the normal
// equations (W^T W + lambda I) delta = W^T e0, e0 the residual at the current translations, matrix-free:
//   W   (N x M, k non-zeros per row)  : one lane per point, gathers its k node entries in slot order;
//   W^T (M x N)                       : a node-major copy of the entries (stable radix sort by node id, so every node's list
//                                       is in ascending point order), one workgroup per node, thread-strided partial sums then
//                                       a fixed tree -- no float atomics, so the result is reproducible (and bit-comparable
//                                       with the restatement in oracle/dfusion_frontend_oracle.c);
//   dot products / vector updates     : one 1024-thread workgroup (M <= 65535), fixed tree.
// Nothing returns to the host between iterations: once every component has converged (scal[3] == 0) the kernels of the remaining
// steps return at once.
#include <hipcub/hipcub.hpp>
#include "dfusion_internal.h"

#pragma clang fp contract(off)

#define SV_BLOCK 1024

// ---- per point: validity, weights, node ids (sort keys), residual at the current translations
__global__ __launch_bounds__(256) void df_sv_setup_kernel(const float* __restrict__ canonical, const float* __restrict__ live, int N, int k,
                                                          const int* __restrict__ idx, const float* __restrict__ d2,
                                                          const float4* __restrict__ pos_sigma, const float4* __restrict__ node_t, int M,
                                                          float* __restrict__ w, unsigned int* __restrict__ keys,
                                                          unsigned int* __restrict__ vals, float* __restrict__ e0)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= N) return;
    const float cx = canonical[3 * v], cy = canonical[3 * v + 1], cz = canonical[3 * v + 2];
    const float lx = live[3 * v], ly = live[3 * v + 1], lz = live[3 * v + 2];
    const bool valid = !(isnan(cx) || isnan(cy) || isnan(cz) || isnan(lx) || isnan(ly) || isnan(lz));   // warp_field.cpp:130-136
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int j = 0; j < k; ++j) {
        const int e = v * k + j;
        const int n = valid ? idx[e] : M;                               // invalid points sort behind every node
        float wj = 0.f;
        if (valid) {
            wj = dqb_weight(d2[e], pos_sigma[n].w);                     // warp_field.cpp:238-241
            const float4 t = node_t[n];                                 // (w, x, y, z): translation in .y .z .w
            sx = sx + wj * t.y; sy = sy + wj * t.z; sz = sz + wj * t.w; // optimisation.hpp:58-60
        }
        w[e] = wj;
        keys[e] = (unsigned int)n;
        vals[e] = (unsigned int)e;
    }
    e0[3 * v] = valid ? (lx - cx) - sx : 0.f;                           // optimisation.hpp:64-66
    e0[3 * v + 1] = valid ? (ly - cy) - sy : 0.f;
    e0[3 * v + 2] = valid ? (lz - cz) - sz : 0.f;
}

// ---- node offsets into the sorted entry list: off[n] = first position whose key >= n (off[M] = number of valid entries)
__global__ __launch_bounds__(256) void df_sv_offsets_kernel(const unsigned int* __restrict__ sorted_keys, int E, int M, unsigned int* __restrict__ off)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > E) return;
    const unsigned int cur = i < E ? min(sorted_keys[i], (unsigned int)M) : (unsigned int)M;
    const unsigned int prev = i > 0 ? min(sorted_keys[i - 1], (unsigned int)M) : 0u;
    const unsigned int lo = i > 0 ? prev + 1 : 0u;
    for (unsigned int n = lo; n <= cur; ++n) off[n] = (unsigned int)i;
}

// ---- u = W p : per point, slot order
// K > 0: compile-time neighbour count, so that the 2K entry loads and 3K gathers of a point are all in flight before the first sum
// (with a run-time k the loop issues and waits entry by entry); K = 0: any k.  Same sums in the same order either way.
template <int K>
__global__ __launch_bounds__(256) void df_sv_w_apply_kernel(const float* __restrict__ w, const unsigned int* __restrict__ keys, int N, int k,
                                                            int M, const float* __restrict__ p, float* __restrict__ u,
                                                            const float* __restrict__ active)
{
    // `active` (nullable): scal[3], the number of CG components still iterating.  Once it is 0 every further step is an exact no-op
    // (alpha = beta = 0), so the kernels of the remaining steps return at once -- the host enqueues all steps without looking.
    if (active && *active == 0.f) return;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= N) return;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if constexpr (K > 0) {
        unsigned int n[K]; float wj[K], px[K], py[K], pz[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { n[j] = keys[v * K + j]; wj[j] = w[v * K + j]; }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const unsigned int nc = min(n[j], (unsigned int)(M - 1));            // clamped gather; the entry is skipped below if n >= M
            px[j] = p[3 * nc]; py[j] = p[3 * nc + 1]; pz[j] = p[3 * nc + 2];
        }
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (n[j] < (unsigned int)M) { sx = sx + wj[j] * px[j]; sy = sy + wj[j] * py[j]; sz = sz + wj[j] * pz[j]; }
    } else {
        for (int j = 0; j < k; ++j) {
            const int e = v * k + j;
            const unsigned int n = keys[e];
            if (n < (unsigned int)M) {
                const float wj = w[e];
                sx = sx + wj * p[3 * n]; sy = sy + wj * p[3 * n + 1]; sz = sz + wj * p[3 * n + 2];
            }
        }
    }
    u[3 * v] = sx; u[3 * v + 1] = sy; u[3 * v + 2] = sz;
}

// ---- node-major copies of the entries' point ids and weights (once per solve; the W^T kernel then reads them in list order)
__global__ __launch_bounds__(256) void df_sv_sorted_kernel(const unsigned int* __restrict__ sorted_vals, const float* __restrict__ w, int E, int k,
                                                           unsigned int* __restrict__ sorted_pt, float* __restrict__ sorted_w)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= E) return;
    const unsigned int e = sorted_vals[i];
    sorted_pt[i] = e / (unsigned int)k;
    sorted_w[i] = w[e];
}

// ---- out = W^T u (+ lambda * p) : one 256-thread workgroup per node over its (ascending-point-order) entry list.
// Thread t sums entries t, t + 256, ... ; the 256 partials are combined by the tree (t, t + s), s = 128 .. 1 (s >= 64 through
// LDS, s <= 32 with __shfl_down inside wave 0) -- a fixed order, restated in the oracle.
__global__ __launch_bounds__(256) void df_sv_wt_apply_kernel(const unsigned int* __restrict__ off, const unsigned int* __restrict__ sorted_pt,
                                                             const float* __restrict__ sorted_w, int M, const float* __restrict__ u,
                                                             float lambda, const float* __restrict__ p, float* __restrict__ out,
                                                             const float* __restrict__ active)
{
    __shared__ float lds[3 * 128];
    if (active && *active == 0.f) return;                               // see df_sv_w_apply_kernel
    const int n = blockIdx.x, t = threadIdx.x, wv = t >> 6, l = t & 63;
    const unsigned int b = off[n], e_end = off[n + 1];
    float s[3] = {0.f, 0.f, 0.f};
    // the entry's point id and weight sit in list order (df_sv_sorted_kernel), so a trip is one coalesced read + the u gather
    for (unsigned int i = b + t; i < e_end; i += 256) {
        const unsigned int v = sorted_pt[i];
        const float we = sorted_w[i];
        s[0] = s[0] + we * u[3 * v]; s[1] = s[1] + we * u[3 * v + 1]; s[2] = s[2] + we * u[3 * v + 2];
    }
    if (wv >= 2) { for (int c = 0; c < 3; ++c) lds[c * 128 + (t - 128)] = s[c]; }
    __syncthreads();
    if (wv < 2) { for (int c = 0; c < 3; ++c) s[c] = s[c] + lds[c * 128 + t]; }
    __syncthreads();
    if (wv == 1) { for (int c = 0; c < 3; ++c) lds[c * 128 + l] = s[c]; }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = s[c] + lds[c * 128 + l];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) a = a + __shfl_down(a, o, 64);
            s[c] = a;
        }
        if (l == 0) {
            if (p) { s[0] = s[0] + lambda * p[3 * n]; s[1] = s[1] + lambda * p[3 * n + 1]; s[2] = s[2] + lambda * p[3 * n + 2]; }
            out[3 * n] = s[0]; out[3 * n + 1] = s[1]; out[3 * n + 2] = s[2];
        }
    }
}

// ---- single-workgroup vector algebra: three independent CG recurrences (x, y, z components share the matrix)
// block-wide sums of three values: thread t owns elements t, t + 1024, ... ; then the tree 512 .. 1 in LDS
__device__ __forceinline__ void sv_block_sum3(float (&s)[3], float* lds /* [3][1024] */)
{
    const int t = threadIdx.x;
#pragma unroll
    for (int c = 0; c < 3; ++c) lds[c * SV_BLOCK + t] = s[c];
    __syncthreads();
    for (int st = SV_BLOCK / 2; st >= 64; st >>= 1) {                   // pairs (t, t + st)
        if (t < st) {
#pragma unroll
            for (int c = 0; c < 3; ++c) lds[c * SV_BLOCK + t] = lds[c * SV_BLOCK + t] + lds[c * SV_BLOCK + t + st];
        }
        __syncthreads();
    }
    if (t < 64) {                                                        // st = 32 .. 1 inside wave 0: the same pairs
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = lds[c * SV_BLOCK + t];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) a = a + __shfl_down(a, o, 64);
            if (t == 0) lds[c * SV_BLOCK] = a;
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 3; ++c) s[c] = lds[c * SV_BLOCK];
    __syncthreads();
}

// scal: [0..2] rr (0 = component converged / frozen), [3] number of components still iterating, [4] initial energy,
// [5] final energy, [6..8] rr of the first residual (the convergence test is relative to it)
#define SV_REL_TOL2 1.0e-10f        // stop a component once |r|^2 <= 1e-10 |r0|^2
__global__ __launch_bounds__(SV_BLOCK) void df_sv_init_kernel(const float* __restrict__ r, int M, float* __restrict__ x, float* __restrict__ p,
                                                              float* __restrict__ scal)
{
    __shared__ float lds[3 * SV_BLOCK];
    float s[3] = {0.f, 0.f, 0.f};
    for (int n = threadIdx.x; n < M; n += SV_BLOCK)
        for (int c = 0; c < 3; ++c) { const float rv = r[3 * n + c]; x[3 * n + c] = 0.f; p[3 * n + c] = rv; s[c] = s[c] + rv * rv; }
    sv_block_sum3(s, lds);
    if (threadIdx.x == 0) { scal[0] = s[0]; scal[1] = s[1]; scal[2] = s[2]; scal[6] = s[0]; scal[7] = s[1]; scal[8] = s[2];
                            scal[3] = (float)((s[0] > 0.f) + (s[1] > 0.f) + (s[2] > 0.f)); }
}

__global__ __launch_bounds__(SV_BLOCK) void df_sv_step_kernel(const float* __restrict__ q, int M, float* __restrict__ x, float* __restrict__ r,
                                                              float* __restrict__ p, float* __restrict__ scal)
{
    __shared__ float lds[3 * SV_BLOCK];
    if (scal[3] == 0.f) return;                                         // all components frozen: the step would change nothing
    float pq[3] = {0.f, 0.f, 0.f};
    for (int n = threadIdx.x; n < M; n += SV_BLOCK)
        for (int c = 0; c < 3; ++c) pq[c] = pq[c] + p[3 * n + c] * q[3 * n + c];
    sv_block_sum3(pq, lds);
    float alpha[3], rr_old[3];
    for (int c = 0; c < 3; ++c) {
        rr_old[c] = scal[c];
        alpha[c] = (pq[c] > 0.f && rr_old[c] > 0.f) ? rr_old[c] / pq[c] : 0.f;        // converged / degenerate component: frozen
    }
    float rr[3] = {0.f, 0.f, 0.f};
    for (int n = threadIdx.x; n < M; n += SV_BLOCK)
        for (int c = 0; c < 3; ++c) {
            x[3 * n + c] = x[3 * n + c] + alpha[c] * p[3 * n + c];
            const float rv = r[3 * n + c] - alpha[c] * q[3 * n + c];
            r[3 * n + c] = rv;
            rr[c] = rr[c] + rv * rv;
        }
    sv_block_sum3(rr, lds);
    float beta[3];
    for (int c = 0; c < 3; ++c) beta[c] = (alpha[c] != 0.f && rr_old[c] > 0.f) ? rr[c] / rr_old[c] : 0.f;
    for (int n = threadIdx.x; n < M; n += SV_BLOCK)
        for (int c = 0; c < 3; ++c) p[3 * n + c] = r[3 * n + c] + beta[c] * p[3 * n + c];
    __syncthreads();
    if (threadIdx.x == 0) {
        float active = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float keep = (alpha[c] != 0.f && rr[c] > SV_REL_TOL2 * scal[6 + c]) ? rr[c] : 0.f;
            scal[c] = keep;
            active += keep > 0.f ? 1.f : 0.f;
        }
        scal[3] = active;
    }
}

// df_sv_step_kernel for M <= EPT * SV_BLOCK: x, r, p, q are read once and stay in registers between the three passes (the passes of
// the kernel above each wait for their own global loads).  Same element -> thread assignment, same sums, same trees.
template <int EPT>
__global__ __launch_bounds__(SV_BLOCK) void df_sv_step_reg_kernel(const float* __restrict__ q, int M, float* __restrict__ x, float* __restrict__ r,
                                                                  float* __restrict__ p, float* __restrict__ scal)
{
    __shared__ float lds[3 * SV_BLOCK];
    if (scal[3] == 0.f) return;                                         // all components frozen: the step would change nothing
    float pv[EPT][3], qv[EPT][3], rv[EPT][3], xv[EPT][3];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int n = threadIdx.x + j * SV_BLOCK;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool in = n < M;
            pv[j][c] = in ? p[3 * n + c] : 0.f; qv[j][c] = in ? q[3 * n + c] : 0.f;
            rv[j][c] = in ? r[3 * n + c] : 0.f; xv[j][c] = in ? x[3 * n + c] : 0.f;
        }
    }
    float pq[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (threadIdx.x + j * SV_BLOCK < M)
#pragma unroll
            for (int c = 0; c < 3; ++c) pq[c] = pq[c] + pv[j][c] * qv[j][c];
    sv_block_sum3(pq, lds);
    float alpha[3], rr_old[3];
    for (int c = 0; c < 3; ++c) {
        rr_old[c] = scal[c];
        alpha[c] = (pq[c] > 0.f && rr_old[c] > 0.f) ? rr_old[c] / pq[c] : 0.f;
    }
    float rr[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < EPT; ++j)
        if (threadIdx.x + j * SV_BLOCK < M)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                xv[j][c] = xv[j][c] + alpha[c] * pv[j][c];
                rv[j][c] = rv[j][c] - alpha[c] * qv[j][c];
                rr[c] = rr[c] + rv[j][c] * rv[j][c];
            }
    sv_block_sum3(rr, lds);
    float beta[3];
    for (int c = 0; c < 3; ++c) beta[c] = (alpha[c] != 0.f && rr_old[c] > 0.f) ? rr[c] / rr_old[c] : 0.f;
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int n = threadIdx.x + j * SV_BLOCK;
        if (n < M)
#pragma unroll
            for (int c = 0; c < 3; ++c) { x[3 * n + c] = xv[j][c]; r[3 * n + c] = rv[j][c]; p[3 * n + c] = rv[j][c] + beta[c] * pv[j][c]; }
    }
    if (threadIdx.x == 0) {
        float active = 0.f;
        for (int c = 0; c < 3; ++c) {
            const float keep = (alpha[c] != 0.f && rr[c] > SV_REL_TOL2 * scal[6 + c]) ? rr[c] : 0.f;
            scal[c] = keep;
            active += keep > 0.f ? 1.f : 0.f;
        }
        scal[3] = active;
    }
}

// ---- energy = sum_v |e_v|^2 (single workgroup, same tree)
__global__ __launch_bounds__(SV_BLOCK) void df_sv_energy_kernel(const float* __restrict__ e, int N, float* __restrict__ out)
{
    __shared__ float lds[3 * SV_BLOCK];
    float s[3] = {0.f, 0.f, 0.f};
    for (int v = threadIdx.x; v < N; v += SV_BLOCK)
        for (int c = 0; c < 3; ++c) s[c] = s[c] + e[3 * v + c] * e[3 * v + c];
    sv_block_sum3(s, lds);
    if (threadIdx.x == 0) *out = (s[0] + s[1]) + s[2];
}

// e1 = e0 - W x  (final residual, for the reported energy)
__global__ __launch_bounds__(256) void df_sv_residual_kernel(const float* __restrict__ e0, const float* wx, int n3, float* e1)   // e1 may alias wx
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n3) e1[i] = e0[i] - wx[i];
}

// ---- write back: T = T0 + x ; translation_ = 0.5 * (0, T) * rotation_ (encodeTranslation, dual_quaternion.hpp:82-85)
__global__ __launch_bounds__(256) void df_sv_writeback_kernel(const float4* __restrict__ rot, const float4* __restrict__ node_t,
                                                              const float* __restrict__ x, int M, float* __restrict__ dq_out)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= M) return;
    const float4 r4 = rot[n], t4 = node_t[n];
    quat r; r.w = r4.x; r.x = r4.y; r.y = r4.z; r.z = r4.w;
    quat T; T.w = 0.f; T.x = t4.y + x[3 * n]; T.y = t4.z + x[3 * n + 1]; T.z = t4.w + x[3 * n + 2];
    const quat d = q_mul(q_scale(0.5f, T), r);
    float* o = dq_out + 8 * (size_t)n;
    o[0] = r.w; o[1] = r.x; o[2] = r.y; o[3] = r.z; o[4] = d.w; o[5] = d.x; o[6] = d.y; o[7] = d.z;
}

static int sv_reserve(DfWarpField* wf, size_t bytes)
{
    if (bytes <= wf->solver_ws_cap) return DF_OK;
    (void)hipFree(wf->solver_ws); wf->solver_ws = nullptr; wf->solver_ws_cap = 0;
    DF_HIP(hipMalloc(&wf->solver_ws, bytes));
    wf->solver_ws_cap = bytes;
    return DF_OK;
}

extern "C" int dfusion_warp_solve_data_term(DfWarpField* wf, int k, const float* canonical, const float* live, int N, int iters, float lambda,
                                            float* dq_out, float* energy, dfStream stream)
{
    if (!wf || !canonical || !live || N <= 0 || iters < 0 || !(lambda >= 0.f) || wf->M <= 0 || k < 1 || k > 8 || wf->M < k) return DF_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int M = wf->M;
    const size_t E = (size_t)N * k;
    if (E > 0x7fffffffu) return DF_E_INVALID;
    // workspace carve-up (256-byte aligned pieces)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_idx = take(E * 4), o_d2 = take(E * 4), o_w = take(E * 4), o_keys = take(E * 4), o_vals = take(E * 4), o_skeys = take(E * 4),
                 o_svals = take(E * 4), o_spt = take(E * 4), o_sw = take(E * 4), o_e0 = take((size_t)N * 12), o_u = take((size_t)N * 12), o_off = take(((size_t)M + 2) * 4),
                 o_x = take((size_t)M * 12), o_r = take((size_t)M * 12), o_p = take((size_t)M * 12), o_q = take((size_t)M * 12),
                 o_scal = take(64), o_dq = take((size_t)M * 32);
    size_t sort_bytes = 0;
    DF_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const unsigned int*)nullptr, (unsigned int*)nullptr, (const unsigned int*)nullptr,
                                              (unsigned int*)nullptr, (int)E, 0, 17, st));
    const size_t o_sort = take(sort_bytes);
    int rc = sv_reserve(wf, off);
    if (rc) return rc;
    char* ws = (char*)wf->solver_ws;
    int* idx = (int*)(ws + o_idx); float* d2 = (float*)(ws + o_d2); float* w = (float*)(ws + o_w);
    unsigned int* keys = (unsigned int*)(ws + o_keys); unsigned int* vals = (unsigned int*)(ws + o_vals);
    unsigned int* skeys = (unsigned int*)(ws + o_skeys); unsigned int* svals = (unsigned int*)(ws + o_svals);
    unsigned int* spt = (unsigned int*)(ws + o_spt); float* sw = (float*)(ws + o_sw);
    float* e0 = (float*)(ws + o_e0); float* u = (float*)(ws + o_u); unsigned int* offs = (unsigned int*)(ws + o_off);
    float* x = (float*)(ws + o_x); float* r = (float*)(ws + o_r); float* p = (float*)(ws + o_p); float* q = (float*)(ws + o_q);
    float* scal = (float*)(ws + o_scal); float* dq = (float*)(ws + o_dq);

    rc = dfusion_knn(wf, k, canonical, N, idx, d2, stream);           // getWeightsAndUpdateKNN's k-NN (NaN queries are masked below)
    if (rc) return rc;
    const dim3 gN((N + 255) / 256), gM((M + 255) / 256);
    hipLaunchKernelGGL(df_sv_setup_kernel, gN, dim3(256), 0, st, canonical, live, N, k, idx, d2, wf->pos_sigma, wf->node_t, M, w, keys, vals, e0);
    DF_LAUNCH_CHECK();
    DF_HIP(hipcub::DeviceRadixSort::SortPairs(ws + o_sort, sort_bytes, keys, skeys, vals, svals, (int)E, 0, 17, st));   // stable
    hipLaunchKernelGGL(df_sv_offsets_kernel, dim3((unsigned)((E + 1 + 255) / 256)), dim3(256), 0, st, skeys, (int)E, M, offs);
    DF_LAUNCH_CHECK();
    hipLaunchKernelGGL(df_sv_sorted_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, svals, w, (int)E, k, spt, sw);
    DF_LAUNCH_CHECK();
    if (energy) { hipLaunchKernelGGL(df_sv_energy_kernel, dim3(1), dim3(SV_BLOCK), 0, st, e0, N, scal + 4); DF_LAUNCH_CHECK(); }
    auto step = M <= 2 * SV_BLOCK ? df_sv_step_reg_kernel<2> : M <= 5 * SV_BLOCK ? df_sv_step_reg_kernel<5> : M <= 8 * SV_BLOCK ? df_sv_step_reg_kernel<8> : df_sv_step_kernel;
    auto w_apply = k == 8 ? df_sv_w_apply_kernel<8> : k == 4 ? df_sv_w_apply_kernel<4> : df_sv_w_apply_kernel<0>;
    // r0 = W^T e0 ; p0 = r0 ; x0 = 0
    const dim3 gW(M);
    hipLaunchKernelGGL(df_sv_wt_apply_kernel, gW, dim3(256), 0, st, offs, spt, sw, M, e0, 0.f, (const float*)nullptr, r, (const float*)nullptr);
    DF_LAUNCH_CHECK();
    hipLaunchKernelGGL(df_sv_init_kernel, dim3(1), dim3(SV_BLOCK), 0, st, r, M, x, p, scal);
    DF_LAUNCH_CHECK();
    for (int it = 0; it < iters; ++it) {
        hipLaunchKernelGGL(w_apply, gN, dim3(256), 0, st, w, keys, N, k, M, p, u, (const float*)(scal + 3));
        hipLaunchKernelGGL(df_sv_wt_apply_kernel, gW, dim3(256), 0, st, offs, spt, sw, M, u, lambda, p, q, (const float*)(scal + 3));
        hipLaunchKernelGGL(step, dim3(1), dim3(SV_BLOCK), 0, st, q, M, x, r, p, scal);
        DF_LAUNCH_CHECK();
    }
    if (energy) {
        hipLaunchKernelGGL(w_apply, gN, dim3(256), 0, st, w, keys, N, k, M, x, u, (const float*)nullptr);
        hipLaunchKernelGGL(df_sv_residual_kernel, dim3((unsigned)((3 * (size_t)N + 255) / 256)), dim3(256), 0, st, e0, u, 3 * N, u);
        hipLaunchKernelGGL(df_sv_energy_kernel, dim3(1), dim3(SV_BLOCK), 0, st, u, N, scal + 5);
        DF_LAUNCH_CHECK();
        DF_HIP(hipMemcpyAsync(energy, scal + 4, 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    hipLaunchKernelGGL(df_sv_writeback_kernel, gM, dim3(256), 0, st, wf->rot, wf->node_t, x, M, dq);
    DF_LAUNCH_CHECK();
    if (dq_out) DF_HIP(hipMemcpyAsync(dq_out, dq, (size_t)M * 32, hipMemcpyDeviceToDevice, st));
    return dfusion_warp_set_transforms(wf, dq, stream);
}
