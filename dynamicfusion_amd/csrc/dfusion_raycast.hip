// dfusion_raycast.hip -- surface ray-casting of the TSDF volume (gfx950).
//
// Replaces /root/reference/kfusion/src/cuda/tsdf_volume.cu:202-474 (intersect, interpolate,
// TsdfRaycaster, raycast_kernel x2, device::raycast x2).
//
// MI355X design notes
//   * A wave64 owns an 8x8 pixel tile (the reference's warp owns a 32x1 strip): neighbouring rays
//     stay within a few voxels of each other along the march, so the 4-byte nearest-voxel gathers of
//     a wave fall into few 128-byte lines.  Workgroups (16x16 pixels) are renumbered so that each
//     XCD (block id % 8) marches one contiguous band of the image and keeps its part of the volume in
//     its own 4 MiB L2.
//   * The zero-crossing march only RECORDS the event (step index, kind); the expensive refinement
//     (2 + 6 trilinear interpolations = 64 gathers) runs after the loop, when the wave has
//     reconverged, instead of stalling 63 marching lanes inside it.
//   * The march is issue-bound, not memory-bound (4 steps' gathers are in flight together): its addresses are made in 32 bits with
//     full-rate instructions (rc_march_addr) and the event search runs only for windows that hold a negative sample (rc_march).
//   * Z-slab sharding: every GPU marches the same global step lattice t_k = tmin + k*time_step (the
//     float accumulations `tcurr += time_step`, `next += vstep` are replayed identically) but only
//     evaluates the steps whose `curr` sample lies in a plane it owns; the per-pixel event key
//     (k<<1 | hit) is min-merged across GPUs by the host layer.
#include "dfusion_internal.h"
#include <stdio.h>
#include <stdlib.h>

struct DfRayArgs {
    const uint32_t* vol; int X, Y, Z;
    int z_store0, z_store_n, z_own0, z_own1;
    float vsx, vsy, vsz, vsix, vsiy, vsiz, gdx, gdy, gdz, sizex, sizey, sizez;
    float time_step;
    DfAff aff; float Rinv[9];
    float finvx, finvy, cx, cy;
    int cols, rows;
    float* pts; size_t ppitch; float* nrm; size_t npitch; uint16_t* depth; size_t dpitch;
    uint32_t* keys;
    int tiles_x, tiles_y;
    int row0;                     // points-of-keys over a band of pixel rows: image row of the band's first row (0 otherwise)
    int addr32;                   // the march's 32-bit voxel index applies (rc_march_addr)
};

__device__ __forceinline__ const uint32_t* rc_vox_addr(const DfRayArgs& a, int x, int y, int z)
{
    x = min(max(x, 0), a.X - 1);
    y = min(max(y, 0), a.Y - 1);
    int zl = min(max(z - a.z_store0, 0), a.z_store_n - 1);
    return a.vol + ((size_t)x + (size_t)y * a.X + (size_t)zl * a.X * a.Y);
}
// The march's form of the same address: it is most of the march's instructions (one per step and ray), so the clamps are one
// v_med3_i32 each and the voxel index is made in 32 bits with 24-bit multiplies (full rate; v_mul_lo_u32 and the 64-bit
// multiply-adds of the generic form are quarter rate).  Valid while z_store_n * Y <= 2^24, X < 2^24 and the slab holds <= 2^30
// voxels (df_raycast_setup decides; larger slabs take the generic form).
__device__ __forceinline__ int rc_med3(int v, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "v"(lo), "v"(hi));
    return r;
}
template <bool ADDR32>
__device__ __forceinline__ const uint32_t* rc_march_addr(const DfRayArgs& a, int x, int y, int z)
{
    if (!ADDR32) return rc_vox_addr(a, x, y, z);
    const unsigned xi = (unsigned)rc_med3(x, 0, a.X - 1), yi = (unsigned)rc_med3(y, 0, a.Y - 1), zi = (unsigned)rc_med3(z - a.z_store0, 0, a.z_store_n - 1);
    unsigned row, idx;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(zi), "s"(a.Y), "v"(yi));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(idx) : "v"(row), "s"(a.X), "v"(xi));
    return reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a.vol) + (idx << 2));      // byte offset < 2^32
}
__device__ __forceinline__ float rc_vox(const DfRayArgs& a, int x, int y, int z)
{   // device.hpp:17-18 ; clamped to the stored range (the reference reads unchecked, tsdf_volume.cu:262-270)
    x = min(max(x, 0), a.X - 1);
    y = min(max(y, 0), a.Y - 1);
    int zl = min(max(z - a.z_store0, 0), a.z_store_n - 1);
    return h2f_bits(a.vol[(size_t)x + (size_t)y * a.X + (size_t)zl * a.X * a.Y]);
}

// tsdf_volume.cu:220-245
__device__ __forceinline__ float rc_interpolate(const DfRayArgs& a, f3 cf)
{
    const float fx = floorf(cf.x), fy = floorf(cf.y), fz = floorf(cf.z);     // __float2int_rd
    const int gx = (int)fx, gy = (int)fy, gz = (int)fz;
    if (gx < 0 || gx >= a.X - 1 || gy < 0 || gy >= a.Y - 1 || gz < 0 || gz >= a.Z - 1) return qnanf_();
    const float aa = cf.x - (float)gx, b = cf.y - (float)gy, c = cf.z - (float)gz;
    float t = 0.f;
    t += rc_vox(a, gx + 0, gy + 0, gz + 0) * (1 - aa) * (1 - b) * (1 - c);
    t += rc_vox(a, gx + 0, gy + 0, gz + 1) * (1 - aa) * (1 - b) * c;
    t += rc_vox(a, gx + 0, gy + 1, gz + 0) * (1 - aa) * b * (1 - c);
    t += rc_vox(a, gx + 0, gy + 1, gz + 1) * (1 - aa) * b * c;
    t += rc_vox(a, gx + 1, gy + 0, gz + 0) * aa * (1 - b) * (1 - c);
    t += rc_vox(a, gx + 1, gy + 0, gz + 1) * aa * (1 - b) * c;
    t += rc_vox(a, gx + 1, gy + 1, gz + 0) * aa * b * (1 - c);
    t += rc_vox(a, gx + 1, gy + 1, gz + 1) * aa * b * c;
    return t;
}

// tsdf_volume.cu:408-426
__device__ __forceinline__ f3 rc_normal(const DfRayArgs& a, f3 p)
{
    const f3 vsi = mk3(a.vsix, a.vsiy, a.vsiz);
    f3 n;
    const float Fx1 = rc_interpolate(a, mul3(mk3(p.x + a.gdx, p.y, p.z), vsi));
    const float Fx2 = rc_interpolate(a, mul3(mk3(p.x - a.gdx, p.y, p.z), vsi));
    n.x = (Fx1 - Fx2) / a.gdx;
    const float Fy1 = rc_interpolate(a, mul3(mk3(p.x, p.y + a.gdy, p.z), vsi));
    const float Fy2 = rc_interpolate(a, mul3(mk3(p.x, p.y - a.gdy, p.z), vsi));
    n.y = (Fy1 - Fy2) / a.gdy;
    const float Fz1 = rc_interpolate(a, mul3(mk3(p.x, p.y, p.z + a.gdz), vsi));
    const float Fz2 = rc_interpolate(a, mul3(mk3(p.x, p.y, p.z - a.gdz), vsi));
    n.z = (Fz1 - Fz2) / a.gdz;
    return normalized3(n);
}

// ---- the three stages of one ray: march (first event), locate (zero-crossing refinement), shade (normal).
struct DfRayHit { uint32_t key; bool hit; float t_hit; f3 p_curr, p_next, org, dir; };

#ifndef DF_RC_WINDOW
#define DF_RC_WINDOW 4
#endif
// :353-404 minus the refinement: first event on a step this slab owns.
template <bool SLAB, bool ADDR32>
__device__ __forceinline__ DfRayHit rc_march(const DfRayArgs& a, int x, int y)
{
    DfRayHit h;
    h.key = 0xffffffffu; h.hit = false; h.t_hit = 0.f;
    h.org = mk3(a.aff.t[0], a.aff.t[1], a.aff.t[2]);
    // device.hpp:43-48 with z = 1.f
    const f3 rp = mk3(1.f * ((float)x - a.cx) * a.finvx, 1.f * ((float)y - a.cy) * a.finvy, 1.f);
    h.dir = normalized3(mat3_mul(a.aff.R, rp));                                       // :354
    h.p_curr = h.org; h.p_next = h.org;
    const f3 org = h.org, dir = h.dir;

    // intersect, :202-218, box [0, volume_size - voxel_size] (:359)
    float tmin, tmax;
    {
        const f3 box_max = mk3(a.sizex - a.vsx, a.sizey - a.vsy, a.sizez - a.vsz);
        const f3 invR = mk3(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
        const f3 tbot = mul3(invR, sub3(mk3(0.f, 0.f, 0.f), org));
        const f3 ttop = mul3(invR, sub3(box_max, org));
        const f3 tmn = mk3(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
        const f3 tmx = mk3(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));
        tmin = fmaxf(fmaxf(tmn.x, tmn.y), fmaxf(tmn.x, tmn.z));
        tmax = fminf(fminf(tmx.x, tmx.y), fminf(tmx.x, tmx.z));
    }
    tmin = fmaxf(0.f, tmin);                                                          // :364-365
    if (!(tmin < tmax)) return h;                                                     // :366
    tmax -= a.time_step;                                                              // :369
    const f3 vstep = scale3(dir, a.time_step);
    f3 next = add3(org, scale3(dir, tmin));
    // fetch_tsdf, :262-270 (__float2int_rn == rint, round-half-even)
    int zn = (int)rintf(next.z * a.vsiz);
    float tsdf_next = h2f_bits(*rc_march_addr<ADDR32>(a, (int)rintf(next.x * a.vsix), (int)rintf(next.y * a.vsiy), zn));   // :373
    // The march (:374-385) is one dependent nearest-voxel gather per step -- 50 to 170 L2 round trips one after the other, which is
    // most of the kernel's duration.  Sample positions do not depend on sample values, so the gathers of DF_RC_WINDOW consecutive
    // steps are issued together (clamped addresses: valid even past the last step, nothing branches around a load) and the events
    // are then looked for in step order: the same positions (repeated `next += vstep`, `tcurr += time_step`), the same tests, the
    // same first event.  0.087 -> 0.063 ms at 640x480 / 512^3; a window of 8 is no better.
    uint32_t k = 0;
    float tcurr = tmin;
    bool finished = false;
    while (!finished && tcurr < tmax) {
        f3 pn[DF_RC_WINDOW]; int znn[DF_RC_WINDOW]; float tc[DF_RC_WINDOW], tv[DF_RC_WINDOW]; bool live[DF_RC_WINDOW], ownc[DF_RC_WINDOW];
        const uint32_t* addr[DF_RC_WINDOW];
        {
            f3 pc = next; int zc = zn; float t = tcurr;
#pragma unroll
            for (int j = 0; j < DF_RC_WINDOW; ++j) {
                live[j] = t < tmax; tc[j] = t;                                         // :374
                pn[j] = add3(pc, vstep);
                znn[j] = (int)rintf(pn[j].z * a.vsiz);
                ownc[j] = zc >= a.z_own0 && zc < a.z_own1;
                addr[j] = rc_march_addr<ADDR32>(a, (int)rintf(pn[j].x * a.vsix), (int)rintf(pn[j].y * a.vsiy), znn[j]);   // clamped: always a valid address
                pc = pn[j]; zc = znn[j]; t += a.time_step;
            }
            tcurr = t;
        }
        // A slab's march tests only the steps whose current sample lies in a plane it owns: the value of step j's next sample is
        // needed for step j's own test and as step j + 1's current value, nothing else.  A window none of whose values is needed
        // by any lane of the wave (the ray is outside the slab's planes: (N - 1) / N of the march on N slabs) issues no loads.
        bool any_need = !SLAB;                                                         // (the unsharded cast owns every plane)
        if (SLAB) {
#pragma unroll
            for (int j = 0; j < DF_RC_WINDOW; ++j) any_need = any_need || (live[j] && (ownc[j] || (znn[j] >= a.z_own0 && znn[j] < a.z_own1)));
            any_need = __builtin_amdgcn_ballot_w64(any_need) != 0ull;
        }
        if (any_need) {
#pragma unroll
            for (int j = 0; j < DF_RC_WINDOW; ++j) tv[j] = h2f_bits(*addr[j]);         // :380, all in flight together
        } else {
#pragma unroll
            for (int j = 0; j < DF_RC_WINDOW; ++j) tv[j] = 0.f;                        // (never looked at)
        }
        // An event needs a NEGATIVE sample (:381 curr < 0, :384 next < 0).  A window with none -- free space, most of the march -- has
        // no event, and a ray that ends inside it leaves the loop through `tcurr < tmax` above (tcurr is already past the window):
        // the step-by-step search below, ~60 vector and ~180 scalar instructions of nested branches, is skipped.
        bool any_neg = tsdf_next < 0.f;
#pragma unroll
        for (int j = 0; j < DF_RC_WINDOW; ++j) any_neg = any_neg || tv[j] < 0.f;
        float tsdf_curr = tsdf_next;
        f3 curr = next;
        if (any_neg) {
#pragma unroll
        for (int j = 0; j < DF_RC_WINDOW; ++j) {
            if (!finished) {
                if (!live[j]) finished = true;
                else if (ownc[j]) {
                    if (tsdf_curr < 0.f && tv[j] > 0.f) { h.key = (k + j) << 1; finished = true; }   // :381
                    else if (tsdf_curr > 0.f && tv[j] < 0.f) {                        // :384
                        h.key = ((k + j) << 1) | 1u; h.hit = true; h.t_hit = tc[j]; h.p_curr = curr; h.p_next = pn[j];
                        finished = true;
                    }
                }
            }
            tsdf_curr = tv[j]; curr = pn[j];
        }
        }
        next = pn[DF_RC_WINDOW - 1]; zn = znn[DF_RC_WINDOW - 1]; tsdf_next = tv[DF_RC_WINDOW - 1];
        k += DF_RC_WINDOW;
    }
    return h;
}

// :386-391 : the refined ray parameter Ts of a hit and its vertex org + dir * Ts (volume frame).  Reads only the trilinear
// neighbourhoods of curr and next.
__device__ __forceinline__ float rc_locate_ts(const DfRayArgs& a, const DfRayHit& h)
{
    const f3 vsi = mk3(a.vsix, a.vsiy, a.vsiz);
    const float Ft = rc_interpolate(a, mul3(h.p_curr, vsi));                          // :386
    const float Ftdt = rc_interpolate(a, mul3(h.p_next, vsi));                        // :387
    return h.t_hit - (a.time_step * Ft) / (Ftdt - Ft);                                // :389  (may extrapolate far!)
}
__device__ __forceinline__ f3 rc_vertex(f3 org, f3 dir, float Ts) { return add3(org, scale3(dir, Ts)); }          // :390
__device__ __forceinline__ f3 rc_locate(const DfRayArgs& a, const DfRayHit& h) { return rc_vertex(h.org, h.dir, rc_locate_ts(a, h)); }
// the ray of pixel (x, y), :353-354 (what rc_march starts from)
__device__ __forceinline__ void rc_ray_of_pixel(const DfRayArgs& a, int x, int y, f3* org, f3* dir)
{
    *org = mk3(a.aff.t[0], a.aff.t[1], a.aff.t[2]);
    const f3 rp = mk3(1.f * ((float)x - a.cx) * a.finvx, 1.f * ((float)y - a.cy) * a.finvy, 1.f);   // device.hpp:43-48 with z = 1.f
    *dir = normalized3(mat3_mul(a.aff.R, rp));
}

// :392-401 : normal at the vertex, validity test, camera-frame outputs.
__device__ __forceinline__ bool rc_shade(const DfRayArgs& a, f3 vertex, float4* out_p, float4* out_n, uint16_t* out_d)
{
    const f3 normal = rc_normal(a, vertex);
    const float chk = normal.x * normal.y * normal.z;
    if (!(chk == chk)) return false;                                                  // :394 !isnan
    const f3 org = mk3(a.aff.t[0], a.aff.t[1], a.aff.t[2]);
    const f3 n = mat3_mul(a.Rinv, normal);
    const f3 v = mat3_mul(a.Rinv, sub3(vertex, org));
    *out_n = make_float4(n.x, n.y, n.z, 0.f);
    *out_p = make_float4(v.x, v.y, v.z, 0.f);
    const float mm = v.z * 1000;                                                      // :333
    *out_d = (uint16_t)(mm <= 0.f ? 0 : (mm >= 65535.f ? 65535 : (int)mm));
    return true;
}

// pixel of this thread: a wave64 owns an 8x8 tile, a workgroup 16x16; XCD-aware, bijective renumbering:
// hardware block id b runs on XCD b % 8; each XCD gets a contiguous run of logical tiles (row-major).
__device__ __forceinline__ bool rc_pixel(const DfRayArgs& a, int* px, int* py)
{
    const int nwg = a.tiles_x * a.tiles_y;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int ty = wg / a.tiles_x, tx = wg - ty * a.tiles_x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    *px = tx * 16 + (w & 1) * 8 + (lane & 7);
    *py = ty * 16 + (w >> 1) * 8 + (lane >> 3);
    return *px < a.cols && *py < a.rows;
}

#ifdef DF_TRACE_RAYCAST           // per-wave timeline of the fused cast (tools/trace_raycast.py): start, march done, end; steps marched
__device__ unsigned long long g_df_rtrace[8192 * 4];
#define DF_RT(i, v) do { if ((threadIdx.x & 63) == 0 && blockIdx.x * 4 + (threadIdx.x >> 6) < 8192) g_df_rtrace[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (i)] = (v); } while (0)
#else
#define DF_RT(i, v) do { } while (0)
#endif
template <int MODE /* 0 = Points (tsdf_volume.cu:340-405), 1 = Depth (:272-338) */, bool ADDR32>
__global__ __launch_bounds__(256) void df_raycast_kernel(const DfRayArgs a)
{
    int x, y;
    DF_RT(0, wall_clock64());
    if (!rc_pixel(a, &x, &y)) return;
    const float qn = qnanf_();
    const DfRayHit h = rc_march<false, ADDR32>(a, x, y);                                       // (launched on whole volumes; a slab's keys come from the march kernel)
    DF_RT(1, wall_clock64());
    float4 out_p = make_float4(qn, qn, qn, qn), out_n = make_float4(qn, qn, qn, qn);   // :351
    uint16_t out_d = 0;                                                               // :283
    if (h.hit) rc_shade(a, rc_locate(a, h), &out_p, &out_n, &out_d);                   // refinement after the loop: wave reconverged
    *reinterpret_cast<float4*>((char*)a.nrm + (size_t)y * a.npitch + 16 * (size_t)x) = out_n;
    if (MODE == 0) *reinterpret_cast<float4*>((char*)a.pts + (size_t)y * a.ppitch + 16 * (size_t)x) = out_p;
    else *reinterpret_cast<uint16_t*>((char*)a.depth + (size_t)y * a.dpitch + 2 * (size_t)x) = out_d;
    if (a.keys) a.keys[(size_t)y * a.cols + x] = h.key;
    DF_RT(2, wall_clock64());
}

// ---- sharded cast, stage 1: first event on owned steps, and for a hit its refined ray parameter Ts (:389).
// The merge key (DF_RC_KEY_* in dfusion.h): [ 0 | step k : 23 | hit : 1 | rank : 7 | Ts bits : 32 ].  A per-pixel MIN over ranks
// (ncclMin on int64; the top bit stays 0) picks the first event along the ray, names its owner AND delivers Ts -- the vertex is
// org + dir * Ts, which every rank recomputes from the pixel exactly as the unsharded cast does (:390), so no vertex image has to
// cross GPUs.  No event at all: 0x7fffffffffffffff, larger than every event.
template <bool ADDR32>
__global__ __launch_bounds__(256) void df_raycast_march_kernel(const DfRayArgs a, unsigned long long* __restrict__ keys64, unsigned int rank_tag)
{
    int x, y;
    if (!rc_pixel(a, &x, &y)) return;
    const DfRayHit h = rc_march<true, ADDR32>(a, x, y);
    unsigned long long k64 = 0x7fffffffffffffffull;
    if (h.key != 0xffffffffu) {
        const float ts = h.hit ? rc_locate_ts(a, h) : 0.f;
        k64 = ((unsigned long long)((h.key << 7) | rank_tag) << 32) | (unsigned long long)__float_as_uint(ts);
    }
    keys64[(size_t)y * a.cols + x] = k64;
}

// ---- sharded cast, stage 2: after the per-pixel MIN merge of the keys, the slab that owns the vertex' plane computes the
// normal.  Pixels this slab does not resolve get all-zero bits (the host sums integer views across ranks); resolved misses get
// the reference's NaN fill.
__global__ __launch_bounds__(256) void df_raycast_shade_kernel(const DfRayArgs a, const unsigned long long* __restrict__ merged_keys)
{
    int x, y;
    if (!rc_pixel(a, &x, &y)) return;
    const unsigned long long k64 = merged_keys[(size_t)y * a.cols + x];
    const float qn = qnanf_();
    float4 out_p = make_float4(0.f, 0.f, 0.f, 0.f), out_n = out_p;
    if (k64 != 0x7fffffffffffffffull && ((k64 >> 39) & 1ull)) {
        f3 org, dir;
        rc_ray_of_pixel(a, x, y, &org, &dir);
        const f3 v = rc_vertex(org, dir, __uint_as_float((unsigned int)k64));
        // owner of the vertex: the slab holding its nearest plane (clamped into the volume; NaN -> plane 0)
        float zf = rintf(v.z * a.vsiz);
        int pz = (zf == zf) ? (int)fminf(fmaxf(zf, 0.f), (float)(a.Z - 1)) : 0;
        if (pz >= a.z_own0 && pz < a.z_own1) {
            out_p = make_float4(qn, qn, qn, qn); out_n = out_p;
            uint16_t d;
            rc_shade(a, v, &out_p, &out_n, &d);
        }
    } else if (a.z_own0 == 0) {                       // misses / back-face breaks: filled once, by the slab owning plane 0
        out_p = make_float4(qn, qn, qn, qn); out_n = out_p;
    }
    *reinterpret_cast<float4*>((char*)a.nrm + (size_t)y * a.npitch + 16 * (size_t)x) = out_n;
    if (a.pts) *reinterpret_cast<float4*>((char*)a.pts + (size_t)y * a.ppitch + 16 * (size_t)x) = out_p;
}

// ---- sharded cast, stage 3 (on the rank that wants the image): the POINTS need no exchange.  The merged key holds the winner's Ts,
// the vertex is origin + direction * Ts from the pixel (:390), the camera-frame point Rinv * (vertex - origin) (:396) -- what
// rc_shade writes -- and whether the hit stands is written in the summed normals: a valid normal has 0 in its 4th component, the
// NaN fill has a NaN there (:394 decides on the volume-frame normal, which only the vertex' owner can compute).  So only the normals
// cross GPUs (4.9 MB instead of 9.8).
__global__ __launch_bounds__(256) void df_raycast_points_of_keys_kernel(const DfRayArgs a, const unsigned long long* __restrict__ merged_keys)
{
    int x, y;
    if (!rc_pixel(a, &x, &y)) return;                                      // (y: row inside the band; a.rows = rows of the band)
    const int yi = y + a.row0;                                             // image row: the keys are the whole image's, normals / points the band's
    const unsigned long long k64 = merged_keys[(size_t)yi * a.cols + x];
    const float qn = qnanf_();
    float4 out_p = make_float4(qn, qn, qn, qn);
    if (k64 != 0x7fffffffffffffffull && ((k64 >> 39) & 1ull)) {
        const float nw = reinterpret_cast<const float4*>((const char*)a.nrm + (size_t)y * a.npitch)[x].w;
        if (nw == nw) {
            f3 org, dir;
            rc_ray_of_pixel(a, x, yi, &org, &dir);
            const f3 vertex = rc_vertex(org, dir, __uint_as_float((unsigned int)k64));
            const f3 v = mat3_mul(a.Rinv, sub3(vertex, org));                              // as rc_shade
            out_p = make_float4(v.x, v.y, v.z, 0.f);
        }
    }
    *reinterpret_cast<float4*>((char*)a.pts + (size_t)y * a.ppitch + 16 * (size_t)x) = out_p;
}

static int df_raycast_setup(DfRayArgs& a, const DfVolume& v, const DfSlab* slab, const float cam2vol[12], const float Rinv[9],
                            const float reproj[4], int cols, int rows, float step_factor, float delta_factor)
{
    if (!cam2vol || !Rinv || !reproj || cols <= 0 || rows <= 0 || !df_volume_valid(v)) return DF_E_INVALID;
    DfSlab s = df_slab_or_full(v, slab);
    if (!df_slab_valid(v, s)) return DF_E_INVALID;
    a.vol = (const uint32_t*)v.data; a.X = v.dims[0]; a.Y = v.dims[1]; a.Z = v.dims[2];
    a.z_store0 = s.z_store0; a.z_store_n = s.z_store_n; a.z_own0 = s.z_own0; a.z_own1 = s.z_own0 + s.z_own_n;
    a.row0 = 0;
    a.addr32 = (unsigned long long)s.z_store_n * (unsigned long long)a.Y <= (1ull << 24) && a.X < (1 << 24) &&
               (unsigned long long)s.z_store_n * (unsigned long long)a.Y * (unsigned long long)a.X <= (1ull << 30);
    a.vsx = v.voxel_size[0]; a.vsy = v.voxel_size[1]; a.vsz = v.voxel_size[2];
    a.sizex = a.vsx * (float)a.X; a.sizey = a.vsy * (float)a.Y; a.sizez = a.vsz * (float)a.Z;      // tsdf_volume.cu:464
    a.time_step = v.trunc_dist * step_factor;                                                      // :465
    a.gdx = a.vsx * delta_factor; a.gdy = a.vsy * delta_factor; a.gdz = a.vsz * delta_factor;      // :466
    a.vsix = 1.f / a.vsx; a.vsiy = 1.f / a.vsy; a.vsiz = 1.f / a.vsz;                              // :467
    a.aff = df_aff(cam2vol);
    memcpy(a.Rinv, Rinv, sizeof(a.Rinv));
    a.finvx = reproj[0]; a.finvy = reproj[1]; a.cx = reproj[2]; a.cy = reproj[3];
    a.cols = cols; a.rows = rows;
    a.tiles_x = (cols + 15) / 16; a.tiles_y = (rows + 15) / 16;
    a.pts = nullptr; a.nrm = nullptr; a.depth = nullptr; a.keys = nullptr; a.ppitch = a.npitch = a.dpitch = 0;
    return DF_OK;
}

extern "C" int dfusion_raycast_points(DfVolume v, const DfSlab* slab, const float cam2vol[12], const float Rinv[9],
                                      const float reproj[4], float* points, size_t ppitch, float* normals, size_t npitch,
                                      int cols, int rows, float step_factor, float delta_factor, uint32_t* keys, dfStream stream)
{
    if (!points || !normals) return DF_E_INVALID;
    DfRayArgs a;
    int rc = df_raycast_setup(a, v, slab, cam2vol, Rinv, reproj, cols, rows, step_factor, delta_factor);
    if (rc) return rc;
    a.pts = points; a.ppitch = ppitch; a.nrm = normals; a.npitch = npitch; a.keys = keys;
    if (a.addr32) hipLaunchKernelGGL((df_raycast_kernel<0, true>), dim3(a.tiles_x * a.tiles_y), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((df_raycast_kernel<0, false>), dim3(a.tiles_x * a.tiles_y), dim3(256), 0, (hipStream_t)stream, a);
    DF_LAUNCH_CHECK();
#ifdef DF_TRACE_RAYCAST
    if (getenv("DF_TRACE_RAYCAST_FILE")) {
        DF_HIP(hipStreamSynchronize((hipStream_t)stream));
        static unsigned long long h[8192 * 4];
        DF_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_df_rtrace), sizeof(h)));
        FILE* f = fopen(getenv("DF_TRACE_RAYCAST_FILE"), "wb");
        if (f) { fwrite(h, 8, 8192 * 4, f); fclose(f); }
    }
#endif
    return DF_OK;
}

extern "C" int dfusion_raycast_depth(DfVolume v, const DfSlab* slab, const float cam2vol[12], const float Rinv[9],
                                     const float reproj[4], uint16_t* depth, size_t dpitch, float* normals, size_t npitch,
                                     int cols, int rows, float step_factor, float delta_factor, dfStream stream)
{
    if (!depth || !normals) return DF_E_INVALID;
    DfRayArgs a;
    int rc = df_raycast_setup(a, v, slab, cam2vol, Rinv, reproj, cols, rows, step_factor, delta_factor);
    if (rc) return rc;
    a.depth = depth; a.dpitch = dpitch; a.nrm = normals; a.npitch = npitch;
    if (a.addr32) hipLaunchKernelGGL((df_raycast_kernel<1, true>), dim3(a.tiles_x * a.tiles_y), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((df_raycast_kernel<1, false>), dim3(a.tiles_x * a.tiles_y), dim3(256), 0, (hipStream_t)stream, a);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ---- sharded (Z-slab) cast in two stages; no reference counterpart (the reference is single-GPU).
extern "C" int dfusion_raycast_march(DfVolume v, const DfSlab* slab, const float cam2vol[12], const float reproj[4], int cols,
                                     int rows, float step_factor, unsigned int rank_tag, unsigned long long* keys, dfStream stream)
{
    if (!keys || rank_tag > DF_RC_KEY_MAX_RANK) return DF_E_INVALID;
    DfRayArgs a;
    const float ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int rc = df_raycast_setup(a, v, slab, cam2vol, ident, reproj, cols, rows, step_factor, 0.5f);
    if (rc) return rc;
    // the merge key carries the step index in 23 bits: a ray's march is tmax - tmin <= the volume's diagonal long, in steps of
    // time_step -- refuse a step so small (or a volume so deep) that the index could reach 2^23 and wrap the ordering
    {
        const double diag = sqrt((double)a.sizex * a.sizex + (double)a.sizey * a.sizey + (double)a.sizez * a.sizez);
        if (!(a.time_step > 0.f) || !(diag / (double)a.time_step < 8388608.0 - 4.0)) return DF_E_INVALID;
    }
    if (a.addr32) hipLaunchKernelGGL(df_raycast_march_kernel<true>, dim3(a.tiles_x * a.tiles_y), dim3(256), 0, (hipStream_t)stream, a, keys, rank_tag);
    else hipLaunchKernelGGL(df_raycast_march_kernel<false>, dim3(a.tiles_x * a.tiles_y), dim3(256), 0, (hipStream_t)stream, a, keys, rank_tag);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

extern "C" int dfusion_raycast_shade(DfVolume v, const DfSlab* slab, const float cam2vol[12], const float Rinv[9],
                                     const float reproj[4], const unsigned long long* merged_keys, float* points,
                                     size_t ppitch, float* normals, size_t npitch, int cols, int rows, float delta_factor,
                                     dfStream stream)
{
    if (!merged_keys || !normals) return DF_E_INVALID;                  // points nullable: see dfusion_raycast_points_of_keys
    DfRayArgs a;
    int rc = df_raycast_setup(a, v, slab, cam2vol, Rinv, reproj, cols, rows, 0.75f, delta_factor);
    if (rc) return rc;
    a.pts = points; a.ppitch = ppitch; a.nrm = normals; a.npitch = npitch;
    hipLaunchKernelGGL(df_raycast_shade_kernel, dim3(a.tiles_x * a.tiles_y), dim3(256), 0, (hipStream_t)stream, a, merged_keys);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

extern "C" int dfusion_raycast_points_of_keys(const float cam2vol[12], const float Rinv[9], const float reproj[4],
                                              const unsigned long long* merged_keys, const float* normals, size_t npitch, float* points,
                                              size_t ppitch, int cols, int rows, dfStream stream)
{
    return dfusion_raycast_points_of_keys_rows(cam2vol, Rinv, reproj, merged_keys, normals, npitch, points, ppitch, cols, rows, 0, rows, stream);
}

extern "C" int dfusion_raycast_points_of_keys_rows(const float cam2vol[12], const float Rinv[9], const float reproj[4],
                                                   const unsigned long long* merged_keys, const float* normals, size_t npitch, float* points,
                                                   size_t ppitch, int cols, int image_rows, int row0, int rows, dfStream stream)
{
    if (!cam2vol || !Rinv || !reproj || !merged_keys || !normals || !points || cols <= 0 || image_rows <= 0) return DF_E_INVALID;
    if (row0 < 0 || rows < 0 || row0 + rows > image_rows) return DF_E_INVALID;
    if (rows == 0) return DF_OK;
    DfRayArgs a;
    memset(&a, 0, sizeof(a));
    a.aff = df_aff(cam2vol);
    memcpy(a.Rinv, Rinv, sizeof(a.Rinv));
    a.finvx = reproj[0]; a.finvy = reproj[1]; a.cx = reproj[2]; a.cy = reproj[3];
    a.cols = cols; a.rows = rows;
    a.tiles_x = (cols + 15) / 16; a.tiles_y = (rows + 15) / 16;
    a.row0 = row0;
    a.pts = points; a.ppitch = ppitch; a.nrm = const_cast<float*>(normals); a.npitch = npitch;
    hipLaunchKernelGGL(df_raycast_points_of_keys_kernel, dim3(a.tiles_x * a.tiles_y), dim3(256), 0, (hipStream_t)stream, a, merged_keys);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ====================================================================================== cloud / normal extraction
// extract_kernel (FullScan6) + extract_normals_kernel, /root/reference/kfusion/src/cuda/tsdf_volume.cu:
// 511-710, 714-795 -- called every frame by KinFu::dynamicfusion (kinfu.cpp:398-399) and once to seed the warp field.
//
// The scan is a pure HBM stream (4 B/voxel read, a few MB of points written).  A lane owns four x-adjacent voxels (one
// global_load_dwordx4 per plane; 1 KiB contiguous per wave) and walks a Z chunk carrying plane z+1 in registers as the
// next iteration's own plane; the +x neighbour of its 4th voxel comes from the next lane by DPP/shuffle, the +y row is a
// second 16-byte load that hits L1/L2 (the row is some other wave's own).  Crossings are compacted per wave with
// __ballot / popcount prefixes (12 slots: 4 voxels x 3 axes) and ONE atomicAdd per wave per plane -- no LDS staging, no
// __device__ globals (the reference keeps global_count / output_count / blocks_done in globals, :506-508).
struct DfExtractArgs {
    const uint32_t* vol; int X, Y, Z;
    int z_store0, z_own0, z_end;       // scans planes [z_own0, z_end), z_end <= Z-1
    int zc;
    float vsx, vsy, vsz;
    DfAff aff;
    float4* out; unsigned long long capacity; unsigned long long* count;
};

// :548 W != 0 && F != 1.f  (the only half equal to 1.f is 0x3c00: an integer test, no conversion in the streaming loop)
__device__ __forceinline__ bool ex_valid(uint32_t v) { return (v >> 16) != 0u && (v & 0xffffu) != 0x3c00u; }
__device__ __forceinline__ bool ex_cross(uint32_t a, uint32_t b)
{
    const float F = h2f_bits(a), Fn = h2f_bits(b);
    return ex_valid(a) && ex_valid(b) && ((F > 0.f && Fn < 0.f) || (F < 0.f && Fn > 0.f));                 // :559
}

// One launch, two phases per workgroup, no contended atomics.  Measured on MI355X (512^3, 311 k crossings): a pass with
// one atomicAdd per wave-item that has crossings is held at 1.9 TB/s -- not by the traversal (walking Z per lane or
// address order: same) but by ~30 k returning atomics on ONE counter (~11 ns each); the bare scan streams at 4.4 TB/s.  So:
//   phase 1 (scan)   every workgroup takes 4 KiB-contiguous pieces of the volume in ADDRESS order (grid-stride: all
//                    resident workgroups advance through the planes together, a pure HBM stream).  A wave-item (256
//                    x-adjacent voxels) holding any valid voxel (W != 0 && F != 1, :548 -- only the truncation shell
//                    qualifies, 6.6 % of the items) goes on the workgroup's list in LDS.
//   phase 2 (refine) the workgroup walks its own list, one item per wave and round: +x neighbour by shuffle, +y row and
//                    +z plane by 16-byte loads (L2 / Infinity Cache hits), 12 crossing tests per lane, __ballot /
//                    popcount-prefix compaction into an LDS buffer flushed with ONE global atomicAdd per ~1000 points;
//                    other workgroups are still streaming meanwhile, which hides the refine latency.
#ifndef DF_EX_GRIDSTRIDE
#define DF_EX_GRIDSTRIDE 0
#endif
#define DF_EX_CAP 1024          // points buffered per workgroup (16 KiB); a round that cannot fit goes straight to global

template <int U>      // 16-byte loads in flight per lane
__global__ __launch_bounds__(256) void df_extract_kernel(const DfExtractArgs a, unsigned int items_per_block)
{
    extern __shared__ unsigned int s_list[];                            // [items_per_block]
    __shared__ float4 s_buf[DF_EX_CAP];
    __shared__ unsigned int s_tot[4];
    __shared__ unsigned int s_n, s_fill;
    __shared__ unsigned long long s_base;
    if (threadIdx.x == 0) { s_n = 0u; s_fill = 0u; }
    __syncthreads();
    const size_t plane = (size_t)a.X * a.Y;
    const size_t n4 = plane * (size_t)(a.z_end - a.z_own0) / 4;         // lane-items: 4 x-adjacent voxels each
    const uint32_t* base = a.vol + (size_t)(a.z_own0 - a.z_store0) * plane;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    {   // ---- phase 1: a workgroup streams CONTIGUOUS runs of 256 U lane-items (16 KiB at U = 4), U non-temporal 16-byte loads in flight
        // per lane -- the form in which a plain copy reaches the part's rate (tools/copy_probe.hip: 5.7-6.3 TB/s against 4.4-5.6 for a
        // grid-stride loop whose loads are a whole grid apart; round 3 scanned that way, 3.3 TB/s).  DF_EX_GRIDSTRIDE=1 restores it for A/B.
        typedef unsigned int df_ex_u4 __attribute__((ext_vector_type(4)));
        const df_ex_u4* base4 = reinterpret_cast<const df_ex_u4*>(base);
#if DF_EX_GRIDSTRIDE
        const size_t stride = (size_t)gridDim.x * 256;
        const size_t first = (size_t)blockIdx.x * 256 + (threadIdx.x & ~63);
        for (size_t w0 = first; w0 < n4; w0 += stride * U) {
            uint4 own[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t i = w0 + u * stride + lane;
                const df_ex_u4 v = i < n4 ? base4[i] : df_ex_u4{0u, 0u, 0u, 0u};
                own[u] = make_uint4(v.x, v.y, v.z, v.w);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ownv = ex_valid(own[u].x) | ex_valid(own[u].y) | ex_valid(own[u].z) | ex_valid(own[u].w);
                if (__any(ownv) && lane == 0) s_list[atomicAdd(&s_n, 1u)] = (unsigned int)((w0 + u * stride) >> 6);
            }
        }
#else
        const size_t chunk = (size_t)256 * U;
        for (size_t c0 = (size_t)blockIdx.x * chunk; c0 < n4; c0 += (size_t)gridDim.x * chunk) {
            uint4 own[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t i = c0 + (size_t)u * 256 + threadIdx.x;
                const df_ex_u4 v = i < n4 ? __builtin_nontemporal_load(base4 + i) : df_ex_u4{0u, 0u, 0u, 0u};
                own[u] = make_uint4(v.x, v.y, v.z, v.w);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ownv = ex_valid(own[u].x) | ex_valid(own[u].y) | ex_valid(own[u].z) | ex_valid(own[u].w);
                if (__any(ownv) && lane == 0) s_list[atomicAdd(&s_n, 1u)] = (unsigned int)((c0 + (size_t)u * 256 + (size_t)wave * 64) >> 6);
            }
        }
#endif
    }
    __syncthreads();

    // cooperative flush of the LDS buffer: one global atomic for the whole workgroup
    auto flush = [&]() {
        const unsigned int fill = s_fill;                                    // uniform (read after a barrier)
        if (fill) {
            if (threadIdx.x == 0) s_base = atomicAdd(a.count, (unsigned long long)fill);
            __syncthreads();
            const unsigned long long b = s_base;
            for (unsigned int t = threadIdx.x; t < fill; t += 256)
                if (b + t < a.capacity) a.out[b + t] = s_buf[t];
            __syncthreads();
            if (threadIdx.x == 0) s_fill = 0u;
            __syncthreads();
        }
    };

    // ---- phase 2
    const unsigned int n_items = s_n;
    for (unsigned int e0 = 0; e0 < n_items; e0 += 4) {                      // block-uniform: one listed item per wave and round
        const unsigned int e = e0 + wave;
        const size_t i = e < n_items ? ((size_t)s_list[e] << 6) + lane : n4;
        const bool act = i < n4;
        const size_t v0 = act ? 4 * i : 0;                                // linear voxel index inside the scanned range
        const int zr = (int)(v0 / plane);
        const int rem = (int)(v0 - (size_t)zr * plane);
        const int y = rem / a.X, x0 = rem - y * a.X, z = a.z_own0 + zr;
        const uint32_t* pz = base + v0;
        const bool has_xn = act && x0 + 4 < a.X, has_yn = act && y + 1 < a.Y;
        const uint4 cur = act ? *reinterpret_cast<const uint4*>(pz) : zero4;
        const uint4 nz = act ? *reinterpret_cast<const uint4*>(pz + plane) : zero4;       // plane z+1 exists: z < z_end <= Z-1
        const uint4 ny = has_yn ? *reinterpret_cast<const uint4*>(pz + a.X) : zero4;      // row y+1
        uint32_t xn = __shfl_down(cur.x, 1, 64);                                          // lane+1 holds x0+4.. (same row iff has_xn)
        if (lane == 63 && has_xn) xn = pz[4];
        if (!has_xn) xn = 0u;
        const uint32_t c[4] = {cur.x, cur.y, cur.z, cur.w};
        const uint32_t nx[4] = {cur.y, cur.z, cur.w, xn};
        const uint32_t nyv[4] = {ny.x, ny.y, ny.z, ny.w};
        const uint32_t nzv[4] = {nz.x, nz.y, nz.z, nz.w};
        unsigned flags = 0;                                                                // bit 3*k + axis
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (ex_cross(c[k], nx[k])) flags |= 1u << (3 * k);
            if (ex_cross(c[k], nyv[k])) flags |= 2u << (3 * k);
            if (ex_cross(c[k], nzv[k])) flags |= 4u << (3 * k);
        }
        unsigned total = 0;
#pragma unroll
        for (int s = 0; s < 12; ++s) total += (unsigned)__popcll(__ballot((flags >> s) & 1u));
        if (lane == 0) s_tot[wave] = total;
        __syncthreads();
        const unsigned int t0 = s_tot[0], t1 = s_tot[1], t2 = s_tot[2], t3 = s_tot[3];
        const unsigned int sum = t0 + t1 + t2 + t3;
        if (s_fill + sum > DF_EX_CAP) flush();                                             // block-uniform condition
        const bool direct = sum > DF_EX_CAP;                                               // pathological round: bypass the buffer
        unsigned long long off = s_fill + (wave > 0 ? t0 : 0u) + (wave > 1 ? t1 : 0u) + (wave > 2 ? t2 : 0u);
        if (direct && total) {
            unsigned long long g = 0;
            if (lane == 0) g = atomicAdd(a.count, (unsigned long long)total);
            off = __shfl(g, 0, 64);
        }
        if (total) {
            const unsigned long long lt = lane_mask_lt();
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                const unsigned long long bal = __ballot((flags >> s) & 1u);
                if ((flags >> s) & 1u) {
                    const int k = s / 3, axis = s - 3 * k;
                    const float F = fabsf(h2f_bits(c[k]));
                    const float Fn = fabsf(h2f_bits(axis == 0 ? nx[k] : axis == 1 ? nyv[k] : nzv[k]));
                    // voxel-CORNER convention of the extractor, :549-550,566
                    f3 V = mk3(((float)(x0 + k) + 0.5f) * a.vsx, ((float)y + 0.5f) * a.vsy, ((float)z + 0.5f) * a.vsz);
                    const float d_inv = 1.f / (F + Fn);                                    // :567
                    if (axis == 0) { const float Vn = V.x + a.vsx; V.x = (V.x * Fn + Vn * F) * d_inv; }
                    if (axis == 1) { const float Vn = V.y + a.vsy; V.y = (V.y * Fn + Vn * F) * d_inv; }
                    if (axis == 2) { const float Vn = V.z + a.vsz; V.z = (V.z * Fn + Vn * F) * d_inv; }
                    const f3 q = aff_mul(a.aff, V);                                        // :570
                    const unsigned long long o = off + (unsigned long long)__popcll(bal & lt);
                    if (direct) { if (o < a.capacity) a.out[o] = make_float4(q.x, q.y, q.z, 0.f); }
                    else s_buf[o] = make_float4(q.x, q.y, q.z, 0.f);
                }
                off += (unsigned long long)__popcll(bal);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0 && !direct) s_fill += sum;
        __syncthreads();
    }
    flush();
}

// extract_normals_kernel, :714-795 (the reference launches it with a (32,8) block but 1-D indexing: 8x redundant; fixed)
__global__ __launch_bounds__(256) void df_extract_normals_kernel(const DfRayArgs a, const float4* __restrict__ points,
                                                                 unsigned long long n, float4* __restrict__ out)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float qn = qnanf_();
    f3 nrm = mk3(qn, qn, qn);
    const float4 pp = points[i];
    const f3 t = mk3(a.aff.t[0], a.aff.t[1], a.aff.t[2]);
    const f3 point = mat3_mul(a.Rinv, sub3(mk3(pp.x, pp.y, pp.z), t));                          // :749
    const int gx = (int)rintf(point.x * a.vsix), gy = (int)rintf(point.y * a.vsiy), gz = (int)rintf(point.z * a.vsiz);
    if (gx > 1 && gy > 1 && gz > 1 && gx < a.X - 2 && gy < a.Y - 2 && gz < a.Z - 2) {            // :752
        const f3 vsi = mk3(a.vsix, a.vsiy, a.vsiz);
        f3 g;
        g.x = (rc_interpolate(a, mul3(mk3(point.x + a.gdx, point.y, point.z), vsi)) -
               rc_interpolate(a, mul3(mk3(point.x - a.gdx, point.y, point.z), vsi))) / a.gdx;
        g.y = (rc_interpolate(a, mul3(mk3(point.x, point.y + a.gdy, point.z), vsi)) -
               rc_interpolate(a, mul3(mk3(point.x, point.y - a.gdy, point.z), vsi))) / a.gdy;
        g.z = (rc_interpolate(a, mul3(mk3(point.x, point.y, point.z + a.gdz), vsi)) -
               rc_interpolate(a, mul3(mk3(point.x, point.y, point.z - a.gdz), vsi))) / a.gdz;
        nrm = normalized3(mat3_mul(a.aff.R, g));                                                  // :789
    }
    out[i] = make_float4(nrm.x, nrm.y, nrm.z, 0.f);
}

extern "C" int dfusion_extract_cloud(DfVolume v, const DfSlab* slab, const float aff[12], float* points, unsigned long long capacity,
                                     unsigned long long* count, dfStream stream)
{
    if (!aff || !points || !count || !df_volume_valid(v)) return DF_E_INVALID;
    DfSlab s = df_slab_or_full(v, slab);
    if (!df_slab_valid(v, s)) return DF_E_INVALID;
    DfExtractArgs a;
    a.vol = (const uint32_t*)v.data; a.X = v.dims[0]; a.Y = v.dims[1]; a.Z = v.dims[2];
    a.z_store0 = s.z_store0; a.z_own0 = s.z_own0;
    a.z_end = min(s.z_own0 + s.z_own_n, v.dims[2] - 1);                                          // :538 z < dims.z - 1
    if (a.z_end > s.z_store0 + s.z_store_n - 1) return DF_E_INVALID;                              // needs plane z_end as a +z halo
    if (a.z_end <= a.z_own0) return DF_OK;
    a.vsx = v.voxel_size[0]; a.vsy = v.voxel_size[1]; a.vsz = v.voxel_size[2];
    a.aff = df_aff(aff);
    a.out = (float4*)points; a.capacity = capacity; a.count = count;
    a.zc = 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t n4 = (size_t)a.X * a.Y * (size_t)(a.z_end - a.z_own0) / 4;
    const size_t n_items = (n4 + 63) / 64;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;                 // 32 workgroups per CU (measured best of 8..64)
    const int U = 4;
    // wave-items a scan workgroup can see: 4 waves x U items per trip x trips
    const size_t trips = (n4 + blocks * 256 * U - 1) / (blocks * 256 * U);
    const unsigned int ipb = (unsigned int)(trips * U * 4);
    (void)n_items;
    const size_t lds = (size_t)ipb * sizeof(unsigned int);
    if (lds > 40 * 1024) return DF_E_INVALID;
    hipLaunchKernelGGL(df_extract_kernel<4>, dim3((unsigned)blocks), dim3(256), lds, st, a, ipb);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

extern "C" int dfusion_extract_normals(DfVolume v, const DfSlab* slab, const float aff[12], const float Rinv[9], const float* points,
                                       unsigned long long n, float gradient_delta_factor, float* normals, dfStream stream)
{
    if (!points || !normals) return DF_E_INVALID;
    if (n == 0) return DF_OK;
    DfRayArgs a;
    const float reproj[4] = {1.f, 1.f, 0.f, 0.f};
    int rc = df_raycast_setup(a, v, slab, aff, Rinv, reproj, 1, 1, 0.75f, gradient_delta_factor);
    if (rc) return rc;
    hipLaunchKernelGGL(df_extract_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                       (const float4*)points, n, (float4*)normals);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ---- the direct row-band merge's local sum (include/dfusion.h dfusion_raycast_sum_pieces): out = the integer sum of n_pieces pieces of n words
__global__ __launch_bounds__(256) void df_sum_pieces_kernel(const uint32_t* __restrict__ pieces, int n_pieces, unsigned long long n, uint32_t* __restrict__ out)
{
    for (unsigned long long i = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (unsigned long long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            uint4 a = *reinterpret_cast<const uint4*>(pieces + i);
            for (int p = 1; p < n_pieces; ++p) {
                const uint4 b = *reinterpret_cast<const uint4*>(pieces + (unsigned long long)p * n + i);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            *reinterpret_cast<uint4*>(out + i) = a;
        } else {
            for (unsigned long long j = i; j < n; ++j) {
                uint32_t a = pieces[j];
                for (int p = 1; p < n_pieces; ++p) a += pieces[(unsigned long long)p * n + j];
                out[j] = a;
            }
        }
    }
}
extern "C" int dfusion_raycast_sum_pieces(const uint32_t* pieces, int n_pieces, unsigned long long n_words, uint32_t* out, dfStream stream)
{
    if (!pieces || !out || n_pieces < 1 || (n_words & 3ull) || ((size_t)pieces & 15) || ((size_t)out & 15)) return DF_E_INVALID;   // (float4 pixels: whole 16-byte words)
    if (n_words == 0) return DF_OK;
    const unsigned long long want = (n_words / 4 + 255) / 256;
    hipLaunchKernelGGL(df_sum_pieces_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, pieces, n_pieces, n_words, out);
    DF_LAUNCH_CHECK();
    return DF_OK;
}

// ---- the direct form of the sharded cast's FIRST collective (include/dfusion.h dfusion_raycast_min_pieces): out = the per-key minimum of n_pieces
// pieces of n keys (non-negative int64 merge keys: the unsigned order is the signed one)
__global__ __launch_bounds__(256) void df_min_pieces_kernel(const unsigned long long* __restrict__ pieces, int n_pieces, unsigned long long n, unsigned long long* __restrict__ out)
{
    for (unsigned long long i = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 2; i < n; i += (unsigned long long)gridDim.x * 512) {
        if (i + 2 <= n) {
            ulonglong2 a = *reinterpret_cast<const ulonglong2*>(pieces + i);
            for (int p = 1; p < n_pieces; ++p) {
                const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(pieces + (unsigned long long)p * n + i);
                a.x = b.x < a.x ? b.x : a.x; a.y = b.y < a.y ? b.y : a.y;
            }
            *reinterpret_cast<ulonglong2*>(out + i) = a;
        } else {
            unsigned long long a = pieces[i];
            for (int p = 1; p < n_pieces; ++p) { const unsigned long long b = pieces[(unsigned long long)p * n + i]; a = b < a ? b : a; }
            out[i] = a;
        }
    }
}
extern "C" int dfusion_raycast_min_pieces(const unsigned long long* pieces, int n_pieces, unsigned long long n_keys, unsigned long long* out, dfStream stream)
{
    if (!pieces || !out || n_pieces < 1 || ((size_t)pieces & 15) || ((size_t)out & 15) || (n_pieces > 1 && (n_keys & 1ull))) return DF_E_INVALID;   // (16-byte accesses: pieces start on even keys)
    if (n_keys == 0) return DF_OK;
    const unsigned long long want = ((n_keys + 1) / 2 + 255) / 256;
    hipLaunchKernelGGL(df_min_pieces_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream, pieces, n_pieces, n_keys, out);
    DF_LAUNCH_CHECK();
    return DF_OK;
}
