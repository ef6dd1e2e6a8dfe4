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
//   * Z-slab sharding: every GPU marches the same global step lattice t_k = tmin + k*time_step (the
//     float accumulations `tcurr += time_step`, `next += vstep` are replayed identically) but only
//     evaluates the steps whose `curr` sample lies in a plane it owns; the per-pixel event key
//     (k<<1 | hit) is min-merged across GPUs by the host layer.
#include "dfusion_internal.h"

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
};

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

template <int MODE /* 0 = Points (tsdf_volume.cu:340-405), 1 = Depth (:272-338) */>
__global__ __launch_bounds__(256) void df_raycast_kernel(const DfRayArgs a)
{
    // XCD-aware, bijective renumbering: hardware block id b runs on XCD b % 8; give each XCD a contiguous
    // run of logical tiles (row-major), i.e. a horizontal band of the image.
    const int nwg = a.tiles_x * a.tiles_y;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int ty = wg / a.tiles_x, tx = wg - ty * a.tiles_x;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = tx * 16 + (w & 1) * 8 + (lane & 7);
    const int y = ty * 16 + (w >> 1) * 8 + (lane >> 3);
    if (x >= a.cols || y >= a.rows) return;

    const float qn = qnanf_();
    const f3 org = mk3(a.aff.t[0], a.aff.t[1], a.aff.t[2]);
    // device.hpp:43-48 with z = 1.f
    const f3 rp = mk3(1.f * ((float)x - a.cx) * a.finvx, 1.f * ((float)y - a.cy) * a.finvy, 1.f);
    const f3 dir = normalized3(mat3_mul(a.aff.R, rp));                                // :354
    const f3 vsi = mk3(a.vsix, a.vsiy, a.vsiz);

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

    uint32_t key = 0xffffffffu;
    bool hit = false;
    float t_hit = 0.f;
    f3 p_curr = org, p_next = org;

    if (tmin < tmax) {                                                                // :366
        tmax -= a.time_step;                                                          // :369
        const f3 vstep = scale3(dir, a.time_step);
        f3 next = add3(org, scale3(dir, tmin));
        // fetch_tsdf, :262-270 (__float2int_rn == rint, round-half-even)
        int zn = (int)rintf(next.z * a.vsiz);
        float tsdf_next = rc_vox(a, (int)rintf(next.x * a.vsix), (int)rintf(next.y * a.vsiy), zn);   // :373
        uint32_t k = 0;
        for (float tcurr = tmin; tcurr < tmax; tcurr += a.time_step, ++k) {           // :374
            const float tsdf_curr = tsdf_next;
            const f3 curr = next;
            const int zc = zn;
            next = add3(next, vstep);
            zn = (int)rintf(next.z * a.vsiz);
            const bool own_c = zc >= a.z_own0 && zc < a.z_own1;
            // the sample is needed as `next` of this step or as `curr` of the following one
            if (own_c || (zn >= a.z_own0 && zn < a.z_own1))
                tsdf_next = rc_vox(a, (int)rintf(next.x * a.vsix), (int)rintf(next.y * a.vsiy), zn);   // :380
            if (!own_c) continue;                                                     // another slab's step
            if (tsdf_curr < 0.f && tsdf_next > 0.f) { key = k << 1; break; }          // :381
            if (tsdf_curr > 0.f && tsdf_next < 0.f) {                                 // :384
                key = (k << 1) | 1u; hit = true; t_hit = tcurr; p_curr = curr; p_next = next;
                break;
            }
        }
    }

    float4 out_p = make_float4(qn, qn, qn, qn), out_n = make_float4(qn, qn, qn, qn);   // :351
    uint16_t out_d = 0;                                                               // :283
    if (hit) {                                                                        // refinement, wave reconverged
        const float Ft = rc_interpolate(a, mul3(p_curr, vsi));                        // :386
        const float Ftdt = rc_interpolate(a, mul3(p_next, vsi));                      // :387
        const float Ts = t_hit - (a.time_step * Ft) / (Ftdt - Ft);                    // :389
        const f3 vertex = add3(org, scale3(dir, Ts));
        const f3 normal = rc_normal(a, vertex);
        const float chk = normal.x * normal.y * normal.z;
        if (chk == chk) {                                                             // :394 !isnan
            const f3 n = mat3_mul(a.Rinv, normal);
            const f3 v = mat3_mul(a.Rinv, sub3(vertex, org));
            out_n = make_float4(n.x, n.y, n.z, 0.f);
            out_p = make_float4(v.x, v.y, v.z, 0.f);
            const float mm = v.z * 1000;                                              // :333
            out_d = (uint16_t)(mm <= 0.f ? 0 : (mm >= 65535.f ? 65535 : (int)mm));
        }
    }
    *reinterpret_cast<float4*>((char*)a.nrm + (size_t)y * a.npitch + 16 * (size_t)x) = out_n;
    if (MODE == 0) *reinterpret_cast<float4*>((char*)a.pts + (size_t)y * a.ppitch + 16 * (size_t)x) = out_p;
    else *reinterpret_cast<uint16_t*>((char*)a.depth + (size_t)y * a.dpitch + 2 * (size_t)x) = out_d;
    if (a.keys) a.keys[(size_t)y * a.cols + x] = key;
}

static int df_raycast_setup(DfRayArgs& a, const DfVolume& v, const DfSlab* slab, const float cam2vol[12], const float Rinv[9],
                            const float reproj[4], int cols, int rows, float step_factor, float delta_factor)
{
    if (!cam2vol || !Rinv || !reproj || cols <= 0 || rows <= 0 || !df_volume_valid(v)) return DF_E_INVALID;
    DfSlab s = df_slab_or_full(v, slab);
    if (!df_slab_valid(v, s)) return DF_E_INVALID;
    a.vol = (const uint32_t*)v.data; a.X = v.dims[0]; a.Y = v.dims[1]; a.Z = v.dims[2];
    a.z_store0 = s.z_store0; a.z_store_n = s.z_store_n; a.z_own0 = s.z_own0; a.z_own1 = s.z_own0 + s.z_own_n;
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
    hipLaunchKernelGGL(df_raycast_kernel<0>, dim3(a.tiles_x * a.tiles_y), dim3(256), 0, (hipStream_t)stream, a);
    DF_LAUNCH_CHECK();
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
    hipLaunchKernelGGL(df_raycast_kernel<1>, dim3(a.tiles_x * a.tiles_y), dim3(256), 0, (hipStream_t)stream, a);
    DF_LAUNCH_CHECK();
    return DF_OK;
}
