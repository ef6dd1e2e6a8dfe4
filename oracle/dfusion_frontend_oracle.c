/*
 * dfusion_frontend_oracle.c -- CPU restatement of the depth front-end and the projective-ICP reduction
 * (SURVEY.md 8(f) "next" #3) and of the warp-field data-term solve (#4, at the end of the file).  TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/ may load it (it is linked into
 * liboracle.so next to dfusion_oracle.c).
 *
 * What is restated (citations relative to /root/reference):
 *   bilateral filter        kfusion/src/cuda/imgproc.cu:11-59
 *   depth truncation        kfusion/src/cuda/imgproc.cu:66-85
 *   depth pyramid           kfusion/src/cuda/imgproc.cu:94-137
 *   normals + depth mask    kfusion/src/cuda/imgproc.cu:145-201      (USE_DEPTH build)
 *   points + normals        kfusion/src/cuda/imgproc.cu:210-252      (default build)
 *   resize depth+normals    kfusion/src/cuda/imgproc.cu:309-361
 *   resize points+normals   kfusion/src/cuda/imgproc.cu:368-414
 *   renderImage (x2), renderTangentColors   kfusion/src/cuda/imgproc.cu:420-583
 *   ICP correspondence      kfusion/src/cuda/proj_icp.cu:30-110 (both variants), row build :350-371
 *   ICP block reduction     kfusion/src/cuda/proj_icp.cu:112-348 + Block::reduce temp_utils.hpp:503-523
 *   ICP final reduction     kfusion/src/cuda/proj_icp.cu:373-397
 *
 * Arithmetic policy: as dfusion_oracle.c (IEEE fp32, fmaf only for __fmaf_rn / dot(), '/' for __fdividef, no other
 * contraction).  Two reference operations are hardware-defined approximations and have no exact restatement:
 *   __expf (imgproc.cu:37)   -> (float)exp((double)x)      [the SFU's ex2.approx differs in the last bits]
 *   rsqrt  (temp_utils.hpp:99) -> 1.0f / sqrtf(x)            [same choice as dfusion_oracle.c]
 * so parity for the bilateral filter and the normals is "parity unpinned" (no reference vectors exist either); the HIP
 * kernels are bit-compared against THIS restatement.
 * The float sums of the ICP reduction depend on the reduction tree; the tree of the reference (256-thread block,
 * strides 128..1; then 256 strided partial sums, strides 128..1) is reproduced exactly, so the 27 sums are comparable
 * bit for bit.
 * Deliberate deviations: mask_depth_kernel's guard `x < cols || y < rows` (:182) is && ; the int products
 * (value-depth)^2 (:35) and d00*d01 (:331) wrap modulo 2^32 as on the GPU ; NaN image coordinates in find_coresp count
 * as outside (reference: texture fetch at NaN).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef struct { float x, y, z; } f3;
static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline float dot3(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }          /* temp_utils.hpp:27-30 */
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }   /* :102-105 */
static inline f3 mat3_mul(const float *R, f3 v)                                                    /* device.hpp:71-72 */
{
    return mk3(dot3(mk3(R[0], R[1], R[2]), v), dot3(mk3(R[3], R[4], R[5]), v), dot3(mk3(R[6], R[7], R[8]), v));
}
static inline f3 aff_mul(const float *A, f3 v) { return add3(mat3_mul(A, v), mk3(A[9], A[10], A[11])); }   /* device.hpp:74 */
static inline f3 normalized3(f3 v) { float r = 1.0f / sqrtf(dot3(v, v)); return scale3(v, r); }    /* temp_utils.hpp:97-100 */
static inline float qnanf(void) { union { uint32_t u; float f; } c; c.u = 0x7fffffffu; return c.f; }
/* Reprojector::operator()(int u, int v, float z), device.hpp:42-47 ; ComputeIcpHelper::reproj(float,...) proj_icp.cu:39-44 */
static inline f3 reproj(float u, float v, float z, const float intr[4], float finvx, float finvy)
{
    return mk3(z * (u - intr[2]) * finvx, z * (v - intr[3]) * finvy, z);
}

#define PIX16(base, pitch, y, x) (((uint16_t *)((char *)(base) + (size_t)(y) * (pitch)))[x])
#define CPIX16(base, pitch, y, x) (((const uint16_t *)((const char *)(base) + (size_t)(y) * (pitch)))[x])
#define PIX4(base, pitch, y, x) ((float *)((char *)(base) + (size_t)(y) * (pitch)) + 4 * (size_t)(x))
#define CPIX4(base, pitch, y, x) ((const float *)((const char *)(base) + (size_t)(y) * (pitch)) + 4 * (size_t)(x))

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ---------------------------------------------------------------- bilateral (imgproc.cu:11-59) */
ORC_API void orc_bilateral(const uint16_t *src, size_t spitch, uint16_t *dst, size_t dpitch, int cols, int rows, int ksz,
                           float sigma_spatial, float sigma_depth /* metres */)
{
    sigma_depth *= 1000;                                             /* :50 */
    const float ss = 0.5f / (sigma_spatial * sigma_spatial), sd = 0.5f / (sigma_depth * sigma_depth);   /* :56 */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const int value = CPIX16(src, spitch, y, x);
            const int tx = imin(x - ksz / 2 + ksz, cols - 1);
            const int ty = imin(y - ksz / 2 + ksz, rows - 1);
            float sum1 = 0, sum2 = 0;
            for (int cy = imax(y - ksz / 2, 0); cy < ty; ++cy)
                for (int cx = imax(x - ksz / 2, 0); cx < tx; ++cx) {
                    const int depth = CPIX16(src, spitch, cy, cx);
                    const float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
                    const float color2 = (float)(int32_t)((uint32_t)(value - depth) * (uint32_t)(value - depth));
                    const float weight = (float)exp((double)(-(space2 * ss + color2 * sd)));            /* __expf, see header */
                    sum1 += (float)depth * weight;
                    sum2 += weight;
                }
            PIX16(dst, dpitch, y, x) = (uint16_t)(int)lrintf(sum1 / sum2);                          /* __float2int_rn */
        }
}

/* ---------------------------------------------------------------- truncation (imgproc.cu:66-85) */
ORC_API void orc_truncate_depth(uint16_t *depth, size_t pitch, int cols, int rows, float max_dist /* metres */)
{
    const uint16_t md = (uint16_t)(max_dist * 1000.f);               /* :83 static_cast<ushort> */
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x)
            if (PIX16(depth, pitch, y, x) > md) PIX16(depth, pitch, y, x) = 0;
}

/* ---------------------------------------------------------------- cloud -> depth (imgproc.cu:273-282): depth = z * 1000 as ushort.  The
 * conversion is the CUDA target's (cvt.rzi.u16.f32): toward zero, saturating, NaN -> 0; in C that range is spelled out (the plain cast is
 * undefined outside [0, 65536) and for NaN). */
ORC_API void orc_cloud_to_depth(const float *cloud, size_t cpitch, uint16_t *depth, size_t dpitch, int cols, int rows)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float mm = ((const float *)((const char *)cloud + (size_t)y * cpitch))[4 * x + 2] * 1000;   /* :280 */
            PIX16(depth, dpitch, y, x) = mm >= 65535.f ? (uint16_t)65535 : (mm > 0.f ? (uint16_t)mm : (uint16_t)0);
        }
}

/* ---------------------------------------------------------------- pyramid (imgproc.cu:94-137); dst is (rows/2) x (cols/2) */
ORC_API void orc_depth_pyramid(const uint16_t *src, size_t spitch, int scols, int srows, uint16_t *dst, size_t dpitch,
                               float sigma_depth /* metres */)
{
    sigma_depth *= 1000;                                             /* :130 */
    const float thr = sigma_depth * 3;                               /* :135 */
    const int dcols = scols / 2, drows = srows / 2;                  /* imgproc.cpp:36 */
    const int D = 5;
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            const int center = CPIX16(src, spitch, 2 * y, 2 * x);
            const int tx = imin(2 * x - D / 2 + D, scols - 1);
            const int ty = imin(2 * y - D / 2 + D, srows - 1);
            int sum = 0, count = 0;
            for (int cy = imax(0, 2 * y - D / 2); cy < ty; ++cy)
                for (int cx = imax(0, 2 * x - D / 2); cx < tx; ++cx) {
                    const int val = CPIX16(src, spitch, cy, cx);
                    if ((float)abs(val - center) < thr) { sum += val; ++count; }
                }
            PIX16(dst, dpitch, y, x) = (uint16_t)(count == 0 ? 0 : sum / count);
        }
}

/* shared by the two normal kernels: returns 1 and fills n (already negated, :170/:243) and v00 if the 3 depths are valid */
static inline int normal_at(const uint16_t *depth, size_t pitch, int cols, int rows, int x, int y, const float intr[4],
                            float finvx, float finvy, f3 *n, f3 *v00)
{
    if (!(x < cols - 1 && y < rows - 1)) return 0;
    const float z00 = (float)CPIX16(depth, pitch, y, x) * 0.001f;
    const float z01 = (float)CPIX16(depth, pitch, y, x + 1) * 0.001f;
    const float z10 = (float)CPIX16(depth, pitch, y + 1, x) * 0.001f;
    if (!(z00 * z01 * z10 != 0)) return 0;
    *v00 = reproj((float)x, (float)y, z00, intr, finvx, finvy);
    const f3 v01 = reproj((float)(x + 1), (float)y, z01, intr, finvx, finvy);
    const f3 v10 = reproj((float)x, (float)(y + 1), z10, intr, finvx, finvy);
    const f3 c = normalized3(cross3(sub3(v01, *v00), sub3(v10, *v00)));
    *n = mk3(-c.x, -c.y, -c.z);
    return 1;
}

/* ---------------------------------------------------------------- computeNormalsAndMaskDepth (imgproc.cu:145-201) */
ORC_API void orc_compute_normals_mask_depth(uint16_t *depth, size_t dpitch, float *normals, size_t npitch, int cols, int rows,
                                            const float intr[4])
{
    const float finvx = 1.f / intr[0], finvy = 1.f / intr[1];        /* Reprojector ctor, precomp.cpp:55 */
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float *o = PIX4(normals, npitch, y, x);
            f3 n, v;
            if (normal_at(depth, dpitch, cols, rows, x, y, intr, finvx, finvy, &n, &v)) { o[0] = n.x; o[1] = n.y; o[2] = n.z; o[3] = 0.f; }
            else { o[0] = o[1] = o[2] = qnanf(); o[3] = 0.f; }
        }
    for (int y = 0; y < rows; ++y)                                   /* mask_depth_kernel, after all normals are known */
        for (int x = 0; x < cols; ++x)
            if (isnan(PIX4(normals, npitch, y, x)[0])) PIX16(depth, dpitch, y, x) = 0;
}

/* ---------------------------------------------------------------- computePointNormals (imgproc.cu:210-252) */
ORC_API void orc_compute_point_normals(const uint16_t *depth, size_t dpitch, float *points, size_t ppitch, float *normals,
                                       size_t npitch, int cols, int rows, const float intr[4])
{
    const float finvx = 1.f / intr[0], finvy = 1.f / intr[1];
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            float *p = PIX4(points, ppitch, y, x), *o = PIX4(normals, npitch, y, x);
            f3 n, v;
            if (normal_at(depth, dpitch, cols, rows, x, y, intr, finvx, finvy, &n, &v)) {
                o[0] = n.x; o[1] = n.y; o[2] = n.z; o[3] = 0.f;
                p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = 0.f;
            } else {
                p[0] = p[1] = p[2] = p[3] = qnanf();                 /* :220 all four NaN */
                o[0] = o[1] = o[2] = o[3] = qnanf();
            }
        }
}

/* ---------------------------------------------------------------- resizeDepthNormals (imgproc.cu:309-361); dst is half size */
ORC_API void orc_resize_depth_normals(const uint16_t *dsrc, size_t dspitch, const float *nsrc, size_t nspitch, int scols,
                                      int srows, uint16_t *ddst, size_t ddpitch, float *ndst, size_t ndpitch)
{
    const int dcols = scols / 2, drows = srows / 2;
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            const int xs = 2 * x, ys = 2 * y;
            const int d00 = CPIX16(dsrc, dspitch, ys, xs), d01 = CPIX16(dsrc, dspitch, ys, xs + 1);
            const int d10 = CPIX16(dsrc, dspitch, ys + 1, xs), d11 = CPIX16(dsrc, dspitch, ys + 1, xs + 1);
            uint16_t d = 0;
            float n[4] = {qnanf(), qnanf(), qnanf(), qnanf()};
            if ((int32_t)((uint32_t)d00 * (uint32_t)d01) != 0 && (int32_t)((uint32_t)d10 * (uint32_t)d11) != 0) {
                d = (uint16_t)((d00 + d01 + d10 + d11) / 4);
                const float *a = CPIX4(nsrc, nspitch, ys, xs), *b = CPIX4(nsrc, nspitch, ys, xs + 1);
                const float *c = CPIX4(nsrc, nspitch, ys + 1, xs), *e = CPIX4(nsrc, nspitch, ys + 1, xs + 1);
                for (int i = 0; i < 3; ++i) n[i] = (float)((double)(a[i] + b[i] + c[i] + e[i]) * 0.25);   /* :343-345 `*0.25` is double */
            }
            PIX16(ddst, ddpitch, y, x) = d;
            memcpy(PIX4(ndst, ndpitch, y, x), n, 16);
        }
}

/* ---------------------------------------------------------------- renderImage / renderTangentColors (imgproc.cu:420-583)
 * The Phong shading KinFu::renderImage shows (kinfu.cpp:312-343,408-436).  __powf(x, 20) is a hardware approximation on the
 * reference's side; (float)pow((double)x, 20.0) stands in (as (float)exp((double)x) does for __expf), __saturatef clamps to [0, 1]
 * with NaN -> 0, the unsigned char casts truncate.  Pinned against the reference's own kernels in tests/test_oracle_refcu.py. */
static inline float saturatef(float a) { return a != a ? 0.f : (a < 0.f ? 0.f : (a > 1.f ? 1.f : a)); }
static inline void shade_pixel(int have, f3 P, f3 N, f3 light, int y, int rows, uint8_t out[4])
{
    f3 color;
    if (!have) {
        const f3 bgr1 = mk3(4.f / 255.f, 2.f / 255.f, 2.f / 255.f), bgr2 = mk3(236.f / 255.f, 120.f / 255.f, 120.f / 255.f);
        const float w = (float)y / rows;
        color = add3(scale3(bgr1, 1 - w), scale3(bgr2, w));                                          /* :439-444 */
    } else {
        const float Ka = 0.3f, Kd = 0.5f, Ks = 0.2f, n = 20.f, Ax = 1.f, Dx = 1.f, Sx = 1.f, Lx = 1.f;
        const f3 L = normalized3(sub3(light, P));
        const f3 V = normalized3(sub3(mk3(0.f, 0.f, 0.f), P));
        const f3 R = normalized3(sub3(scale3(scale3(N, 2.f), dot3(N, L)), L));                       /* 2 * N * dot(N, L) - L */
        const float Ix = Ax * Ka * Dx + Lx * Kd * Dx * fmaxf(0.f, dot3(N, L)) + Lx * Ks * Sx * (float)pow((double)fmaxf(0.f, dot3(R, V)), (double)n);
        color = mk3(Ix, Ix, Ix);
    }
    out[0] = (uint8_t)(saturatef(color.x) * 255.f); out[1] = (uint8_t)(saturatef(color.y) * 255.f);
    out[2] = (uint8_t)(saturatef(color.z) * 255.f); out[3] = 0;
}
ORC_API void orc_render_points(const float *points, size_t ppitch, const float *normals, size_t npitch, int cols, int rows,
                               const float light[3], uint8_t *image, size_t ipitch)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float *p = (const float *)((const char *)points + (size_t)y * ppitch) + 4 * x;
            const float *n = (const float *)((const char *)normals + (size_t)y * npitch) + 4 * x;
            shade_pixel(!isnan(p[0]), mk3(p[0], p[1], p[2]), mk3(n[0], n[1], n[2]), mk3(light[0], light[1], light[2]), y, rows,
                        image + (size_t)y * ipitch + 4 * x);                                        /* :474-523 */
        }
}
ORC_API void orc_render_depth(const uint16_t *depth, size_t dpitch, const float *normals, size_t npitch, int cols, int rows,
                              const float intr[4], const float light[3], uint8_t *image, size_t ipitch)
{
    const float finvx = 1.f / intr[0], finvy = 1.f / intr[1];
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const int d = *(const uint16_t *)((const char *)depth + (size_t)y * dpitch + 2 * (size_t)x);
            const float *n = (const float *)((const char *)normals + (size_t)y * npitch) + 4 * x;
            const float z = d * 0.001f;
            const f3 P = mk3(z * (x - intr[2]) * finvx, z * (y - intr[3]) * finvy, z);              /* Reprojector, device.hpp:42-48 */
            shade_pixel(d != 0, P, mk3(n[0], n[1], n[2]), mk3(light[0], light[1], light[2]), y, rows,
                        image + (size_t)y * ipitch + 4 * x);                                        /* :420-471 */
        }
}
ORC_API void orc_render_tangent_colors(const float *normals, size_t npitch, int cols, int rows, uint8_t *image, size_t ipitch)
{
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            const float *n = (const float *)((const char *)normals + (size_t)y * npitch) + 4 * x;
            uint8_t *o = image + (size_t)y * ipitch + 4 * x;
            /* (unsigned char)(float): out-of-range and NaN values are undefined in C; the GPU conversion saturates (NaN -> 0) */
            const float r = (5.f - n[0] * 3.5f) * 25.5f, g = (5.f - n[1] * 2.5f) * 25.5f, b = (5.f - n[2] * 3.5f) * 25.5f;
            o[0] = (uint8_t)(b != b ? 0.f : fminf(fmaxf(b, 0.f), 255.f)); o[1] = (uint8_t)(g != g ? 0.f : fminf(fmaxf(g, 0.f), 255.f));
            o[2] = (uint8_t)(r != r ? 0.f : fminf(fmaxf(r, 0.f), 255.f)); o[3] = 0;                  /* :552-573: (b, g, r, 0) */
        }
}

/* ---------------------------------------------------------------- resizePointsNormals (imgproc.cu:368-414) */
ORC_API void orc_resize_points_normals(const float *vsrc, size_t vspitch, const float *nsrc, size_t nspitch, int scols, int srows,
                                       float *vdst, size_t vdpitch, float *ndst, size_t ndpitch)
{
    const int dcols = scols / 2, drows = srows / 2;
    for (int y = 0; y < drows; ++y)
        for (int x = 0; x < dcols; ++x) {
            const int xs = 2 * x, ys = 2 * y;
            float *vo = PIX4(vdst, vdpitch, y, x), *no = PIX4(ndst, ndpitch, y, x);
            vo[0] = vo[1] = vo[2] = qnanf(); vo[3] = 0.f;
            no[0] = no[1] = no[2] = qnanf(); no[3] = 0.f;
            const float *a = CPIX4(vsrc, vspitch, ys, xs), *b = CPIX4(vsrc, vspitch, ys, xs + 1);
            const float *c = CPIX4(vsrc, vspitch, ys + 1, xs), *e = CPIX4(vsrc, vspitch, ys + 1, xs + 1);
            if (!isnan(a[0] * b[0] * c[0] * e[0])) {
                for (int i = 0; i < 3; ++i) vo[i] = (((a[i] + b[i]) + c[i]) + e[i]) * 0.25f;
                const float *na = CPIX4(nsrc, nspitch, ys, xs), *nb = CPIX4(nsrc, nspitch, ys, xs + 1);
                const float *nc = CPIX4(nsrc, nspitch, ys + 1, xs), *ne = CPIX4(nsrc, nspitch, ys + 1, xs + 1);
                for (int i = 0; i < 3; ++i) no[i] = (((na[i] + nb[i]) + nc[i]) + ne[i]) * 0.25f;
            }
        }
}

/* ---------------------------------------------------------------- projective ICP sums
 * find_coresp (proj_icp.cu:47-110) for the pixel, returns the reference's filter code (0 = accepted). */
typedef struct {
    int cols, rows;
    const float *aff;               /* curr -> prev, 12 floats */
    float intr[4], finvx, finvy;    /* level intrinsics (setLevelIntr, projective_icp.cpp:17-23) */
    float min_cosine, dist2_thres;
    const float *vcurr; size_t vcpitch; const float *ncurr; size_t ncpitch;
    const float *vprev; size_t vppitch; const float *nprev; size_t nppitch;
    const uint16_t *dcurr; size_t dcpitch; const uint16_t *dprev; size_t dppitch;   /* depth variant */
} IcpArgs;

static inline int find_coresp(const IcpArgs *A, int x, int y, f3 *nd, f3 *d, f3 *s)
{
    if (A->dcurr) {
        const int src_z = CPIX16(A->dcurr, A->dcpitch, y, x);
        if (src_z == 0) return 40;
        *s = aff_mul(A->aff, reproj((float)x, (float)y, (float)src_z * 0.001f, A->intr, A->finvx, A->finvy));
    } else {
        const float *p = CPIX4(A->vcurr, A->vcpitch, y, x);
        if (isnan(p[0])) return 40;
        *s = aff_mul(A->aff, mk3(p[0], p[1], p[2]));
    }
    const float u = fmaf(A->intr[0], s->x / s->z, A->intr[2]);      /* :33-34 */
    const float v = fmaf(A->intr[1], s->y / s->z, A->intr[3]);
    if (s->z <= 0 || !(u >= 0 && v >= 0 && u < (float)A->cols && v < (float)A->rows)) return 80;
    const int ui = (int)u, vi = (int)v;                                 /* tex2D point filter */
    if (A->dcurr) {
        const int dst_z = CPIX16(A->dprev, A->dppitch, vi, ui);
        if (dst_z == 0) return 120;
        *d = reproj(u, v, (float)dst_z * 0.001f, A->intr, A->finvx, A->finvy);
    } else {
        const float *q = CPIX4(A->vprev, A->vppitch, vi, ui);
        if (isnan(q[0])) return 120;
        *d = mk3(q[0], q[1], q[2]);
    }
    const f3 diff = sub3(*s, *d);
    if (dot3(diff, diff) > A->dist2_thres) return 160;
    const float *nc = CPIX4(A->ncurr, A->ncpitch, y, x);
    const f3 ns = mat3_mul(A->aff, mk3(nc[0], nc[1], nc[2]));
    const float *np = CPIX4(A->nprev, A->nppitch, vi, ui);
    *nd = mk3(np[0], np[1], np[2]);
    const float cosine = fabsf(dot3(ns, *nd));
    if (cosine < A->min_cosine) return 200;
    return 0;
}

/* Block::reduce<256> (temp_utils.hpp:503-523): only v[0] matters; pairs (t, t+s) for s = 128..1 */
static inline float tree256(float *v)
{
    for (int s = 128; s >= 1; s >>= 1)
        for (int t = 0; t < s; ++t) v[t] = v[t] + v[t + s];
    return v[0];
}

static void icp_sums(const IcpArgs *A, float out[27], int *accepted)
{
    const int gx = (A->cols + 31) / 32, gy = (A->rows + 7) / 8;       /* proj_icp.cu:406-407 */
    const int partials = gx * gy;
    float *buf = (float *)malloc((size_t)27 * partials * sizeof(float));
    int acc = 0;
#pragma omp parallel for schedule(static) reduction(+ : acc)
    for (int b = 0; b < partials; ++b) {
        const int bx = b % gx, by = b / gx;                          /* pos = blockIdx.x + gridDim.x * blockIdx.y (:117) */
        float rows7[256][7];
        for (int t = 0; t < 256; ++t) {
            const int x = (t & 31) + bx * 32, y = (t >> 5) + by * 8; /* tid = ty*32 + tx */
            f3 n, d, s;
            float *r = rows7[t];
            const int filtered = (x < A->cols && y < A->rows) ? find_coresp(A, x, y, &n, &d, &s) : 1;
            if (!filtered) {
                const f3 c = cross3(s, n);
                r[0] = c.x; r[1] = c.y; r[2] = c.z; r[3] = n.x; r[4] = n.y; r[5] = n.z;
                r[6] = dot3(n, sub3(d, s));
                ++acc;
            } else {
                r[0] = r[1] = r[2] = r[3] = r[4] = r[5] = r[6] = 0.f;
            }
        }
        int k = 0;
        float v[256];
        for (int i = 0; i < 6; ++i)
            for (int j = i; j < 7; ++j) {                            /* :128-346 */
                for (int t = 0; t < 256; ++t) v[t] = rows7[t][i] * rows7[t][j];
                buf[(size_t)k * partials + b] = tree256(v);
                ++k;
            }
    }
    for (int k = 0; k < 27; ++k) {                                   /* icp_final_reduce_kernel :373-397 */
        float v[256];
        for (int t = 0; t < 256; ++t) {
            float sum = 0.f;
            for (int i = t; i < partials; i += 256) sum += buf[(size_t)k * partials + i];
            v[t] = sum;
        }
        out[k] = tree256(v);
    }
    free(buf);
    if (accepted) *accepted = acc;
}

/* points variant (default build).  intr = level intrinsics {fx, fy, cx, cy} (already divided by 2^level). */
ORC_API void orc_icp_sums_points(const float *vcurr, size_t vcpitch, const float *ncurr, size_t ncpitch, const float *vprev,
                                 size_t vppitch, const float *nprev, size_t nppitch, int cols, int rows, const float aff[12],
                                 const float intr[4], float dist2_thres, float min_cosine, float out[27], int *accepted)
{
    IcpArgs A;
    memset(&A, 0, sizeof(A));
    A.cols = cols; A.rows = rows; A.aff = aff;
    memcpy(A.intr, intr, 16); A.finvx = 1.f / intr[0]; A.finvy = 1.f / intr[1];
    A.min_cosine = min_cosine; A.dist2_thres = dist2_thres;
    A.vcurr = vcurr; A.vcpitch = vcpitch; A.ncurr = ncurr; A.ncpitch = ncpitch;
    A.vprev = vprev; A.vppitch = vppitch; A.nprev = nprev; A.nppitch = nppitch;
    icp_sums(&A, out, accepted);
}

/* depth variant (USE_DEPTH build) */
ORC_API void orc_icp_sums_depth(const uint16_t *dcurr, size_t dcpitch, const float *ncurr, size_t ncpitch, const uint16_t *dprev,
                                size_t dppitch, const float *nprev, size_t nppitch, int cols, int rows, const float aff[12],
                                const float intr[4], float dist2_thres, float min_cosine, float out[27], int *accepted)
{
    IcpArgs A;
    memset(&A, 0, sizeof(A));
    A.cols = cols; A.rows = rows; A.aff = aff;
    memcpy(A.intr, intr, 16); A.finvx = 1.f / intr[0]; A.finvy = 1.f / intr[1];
    A.min_cosine = min_cosine; A.dist2_thres = dist2_thres;
    A.dcurr = dcurr; A.dcpitch = dcpitch; A.ncurr = ncurr; A.ncpitch = ncpitch;
    A.dprev = dprev; A.dppitch = dppitch; A.nprev = nprev; A.nppitch = nppitch;
    icp_sums(&A, out, accepted);
}

/* ---------------------------------------------------------------- warp-field data term
 * The energy the reference hands to Opt (kfusion/solvers/dynamicfusion.t:26-52) and to Ceres (warp_field.cpp:117-163,
 * optimisation.hpp:36-71):  E(T) = sum_v | (live_v - canonical_v) - sum_i w_vi T_{n_vi} |^2  over the node translations.
 * Opt (pinned by the reference's CMake to the niessner/Opt checkout) and Ceres are third-party and absent from the reference
 * tree; E is linear least squares, so what they converge to is the least-squares solution reached from the current translations.
 * This restates the conjugate-gradient solve of dynamicfusion_amd/csrc/dfusion_solver.hip operation for operation (thread-strided
 * partial sums, tree orders), so that the two are comparable bit for bit; the reference's own solver tests
 * (tests/ceres_warp_test.cpp: after energy_data + warp the source vertices sit on the targets within 1e-3) pin the behaviour. */
void orc_knn(const float *pos, int M, const float *queries, int N, int k, int *idx_out, float *d2_out);
void orc_node_translation(const float dq[8], float out[4]);

static void sv_wt_apply(const unsigned *off, const unsigned *svals, const float *w, int k, int M, const float *u, float lambda,
                        const float *p, float *out)
{
    for (int n = 0; n < M; ++n) {
        float part[3][256];
        for (int t = 0; t < 256; ++t) {                      /* thread t of the node's workgroup: entries t, t + 256, ... */
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (unsigned i = off[n] + (unsigned)t; i < off[n + 1]; i += 256) {
                const unsigned e = svals[i], v = e / (unsigned)k;
                const float we = w[e];
                sx = sx + we * u[3 * v]; sy = sy + we * u[3 * v + 1]; sz = sz + we * u[3 * v + 2];
            }
            part[0][t] = sx; part[1][t] = sy; part[2][t] = sz;
        }
        for (int c = 0; c < 3; ++c) {
            float r = tree256(part[c]);                      /* pairs (t, t + s), s = 128 .. 1 */
            if (p) r = r + lambda * p[3 * n + c];
            out[3 * n + c] = r;
        }
    }
}

static void sv_w_apply(const float *w, const unsigned *keys, int N, int k, int M, const float *p, float *u)
{
    for (int v = 0; v < N; ++v) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int j = 0; j < k; ++j) {
            const int e = v * k + j;
            const unsigned n = keys[e];
            if (n < (unsigned)M) { const float wj = w[e]; sx = sx + wj * p[3 * n]; sy = sy + wj * p[3 * n + 1]; sz = sz + wj * p[3 * n + 2]; }
        }
        u[3 * v] = sx; u[3 * v + 1] = sy; u[3 * v + 2] = sz;
    }
}

/* 1024-thread block sum: thread t owns elements t, t + 1024, ... (partials given), then the tree 512 .. 1 */
static float sv_tree1024(float *part) { for (int st = 512; st >= 1; st >>= 1) for (int t = 0; t < st; ++t) part[t] = part[t] + part[t + st]; return part[0]; }

static float sv_energy(const float *e, int N)
{
    float part[3][1024];
    memset(part, 0, sizeof(part));
    for (int t = 0; t < 1024; ++t)
        for (int v = t; v < N; v += 1024)
            for (int c = 0; c < 3; ++c) part[c][t] = part[c][t] + e[3 * v + c] * e[3 * v + c];
    const float s0 = sv_tree1024(part[0]), s1 = sv_tree1024(part[1]), s2 = sv_tree1024(part[2]);
    return (s0 + s1) + s2;
}

ORC_API void orc_solve_data_term(const float *pos, const float *dq, const float *sigma, int M, int k, const float *canonical,
                                 const float *live, int N, int iters, float lambda, float *dq_out, float energy[2])
{
    const size_t E = (size_t)N * k;
    int *idx = (int *)malloc(E * sizeof(int)); float *d2 = (float *)malloc(E * sizeof(float));
    float *w = (float *)calloc(E, sizeof(float)); unsigned *keys = (unsigned *)malloc(E * sizeof(unsigned));
    unsigned *svals = (unsigned *)malloc(E * sizeof(unsigned)); unsigned *off = (unsigned *)calloc((size_t)M + 2, sizeof(unsigned));
    float *e0 = (float *)calloc((size_t)N * 3, sizeof(float)), *u = (float *)calloc((size_t)N * 3, sizeof(float));
    float *node_t = (float *)malloc((size_t)M * 16);
    float *x = (float *)calloc((size_t)M * 3, 4), *r = (float *)calloc((size_t)M * 3, 4), *p = (float *)calloc((size_t)M * 3, 4), *q = (float *)calloc((size_t)M * 3, 4);
    for (int n = 0; n < M; ++n) orc_node_translation(dq + 8 * n, node_t + 4 * n);
    orc_knn(pos, M, canonical, N, k, idx, d2);
    for (int v = 0; v < N; ++v) {
        const float *c = canonical + 3 * v, *l = live + 3 * v;
        const int valid = !(isnan(c[0]) || isnan(c[1]) || isnan(c[2]) || isnan(l[0]) || isnan(l[1]) || isnan(l[2]));
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int j = 0; j < k; ++j) {
            const size_t e = (size_t)v * k + j;
            const int n = valid ? idx[e] : M;
            float wj = 0.f;
            if (valid) {
                const float sg = sigma[n];
                wj = (float)exp((double)(-d2[e] / (2 * sg * sg)));
                const float *t = node_t + 4 * n;
                sx = sx + wj * t[1]; sy = sy + wj * t[2]; sz = sz + wj * t[3];
            }
            w[e] = wj; keys[e] = (unsigned)n;
        }
        e0[3 * v] = valid ? (l[0] - c[0]) - sx : 0.f; e0[3 * v + 1] = valid ? (l[1] - c[1]) - sy : 0.f; e0[3 * v + 2] = valid ? (l[2] - c[2]) - sz : 0.f;
    }
    /* node-major entry list, ascending entry index inside a node (what a stable sort by node id produces) */
    for (size_t e = 0; e < E; ++e) if (keys[e] < (unsigned)M) ++off[keys[e] + 1];
    for (int n = 0; n < M; ++n) off[n + 1] += off[n];
    { unsigned *cur = (unsigned *)malloc((size_t)M * sizeof(unsigned)); memcpy(cur, off, (size_t)M * sizeof(unsigned));
      for (size_t e = 0; e < E; ++e) if (keys[e] < (unsigned)M) svals[cur[keys[e]]++] = (unsigned)e;
      free(cur); }
    if (energy) energy[0] = sv_energy(e0, N);
    sv_wt_apply(off, svals, w, k, M, e0, 0.f, NULL, r);
    float rr[3], rr0[3];
    {   /* df_sv_init_kernel */
        float part[3][1024]; memset(part, 0, sizeof(part));
        for (int t = 0; t < 1024; ++t) for (int n = t; n < M; n += 1024) for (int c = 0; c < 3; ++c) { const float rv = r[3 * n + c]; p[3 * n + c] = rv; part[c][t] = part[c][t] + rv * rv; }
        for (int c = 0; c < 3; ++c) rr0[c] = rr[c] = sv_tree1024(part[c]);
    }
    for (int it = 0; it < iters; ++it) {
        sv_w_apply(w, keys, N, k, M, p, u);
        sv_wt_apply(off, svals, w, k, M, u, lambda, p, q);
        float part[3][1024], pq[3], alpha[3], rn[3], beta[3];
        memset(part, 0, sizeof(part));
        for (int t = 0; t < 1024; ++t) for (int n = t; n < M; n += 1024) for (int c = 0; c < 3; ++c) part[c][t] = part[c][t] + p[3 * n + c] * q[3 * n + c];
        for (int c = 0; c < 3; ++c) { pq[c] = sv_tree1024(part[c]); alpha[c] = (pq[c] > 0.f && rr[c] > 0.f) ? rr[c] / pq[c] : 0.f; }
        memset(part, 0, sizeof(part));
        for (int t = 0; t < 1024; ++t) for (int n = t; n < M; n += 1024) for (int c = 0; c < 3; ++c) {
            x[3 * n + c] = x[3 * n + c] + alpha[c] * p[3 * n + c];
            const float rv = r[3 * n + c] - alpha[c] * q[3 * n + c];
            r[3 * n + c] = rv; part[c][t] = part[c][t] + rv * rv;
        }
        for (int c = 0; c < 3; ++c) { rn[c] = sv_tree1024(part[c]); beta[c] = (alpha[c] != 0.f && rr[c] > 0.f) ? rn[c] / rr[c] : 0.f; }
        for (int n = 0; n < M; ++n) for (int c = 0; c < 3; ++c) p[3 * n + c] = r[3 * n + c] + beta[c] * p[3 * n + c];
        for (int c = 0; c < 3; ++c) rr[c] = (alpha[c] != 0.f && rn[c] > 1.0e-10f * rr0[c]) ? rn[c] : 0.f;   /* converged: frozen */
    }
    if (energy) {
        sv_w_apply(w, keys, N, k, M, x, u);
        for (size_t i = 0; i < (size_t)N * 3; ++i) u[i] = e0[i] - u[i];
        energy[1] = sv_energy(u, N);
    }
    for (int n = 0; n < M; ++n) {       /* encodeTranslation (dual_quaternion.hpp:82-85): 0.5 * (0, T) * rotation_ ; product quaternion.hpp:186-194 */
        const float *ro = dq + 8 * n, *t = node_t + 4 * n;
        const float aw = 0.5f * 0.f, ax = 0.5f * (t[1] + x[3 * n]), ay = 0.5f * (t[2] + x[3 * n + 1]), az = 0.5f * (t[3] + x[3 * n + 2]);
        const float bw = ro[0], bx = ro[1], by = ro[2], bz = ro[3];
        float *o = dq_out + 8 * n;
        o[0] = bw; o[1] = bx; o[2] = by; o[3] = bz;
        o[4] = ((aw * bw) - (ax * bx) - (ay * by) - (az * bz));
        o[5] = ((aw * bx) + (ax * bw) + (ay * bz) - (az * by));
        o[6] = ((aw * by) - (ax * bz) + (ay * bw) + (az * bx));
        o[7] = ((aw * bz) + (ax * by) - (ay * bx) + (az * bw));
    }
    free(idx); free(d2); free(w); free(keys); free(svals); free(off); free(e0); free(u); free(node_t); free(x); free(r); free(p); free(q);
}
