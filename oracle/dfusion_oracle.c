/*
 * dfusion_oracle.c -- CPU restatement of the DynamicFusion per-frame hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path is the HIP library
 * (dynamicfusion_amd/csrc) and never routes through anything in oracle/.
 *
 * What is restated (all citations relative to /root/reference):
 *   compute_dists        kfusion/src/cuda/imgproc.cu:259-272, host imgproc.cpp:87-91
 *   pack/unpack          kfusion/src/cuda/device.hpp:53-61
 *   voxel indexing       kfusion/src/cuda/device.hpp:17-27
 *   clear                kfusion/src/cuda/tsdf_volume.cu:15-28
 *   integrate (rigid)    kfusion/src/cuda/tsdf_volume.cu:51-112, host :141-161
 *   raycast              kfusion/src/cuda/tsdf_volume.cu:202-474
 *   extract cloud/normals kfusion/src/cuda/tsdf_volume.cu:511-710,714-795
 *   k-NN                 kfusion/src/warp_field.cpp:247-251 + knn_point_cloud.hpp:16-32 + the vendored
 *                        nanoflann v1.2.3 (kd-tree build, visit order and result set restated, so that exact
 *                        distance ties resolve as in the reference)
 *   weighting            kfusion/src/warp_field.cpp:238-241
 *   DQB                  kfusion/src/warp_field.cpp:203-217
 *   DualQuaternion math  kfusion/src/utils/dual_quaternion.hpp:59-63,120-125,204-210
 *   Quaternion math      kfusion/src/utils/quaternion.hpp:124-130,172-194,211-228
 *   warp                 kfusion/src/warp_field.cpp:180-195
 *   warped integrate     composition DQB o TsdfIntegrator (the reference has
 *                        no per-voxel warped integrate -- its oracle IS this composition)
 *
 * Arithmetic policy: IEEE fp32, explicit fmaf exactly where the reference
 * writes __fmaf_rn / dot(), IEEE '/' and sqrtf for __fdividef / __fsqrt_rn / rsqrt,
 * F16C for __float2half_rn / __half2float, lrintf (RN-even) for __float2int_rn, floorf for
 * __float2int_rd.  Build with -ffp-contract=off so nothing else is fused.
 *
 * Parity pinning: the reference has NO tests for integrate / raycast / dists / clear, so that
 * part is "parity unpinned" by golden vectors (pinned by source restatement only).  The
 * quaternion / dual-quaternion / k-NN / DQB part is pinned against the reference's own
 * headers compiled unmodified (oracle/_ref, built by oracle/Makefile) and against the
 * known-answer vectors of tests/utils/test_quaternion.cc and test_dual_quaternion.cc.
 *
 * Deliberate deviations (each is also a documented policy of the HIP path):
 *   - NaN pixel coordinates in integrate => skip (reference relies on texture border).
 *   - fetch_tsdf index is clamped to the stored range (reference reads unchecked).
 *   - exp() in weighting is the double overload (what gcc 5 / Ubuntu 16.04, the
 *     reference's platform, resolves `exp(float)` to with <cmath> only), then cast to float.
 *   - (ushort)(z*1000) in the depth raycast saturates to [0,65535].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <immintrin.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

typedef struct OrcVolume {
    void *data;          /* ushort2 voxels, first stored plane = slab z_store0 */
    int dims[3];         /* GLOBAL dims (x,y,z) */
    float voxel_size[3];
    float trunc_dist;
    int max_weight;
} OrcVolume;             /* field-for-field device::TsdfVolume, kfusion/src/internal.hpp:29-49 */

typedef struct OrcSlab {
    int z_store0, z_store_n;   /* planes physically present behind data */
    int z_own0, z_own_n;       /* planes this shard integrates / owns ray steps for */
} OrcSlab;

typedef struct { float x, y, z; } f3;

/* ---------------------------------------------------------------- half <-> float */
static inline uint16_t f2h(float f) { return (uint16_t)_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT); }
static inline float h2f(uint16_t h) { return _cvtsh_ss(h); }

ORC_API uint16_t orc_float2half(float f) { return f2h(f); }
ORC_API float orc_half2float(uint16_t h) { return h2f(h); }

/* ---------------------------------------------------------------- vector helpers
 * temp_utils.hpp:27-30 : dot = fma(x1,x2, fma(y1,y2, z1*z2))
 * device.hpp:71-74     : Mat3f*v = three dots ; Aff3f*v = R*v + t                      */
static inline float dot3(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline f3 scale3(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
static inline f3 mat3_mul(const float *R, f3 v)
{
    return mk3(dot3(mk3(R[0], R[1], R[2]), v), dot3(mk3(R[3], R[4], R[5]), v), dot3(mk3(R[6], R[7], R[8]), v));
}
/* aff = R[9] row-major then t[3] (internal.hpp:26-27 via device_cast precomp.hpp:19-28) */
static inline f3 aff_mul(const float *A, f3 v) { return add3(mat3_mul(A, v), mk3(A[9], A[10], A[11])); }
/* temp_utils.hpp:97-100 : v * rsqrt(dot(v,v)) ; rsqrt -> 1/sqrtf (IEEE) */
static inline f3 normalized3(f3 v) { float r = 1.0f / sqrtf(dot3(v, v)); return scale3(v, r); }

static inline float qnanf(void) { union { uint32_t u; float f; } c; c.u = 0x7fffffffu; return c.f; } /* temp_utils.hpp:16 */

static inline void slab_or_full(const OrcVolume *v, const OrcSlab *s, OrcSlab *out)
{
    if (s) { *out = *s; }
    else { out->z_store0 = 0; out->z_store_n = v->dims[2]; out->z_own0 = 0; out->z_own_n = v->dims[2]; }
}

/* ---------------------------------------------------------------- compute_dists
 * imgproc.cu:259-272.  finv = 1/f on the host (imgproc.cu:292).  The reference guard is
 * `x<cols || y<rows` (bug, harmless when the grid divides evenly); restated as &&.        */
ORC_API void orc_compute_dists(const uint16_t *depth, size_t depth_pitch, uint16_t *dists, size_t dists_pitch,
                               int cols, int rows, const float intr[4] /* fx fy cx cy */)
{
    const float finvx = 1.f / intr[0], finvy = 1.f / intr[1], cx = intr[2], cy = intr[3];
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const uint16_t *drow = (const uint16_t *)((const char *)depth + (size_t)y * depth_pitch);
        uint16_t *orow = (uint16_t *)((char *)dists + (size_t)y * dists_pitch);
        for (int x = 0; x < cols; ++x) {
            float xl = ((float)x - cx) * finvx;
            float yl = ((float)y - cy) * finvy;
            float lambda = sqrtf(xl * xl + yl * yl + 1);
            orow[x] = f2h((float)drow[x] * lambda * 0.001f);
        }
    }
}

/* ---------------------------------------------------------------- clear
 * tsdf_volume.cu:15-28 : every voxel <- pack_tsdf(0,0) == 0x00000000                        */
ORC_API void orc_clear(OrcVolume v, const OrcSlab *slab)
{
    OrcSlab s; slab_or_full(&v, slab, &s);
    memset(v.data, 0, (size_t)v.dims[0] * v.dims[1] * s.z_store_n * 4);
}

/* ---------------------------------------------------------------- shared TSDF update
 * tsdf_volume.cu:77-104 starting at the projection line; vc is the camera-frame position. */
static inline int tsdf_update(uint16_t *vox /* [0]=half tsdf, [1]=weight */, f3 vc, const uint16_t *dists,
                              size_t pitch, int cols, int rows, const float proj[4], float trunc,
                              float trunc_inv, int max_weight)
{
    float u = fmaf(proj[0], vc.x / vc.z, proj[2]);      /* device.hpp:35 */
    float w = fmaf(proj[1], vc.y / vc.z, proj[3]);      /* device.hpp:36 */
    /* :82 ; written so that NaN coordinates are skipped (policy, see header) */
    if (!(u >= 0 && w >= 0 && u < (float)cols && w < (float)rows)) return 0;
    const uint16_t *row = (const uint16_t *)((const char *)dists + (size_t)(int)w * pitch);
    float Dp = h2f(row[(int)u]);                         /* :85 point filter */
    if (Dp == 0 || vc.z <= 0) return 0;                  /* :86 */
    float sdf = Dp - sqrtf(dot3(vc, vc));                /* :89 */
    if (sdf >= -trunc) {                                 /* :91 */
        float tsdf = fminf(1.f, sdf * trunc_inv);        /* :93 */
        int weight_prev = vox[1];
        float tsdf_prev = h2f(vox[0]);                   /* :97 */
        float tsdf_new = fmaf(tsdf_prev, (float)weight_prev, tsdf) / (float)(weight_prev + 1); /* :99 */
        int weight_new = weight_prev + 1 < max_weight ? weight_prev + 1 : max_weight;          /* :100 */
        vox[0] = f2h(tsdf_new);
        vox[1] = (uint16_t)weight_new;                   /* :103 */
        return 1;
    }
    return 0;
}

/* ---------------------------------------------------------------- integrate (rigid)
 * tsdf_volume.cu:60-106.  vc accumulates `vc += zstep` from z = 0 for every plane, including
 * skipped ones; a slab starting at z0 replays the first z0 additions so that it is bit-identical
 * with the unsharded sweep.  Returns the number of voxels whose update branch (:91) was taken.  */
ORC_API uint64_t orc_integrate(const uint16_t *dists, size_t pitch, int cols, int rows, OrcVolume v,
                               const OrcSlab *slab, const float vol2cam[12], const float proj[4])
{
    OrcSlab s; slab_or_full(&v, slab, &s);
    const int X = v.dims[0], Y = v.dims[1];
    const float trunc_inv = 1.f / v.trunc_dist;          /* :147 */
    const f3 zstep = scale3(mk3(vol2cam[2], vol2cam[5], vol2cam[8]), v.voxel_size[2]); /* :69 */
    uint16_t *base = (uint16_t *)v.data;
    uint64_t n_upd = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_upd)
    for (int y = 0; y < Y; ++y)
        for (int x = 0; x < X; ++x) {
            f3 vx = mk3((float)x * v.voxel_size[0], (float)y * v.voxel_size[1], 0.f);   /* :71 */
            f3 vc = aff_mul(vol2cam, vx);                                                 /* :72 */
            for (int z = 0; z < s.z_own0 + s.z_own_n; ++z, vc = add3(vc, zstep)) {       /* :75 */
                if (z < s.z_own0) continue;                                               /* replay only */
                uint16_t *vox = base + 2 * ((size_t)x + (size_t)y * X + (size_t)(z - s.z_store0) * X * Y);
                n_upd += tsdf_update(vox, vc, dists, pitch, cols, rows, proj, v.trunc_dist, trunc_inv, v.max_weight);
            }
        }
    return n_upd;
}

/* ================================================================ quaternion math
 * Quaternion<float> = (w,x,y,z).  quaternion.hpp.                                           */
typedef struct { float w, x, y, z; } quat;

/* quaternion.hpp:186-194 (left-associated float sums, no fusion) */
static inline quat q_mul(quat a, quat b)
{
    quat r;
    r.w = ((a.w * b.w) - (a.x * b.x) - (a.y * b.y) - (a.z * b.z));
    r.x = ((a.w * b.x) + (a.x * b.w) + (a.y * b.z) - (a.z * b.y));
    r.y = ((a.w * b.y) - (a.x * b.z) + (a.y * b.w) + (a.z * b.x));
    r.z = ((a.w * b.z) + (a.x * b.y) - (a.y * b.x) + (a.z * b.w));
    return r;
}
static inline quat q_conj(quat a) { quat r = {a.w, -a.x, -a.y, -a.z}; return r; }       /* :206-209 */
static inline float q_norm(quat a) { return sqrtf((a.w * a.w) + (a.x * a.x) + (a.y * a.y) + (a.z * a.z)); } /* :211-214 */
/* :220-228 : (*this) = (1.0/theNorm) * (*this) ; scalar is double, product rounded per component */
static inline quat q_normalize(quat a)
{
    double inv = 1.0 / (double)q_norm(a);
    quat r = {(float)(inv * (double)a.w), (float)(inv * (double)a.x), (float)(inv * (double)a.y), (float)(inv * (double)a.z)};
    return r;
}
static inline quat q_scale_f(float s, quat a) { quat r = {s * a.w, s * a.x, s * a.y, s * a.z}; return r; } /* :172-178, U=float */
static inline quat q_add(quat a, quat b) { quat r = {a.w + b.w, a.x + b.x, a.y + b.y, a.z + b.z}; return r; } /* :138-144 */

/* dual_quaternion.hpp:120-125 : 2 * translation_ * normalize(rotation_).conjugate() */
static inline quat dq_get_translation(quat rot, quat dual)
{
    quat rn = q_normalize(rot);
    return q_mul(q_scale_f(2.f, dual), q_conj(rn));   /* int 2 * float == exact doubling */
}

ORC_API void orc_quat_mul(const float a[4], const float b[4], float out[4])
{
    quat r = q_mul(*(const quat *)a, *(const quat *)b); memcpy(out, &r, 16);
}
ORC_API void orc_quat_normalize(const float a[4], float out[4])
{
    quat r = q_normalize(*(const quat *)a); memcpy(out, &r, 16);
}
/* quaternion.hpp:75-83 encodeRotation(theta,x,y,z) with T=float (sin/cos resolve to the double
 * overloads under <cmath> only, results narrowed to float on assignment). */
ORC_API void orc_quat_encode_rotation(float theta, float x, float y, float z, float out[4])
{
    double sin_half = sin((double)(theta / 2));
    quat q; q.w = (float)cos((double)(theta / 2));
    q.x = (float)((double)x * sin_half); q.y = (float)((double)y * sin_half); q.z = (float)((double)z * sin_half);
    q = q_normalize(q); memcpy(out, &q, 16);
}
/* quaternion.hpp:108-117 rotate(x,y,z) = q * (0,x,y,z) * q.conjugate() (NOT normalised) */
ORC_API void orc_quat_rotate_xyz(const float q[4], float v[3])
{
    quat a = *(const quat *)q; quat p = {0.f, v[0], v[1], v[2]};
    quat r = q_mul(q_mul(a, p), q_conj(a)); v[0] = r.x; v[1] = r.y; v[2] = r.z;
}
/* node translation quaternion t_i (w kept: it feeds rounding of the blended dual part) */
ORC_API void orc_node_translation(const float dq[8], float out[4])
{
    quat t = dq_get_translation(*(const quat *)dq, *(const quat *)(dq + 4)); memcpy(out, &t, 16);
}
/* dual_quaternion.hpp:212-229 from_twist -> DualQuaternion(Quaternion(0,x,y,z), rotation) */
ORC_API void orc_dq_from_twist(const float r[3], const float t[3], float dq_out[8])
{
    float norm = (float)sqrt((double)(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
    quat rot;
    if (norm > 1e-6f) {
        float cosNorm = (float)cos((double)norm);
        float sign = (float)((cosNorm > 0.f) - (cosNorm < 0.f));
        cosNorm *= sign;
        float sinNorm_norm = (float)((double)sign * sin((double)norm) / (double)norm);
        rot.w = cosNorm; rot.x = r[0] * sinNorm_norm; rot.y = r[1] * sinNorm_norm; rot.z = r[2] * sinNorm_norm;
    } else { rot.w = 1; rot.x = rot.y = rot.z = 0; }
    quat tq = {0.f, t[0], t[1], t[2]};
    quat half = {0.5f * tq.w, 0.5f * tq.x, 0.5f * tq.y, 0.5f * tq.z};     /* 0.5 * translation (exact) */
    quat dual = q_mul(half, rot);                                          /* :59-63 */
    memcpy(dq_out, &rot, 16); memcpy(dq_out + 4, &dual, 16);
}

/* ================================================================ k-NN + DQB
 * Node arrays: pos[M*3], dq[M*8] = {rotation (w,x,y,z), translation_/dual (w,x,y,z)}
 * (the first 8 floats of utils::DualQuaternion<float>), sigma[M] = deformation_node::weight.  */

/* knn_point_cloud.hpp:25-31 : d0*d0+d1*d1+d2*d2 with d = query - node */
static inline float knn_dist2(const float *q, const float *p)
{
    const float d0 = q[0] - p[0], d1 = q[1] - p[1], d2 = q[2] - p[2];
    return d0 * d0 + d1 * d1 + d2 * d2;
}

/* ---------------------------------------------------------------- nanoflann v1.2.3, restated
 * kfusion/include/nanoflann/nanoflann.hpp as the reference instantiates it (warp_field.hpp:13-17, warp_field.cpp:20,23,247-251):
 * KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float, PointCloud>, PointCloud, 3>, leaf_max_size 10, KNNResultSet<float>,
 * SearchParams(10) (eps = 0).  The k nearest are unique only up to EXACT distance ties; which of two equidistant nodes is
 * returned (and blended) is decided by the order the tree visits them and by KNNResultSet::addPoint keeping the first found,
 * so the build (divideTree / middleSplit_ / planeSplit, :1042-1176) and the search (searchLevel :1200-1254, addPoint :110-131)
 * are restated statement by statement, float arithmetic included.  Pinned against the reference's own header compiled
 * unmodified (oracle/_ref, tests/test_oracle_golden.py) on random and on gridded (tie-heavy) node sets. */
#define NF_LEAF_MAX 10                                   /* KDTreeSingleIndexAdaptorParams(10), warp_field.cpp:20 */
typedef struct { int child1, child2; int left, right; int divfeat; float divlow, divhigh; } nf_node;
typedef struct { float low, high; } nf_interval;
typedef struct {
    const float *pos; int M;
    int *vind; nf_node *nodes; int n_nodes, cap_nodes;
    nf_interval root_bbox[3];
    int root;
} nf_tree;

static inline float nf_get(const nf_tree *t, int idx, int c) { return t->pos[3 * idx + c]; }       /* kdtree_get_pt */

static void nf_min_max(const nf_tree *t, const int *ind, int count, int element, float *min_elem, float *max_elem)   /* :1094-1103 */
{
    *min_elem = nf_get(t, ind[0], element);
    *max_elem = nf_get(t, ind[0], element);
    for (int i = 1; i < count; ++i) {
        float val = nf_get(t, ind[i], element);
        if (val < *min_elem) *min_elem = val;
        if (val > *max_elem) *max_elem = val;
    }
}

/* :1151-1176 (IndexType is size_t there: the `right &&` / `!right` tests keep the unsigned index from wrapping) */
static void nf_plane_split(const nf_tree *t, int *ind, int count, int cutfeat, float cutval, int *lim1, int *lim2)
{
    size_t left = 0, right = (size_t)count - 1;
    for (;;) {
        while (left <= right && nf_get(t, ind[left], cutfeat) < cutval) ++left;
        while (right && left <= right && nf_get(t, ind[right], cutfeat) >= cutval) --right;
        if (left > right || !right) break;
        int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim1 = (int)left;
    right = (size_t)count - 1;
    for (;;) {
        while (left <= right && nf_get(t, ind[left], cutfeat) <= cutval) ++left;
        while (right && left <= right && nf_get(t, ind[right], cutfeat) > cutval) --right;
        if (left > right || !right) break;
        int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp;
        ++left; --right;
    }
    *lim2 = (int)left;
}

static void nf_middle_split(const nf_tree *t, int *ind, int count, int *index, int *cutfeat, float *cutval, const nf_interval *bbox)   /* :1105-1140 */
{
    const float EPS = 0.00001f;
    float max_span = bbox[0].high - bbox[0].low;
    for (int i = 1; i < 3; ++i) {
        float span = bbox[i].high - bbox[i].low;
        if (span > max_span) max_span = span;
    }
    float max_spread = -1;
    *cutfeat = 0;
    for (int i = 0; i < 3; ++i) {
        float span = bbox[i].high - bbox[i].low;
        if (span > (1 - EPS) * max_span) {
            float min_elem, max_elem;
            nf_min_max(t, ind, count, i, &min_elem, &max_elem);
            float spread = max_elem - min_elem;
            if (spread > max_spread) { *cutfeat = i; max_spread = spread; }
        }
    }
    float split_val = (bbox[*cutfeat].low + bbox[*cutfeat].high) / 2;
    float min_elem, max_elem;
    nf_min_max(t, ind, count, *cutfeat, &min_elem, &max_elem);
    if (split_val < min_elem) *cutval = min_elem;
    else if (split_val > max_elem) *cutval = max_elem;
    else *cutval = split_val;
    int lim1, lim2;
    nf_plane_split(t, ind, count, *cutfeat, *cutval, &lim1, &lim2);
    if (lim1 > count / 2) *index = lim1;
    else if (lim2 < count / 2) *index = lim2;
    else *index = count / 2;
}

static int nf_divide(nf_tree *t, int left, int right, nf_interval *bbox)                          /* divideTree :1042-1091 */
{
    int me = t->n_nodes++;
    nf_node *node = &t->nodes[me];
    if ((right - left) <= NF_LEAF_MAX) {
        node->child1 = node->child2 = -1;
        node->left = left; node->right = right;
        for (int i = 0; i < 3; ++i) { bbox[i].low = nf_get(t, t->vind[left], i); bbox[i].high = nf_get(t, t->vind[left], i); }
        for (int k = left + 1; k < right; ++k)
            for (int i = 0; i < 3; ++i) {
                if (bbox[i].low > nf_get(t, t->vind[k], i)) bbox[i].low = nf_get(t, t->vind[k], i);
                if (bbox[i].high < nf_get(t, t->vind[k], i)) bbox[i].high = nf_get(t, t->vind[k], i);
            }
    } else {
        int idx, cutfeat; float cutval;
        nf_middle_split(t, t->vind + left, right - left, &idx, &cutfeat, &cutval, bbox);
        node->divfeat = cutfeat;
        nf_interval left_bbox[3], right_bbox[3];
        memcpy(left_bbox, bbox, sizeof(left_bbox));
        left_bbox[cutfeat].high = cutval;
        int c1 = nf_divide(t, left, left + idx, left_bbox);
        memcpy(right_bbox, bbox, sizeof(right_bbox));
        right_bbox[cutfeat].low = cutval;
        int c2 = nf_divide(t, left + idx, right, right_bbox);
        node = &t->nodes[me];
        node->child1 = c1; node->child2 = c2;
        node->divlow = left_bbox[cutfeat].high;
        node->divhigh = right_bbox[cutfeat].low;
        for (int i = 0; i < 3; ++i) {
            bbox[i].low = right_bbox[i].low < left_bbox[i].low ? right_bbox[i].low : left_bbox[i].low;      /* std::min(a,b): b<a ? b : a */
            bbox[i].high = left_bbox[i].high < right_bbox[i].high ? right_bbox[i].high : left_bbox[i].high; /* std::max(a,b): a<b ? b : a */
        }
    }
    return me;
}

static nf_tree *nf_build(const float *pos, int M)                                                  /* buildIndex :855-866 */
{
    nf_tree *t = (nf_tree *)calloc(1, sizeof(nf_tree));
    t->pos = pos; t->M = M;
    t->vind = (int *)malloc(sizeof(int) * (size_t)(M > 0 ? M : 1));
    for (int i = 0; i < M; ++i) t->vind[i] = i;                                                    /* init_vind */
    t->cap_nodes = 2 * M + 2;
    t->nodes = (nf_node *)calloc((size_t)t->cap_nodes, sizeof(nf_node));
    t->root = -1;
    if (M == 0) return t;
    for (int i = 0; i < 3; ++i) t->root_bbox[i].low = t->root_bbox[i].high = nf_get(t, 0, i);      /* computeBoundingBox :1010-1032 */
    for (int k = 1; k < M; ++k)
        for (int i = 0; i < 3; ++i) {
            if (nf_get(t, k, i) < t->root_bbox[i].low) t->root_bbox[i].low = nf_get(t, k, i);
            if (nf_get(t, k, i) > t->root_bbox[i].high) t->root_bbox[i].high = nf_get(t, k, i);
        }
    t->root = nf_divide(t, 0, M, t->root_bbox);
    return t;
}
static void nf_free(nf_tree *t) { if (t) { free(t->vind); free(t->nodes); free(t); } }

typedef struct { int *indices; float *dists; int capacity, count; } nf_result;                      /* KNNResultSet :78-137 */
static inline void nf_add_point(nf_result *r, float dist, int index)                               /* addPoint :110-131 */
{
    int i;
    for (i = r->count; i > 0; --i) {
        if (r->dists[i - 1] > dist) {
            if (i < r->capacity) { r->dists[i] = r->dists[i - 1]; r->indices[i] = r->indices[i - 1]; }
        } else break;
    }
    if (i < r->capacity) { r->dists[i] = dist; r->indices[i] = index; }
    if (r->count < r->capacity) r->count++;
}

static void nf_search_level(const nf_tree *t, nf_result *rs, const float *vec, int node_i, float mindistsq, float *dists)   /* :1200-1254 */
{
    const nf_node *node = &t->nodes[node_i];
    if (node->child1 < 0 && node->child2 < 0) {
        float worst_dist = rs->dists[rs->capacity - 1];                                             /* worstDist(), read once per leaf */
        for (int i = node->left; i < node->right; ++i) {
            const int index = t->vind[i];
            float dist = knn_dist2(vec, t->pos + 3 * index);
            if (dist < worst_dist) nf_add_point(rs, dist, index);
        }
        return;
    }
    int idx = node->divfeat;
    float val = vec[idx];
    float diff1 = val - node->divlow;
    float diff2 = val - node->divhigh;
    int best, other; float cut_dist;
    if ((diff1 + diff2) < 0) { best = node->child1; other = node->child2; cut_dist = (val - node->divhigh) * (val - node->divhigh); }
    else { best = node->child2; other = node->child1; cut_dist = (val - node->divlow) * (val - node->divlow); }
    nf_search_level(t, rs, vec, best, mindistsq, dists);
    float dst = dists[idx];
    mindistsq = mindistsq + cut_dist - dst;
    dists[idx] = cut_dist;
    if (mindistsq * 1.0f <= rs->dists[rs->capacity - 1]) nf_search_level(t, rs, vec, other, mindistsq, dists);   /* epsError = 1 + 0 */
    dists[idx] = dst;
}

/* findNeighbors :903-917 after resultSet.init (:92-99).  M >= k is a precondition. */
static inline void knn_tree(const nf_tree *t, const float q[3], int k, int *idx, float *d2)
{
    nf_result rs = {idx, d2, k, 0};
    d2[k - 1] = 3.402823466e+38f;                                                                   /* numeric_limits<float>::max() */
    if (t->M == 0) return;
    float dists[3] = {0.f, 0.f, 0.f};
    float distsq = 0.f;
    for (int i = 0; i < 3; ++i) {                                                                   /* computeInitialDistances :1179-1196 */
        if (q[i] < t->root_bbox[i].low) { dists[i] = (q[i] - t->root_bbox[i].low) * (q[i] - t->root_bbox[i].low); distsq += dists[i]; }
        if (q[i] > t->root_bbox[i].high) { dists[i] = (q[i] - t->root_bbox[i].high) * (q[i] - t->root_bbox[i].high); distsq += dists[i]; }
    }
    nf_search_level(t, &rs, q, t->root, distsq, dists);
}

/* exhaustive scan in node-index order, strict '<' insertion (ties -> lower index).  Kept only as an independent check of the
 * tree search on tie-free data (tests) -- the oracle's answers come from knn_tree. */
static inline void knn_brute(const float *pos, int M, const float q[3], int k, int *idx, float *d2)
{
    int count = 0;
    for (int j = 0; j < M; ++j) {
        float d = knn_dist2(q, pos + 3 * j);
        if (count == k && !(d < d2[k - 1])) continue;
        int i = count < k ? count : k - 1;
        while (i > 0 && d2[i - 1] > d) { d2[i] = d2[i - 1]; idx[i] = idx[i - 1]; --i; }
        d2[i] = d; idx[i] = j;
        if (count < k) ++count;
    }
}

ORC_API void orc_knn(const float *pos, int M, const float *queries, int N, int k, int *idx_out, float *d2_out)
{
    nf_tree *t = nf_build(pos, M);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) knn_tree(t, queries + 3 * i, k, idx_out + (size_t)i * k, d2_out + (size_t)i * k);
    nf_free(t);
}
ORC_API void orc_knn_brute(const float *pos, int M, const float *queries, int N, int k, int *idx_out, float *d2_out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) knn_brute(pos, M, queries + 3 * i, k, idx_out + (size_t)i * k, d2_out + (size_t)i * k);
}
/* tree shape for tests: depth, leaf count, and vind (the permutation the build leaves behind) */
ORC_API void orc_nanoflann_tree_info(const float *pos, int M, int *vind_out, int *n_nodes, int *depth_out)
{
    nf_tree *t = nf_build(pos, M);
    if (vind_out) memcpy(vind_out, t->vind, sizeof(int) * (size_t)M);
    if (n_nodes) *n_nodes = t->n_nodes;
    if (depth_out) {
        int best = 0;
        int *stack = (int *)malloc(sizeof(int) * 2 * (size_t)(t->n_nodes + 1)); int sp = 0;
        if (t->root >= 0) { stack[sp++] = t->root; stack[sp++] = 1; }
        while (sp) { int d = stack[--sp], n = stack[--sp]; if (d > best) best = d;
            if (t->nodes[n].child1 >= 0) { stack[sp++] = t->nodes[n].child1; stack[sp++] = d + 1; stack[sp++] = t->nodes[n].child2; stack[sp++] = d + 1; } }
        free(stack); *depth_out = best;
    }
    nf_free(t);
}

/* warp_field.cpp:238-241 */
static inline float dqb_weight(float d2, float sigma) { return (float)exp((double)(-d2 / (2 * sigma * sigma))); }

/* warp_field.cpp:203-217 + dual_quaternion.hpp:59-63.  node_t = per-node getTranslation().   */
static inline void dqb_blend(const float *dq, const float *node_t, const float *sigma, int k, const int *idx,
                             const float *d2, quat *rot_out, quat *dual_out)
{
    quat tsum = {0, 0, 0, 0}, rsum = {0, 0, 0, 0};
    for (int i = 0; i < k; ++i) {
        int j = idx[i];
        float w = dqb_weight(d2[i], sigma[j]);
        tsum = q_add(tsum, q_scale_f(w, *(const quat *)(node_t + 4 * j)));   /* :211 */
        rsum = q_add(rsum, q_scale_f(w, *(const quat *)(dq + 8 * j)));       /* :212 */
    }
    rsum = q_normalize(rsum);                                                 /* :214 */
    quat half = {0.5f * tsum.w, 0.5f * tsum.x, 0.5f * tsum.y, 0.5f * tsum.z};
    *rot_out = rsum;
    *dual_out = q_mul(half, rsum);                                            /* dq ctor :59-63 */
}

/* dual_quaternion.hpp:204-210 transform ; quaternion.hpp:124-130 rotate(Vec3f&) with cv::Vec3f
 * semantics: cross = (a1*b2-a2*b1, a2*b0-a0*b2, a0*b1-a1*b0), component-wise float ops.      */
static inline f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline f3 dq_transform(quat rot, quat dual, f3 p)
{
    quat t = dq_get_translation(rot, dual);                    /* :206-207 */
    quat rn = q_normalize(rot);                                /* rotate(): rot.normalize() */
    f3 qv = mk3(rn.x, rn.y, rn.z);
    f3 inner = add3(cross3(qv, p), scale3(p, rn.w));           /* q_vec.cross(v) + v*rot.w_ */
    p = add3(p, cross3(scale3(qv, 2.f), inner));               /* v += (q_vec*2.f).cross(...) */
    return add3(p, mk3(t.x, t.y, t.z));                        /* point += translation */
}

static void node_translations(const float *dq, int M, float *node_t)
{
    for (int j = 0; j < M; ++j) orc_node_translation(dq + 8 * j, node_t + 4 * j);
}

/* DQB-warp one point (warp_field.cpp:187-188): returns DQB(p).transform(p) */
static inline f3 warp_point(const nf_tree *tree, const float *dq, const float *node_t, const float *sigma, int k, f3 p)
{
    int idx[16]; float d2[16]; float q[3] = {p.x, p.y, p.z};
    knn_tree(tree, q, k, idx, d2);
    quat rot, dual; dqb_blend(dq, node_t, sigma, k, idx, d2, &rot, &dual);
    return dq_transform(rot, dual, p);
}

/* cv::Affine3f * Vec3f : m0*x + m1*y + m2*z + m3, left-associated floats (opencv affine.hpp) */
static inline f3 cv_affine_mul(const float *A /* R[9], t[3] */, f3 v)
{
    return mk3(A[0] * v.x + A[1] * v.y + A[2] * v.z + A[9], A[3] * v.x + A[4] * v.y + A[5] * v.z + A[10],
               A[6] * v.x + A[7] * v.y + A[8] * v.z + A[11]);
}

/* warp_field.cpp:180-195 WarpField::warp.  points/normals are N x 3 floats.  The reference's
 * `i++`-skipped-on-NaN index drift is FIXED (index by position); normals are
 * transformed like points (translation included), as the reference does.                     */
ORC_API void orc_warp_points(const float *pos, const float *dq, const float *sigma, int M, int k, float *points,
                             float *normals, int N, const float warp_to_live[12])
{
    float *node_t = (float *)malloc((size_t)M * 16);
    node_translations(dq, M, node_t);
    nf_tree *tree = nf_build(pos, M);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        float *p = points + 3 * (size_t)i;
        float *n = normals ? normals + 3 * (size_t)i : 0;
        if (isnan(p[0]) || (n && isnan(n[0]))) continue;
        int idx[16]; float d2[16];
        knn_tree(tree, p, k, idx, d2);
        quat rot, dual; dqb_blend(dq, node_t, sigma, k, idx, d2, &rot, &dual);
        f3 pw = cv_affine_mul(warp_to_live, dq_transform(rot, dual, mk3(p[0], p[1], p[2])));
        p[0] = pw.x; p[1] = pw.y; p[2] = pw.z;
        if (n) {
            f3 nw = cv_affine_mul(warp_to_live, dq_transform(rot, dual, mk3(n[0], n[1], n[2])));
            n[0] = nw.x; n[1] = nw.y; n[2] = nw.z;
        }
    }
    nf_free(tree);
    free(node_t);
}

/* DQB only (for direct parity tests): out_dq[N*8] = {rot, dual} of DQB(p) */
ORC_API void orc_dqb(const float *pos, const float *dq, const float *sigma, int M, int k, const float *points, int N,
                     float *out_dq)
{
    float *node_t = (float *)malloc((size_t)M * 16);
    node_translations(dq, M, node_t);
    nf_tree *tree = nf_build(pos, M);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        int idx[16]; float d2[16];
        knn_tree(tree, points + 3 * (size_t)i, k, idx, d2);
        quat rot, dual; dqb_blend(dq, node_t, sigma, k, idx, d2, &rot, &dual);
        memcpy(out_dq + 8 * (size_t)i, &rot, 16); memcpy(out_dq + 8 * (size_t)i + 4, &dual, 16);
    }
    nf_free(tree);
    free(node_t);
}

/* ---------------------------------------------------------------- integrate (warped)
 * x_c = vol2world*(i*vsx, j*vsy, k*vsz) ; x_w = DQB(x_c).transform(x_c) ;
 * vc = world2cam * x_w ; continue at tsdf_volume.cu:77.  No incremental zstep.              */
ORC_API uint64_t orc_integrate_warped(const uint16_t *dists, size_t pitch, int cols, int rows, OrcVolume v,
                                      const OrcSlab *slab, const float vol2world[12], const float world2cam[12],
                                      const float proj[4], const float *pos, const float *dq, const float *sigma,
                                      int M, int k)
{
    OrcSlab s; slab_or_full(&v, slab, &s);
    const int X = v.dims[0], Y = v.dims[1];
    const float trunc_inv = 1.f / v.trunc_dist;
    uint16_t *base = (uint16_t *)v.data;
    float *node_t = (float *)malloc((size_t)M * 16);
    node_translations(dq, M, node_t);
    nf_tree *tree = nf_build(pos, M);
    uint64_t n_upd = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : n_upd)
    for (int z = s.z_own0; z < s.z_own0 + s.z_own_n; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                f3 vx = mk3((float)x * v.voxel_size[0], (float)y * v.voxel_size[1], (float)z * v.voxel_size[2]);
                f3 xc = aff_mul(vol2world, vx);
                f3 xw = warp_point(tree, dq, node_t, sigma, k, xc);
                f3 vc = aff_mul(world2cam, xw);
                uint16_t *vox = base + 2 * ((size_t)x + (size_t)y * X + (size_t)(z - s.z_store0) * X * Y);
                n_upd += tsdf_update(vox, vc, dists, pitch, cols, rows, proj, v.trunc_dist, trunc_inv, v.max_weight);
            }
    nf_free(tree);
    free(node_t);
    return n_upd;
}

/* ================================================================ raycast
 * tsdf_volume.cu:202-474                                                                     */
typedef struct {
    const uint16_t *data; int X, Y, Z; OrcSlab s;
    f3 vs, vsi, gd, volume_size; float time_step;
} rc_ctx;

static inline float vox_tsdf(const rc_ctx *c, int x, int y, int z)
{   /* device.hpp:17-18 index ; clamped to the stored range (policy) */
    x = x < 0 ? 0 : (x >= c->X ? c->X - 1 : x);
    y = y < 0 ? 0 : (y >= c->Y ? c->Y - 1 : y);
    int zl = z - c->s.z_store0; zl = zl < 0 ? 0 : (zl >= c->s.z_store_n ? c->s.z_store_n - 1 : zl);
    return h2f(c->data[2 * ((size_t)x + (size_t)y * c->X + (size_t)zl * c->X * c->Y)]);
}
/* :262-270 */
static inline float fetch_tsdf(const rc_ctx *c, f3 p, int *zi)
{
    int x = (int)lrintf(p.x * c->vsi.x), y = (int)lrintf(p.y * c->vsi.y), z = (int)lrintf(p.z * c->vsi.z);
    if (zi) *zi = z;
    return vox_tsdf(c, x, y, z);
}
/* :220-245 */
static inline float interpolate(const rc_ctx *c, f3 cf)
{
    int gx = (int)floorf(cf.x), gy = (int)floorf(cf.y), gz = (int)floorf(cf.z);
    if (gx < 0 || gx >= c->X - 1 || gy < 0 || gy >= c->Y - 1 || gz < 0 || gz >= c->Z - 1) return qnanf();
    float a = cf.x - gx, b = cf.y - gy, cc = cf.z - gz;
    float t = 0.f;
    t += vox_tsdf(c, gx + 0, gy + 0, gz + 0) * (1 - a) * (1 - b) * (1 - cc);
    t += vox_tsdf(c, gx + 0, gy + 0, gz + 1) * (1 - a) * (1 - b) * cc;
    t += vox_tsdf(c, gx + 0, gy + 1, gz + 0) * (1 - a) * b * (1 - cc);
    t += vox_tsdf(c, gx + 0, gy + 1, gz + 1) * (1 - a) * b * cc;
    t += vox_tsdf(c, gx + 1, gy + 0, gz + 0) * a * (1 - b) * (1 - cc);
    t += vox_tsdf(c, gx + 1, gy + 0, gz + 1) * a * (1 - b) * cc;
    t += vox_tsdf(c, gx + 1, gy + 1, gz + 0) * a * b * (1 - cc);
    t += vox_tsdf(c, gx + 1, gy + 1, gz + 1) * a * b * cc;
    return t;
}
/* :408-426 */
static inline f3 compute_normal(const rc_ctx *c, f3 p)
{
    f3 n;
    float Fx1 = interpolate(c, mul3(mk3(p.x + c->gd.x, p.y, p.z), c->vsi));
    float Fx2 = interpolate(c, mul3(mk3(p.x - c->gd.x, p.y, p.z), c->vsi));
    n.x = (Fx1 - Fx2) / c->gd.x;
    float Fy1 = interpolate(c, mul3(mk3(p.x, p.y + c->gd.y, p.z), c->vsi));
    float Fy2 = interpolate(c, mul3(mk3(p.x, p.y - c->gd.y, p.z), c->vsi));
    n.y = (Fy1 - Fy2) / c->gd.y;
    float Fz1 = interpolate(c, mul3(mk3(p.x, p.y, p.z + c->gd.z), c->vsi));
    float Fz2 = interpolate(c, mul3(mk3(p.x, p.y, p.z - c->gd.z), c->vsi));
    n.z = (Fz1 - Fz2) / c->gd.z;
    return normalized3(n);
}
/* :202-218 */
static inline void intersect(f3 org, f3 dir, f3 box_max, float *tnear, float *tfar)
{
    f3 invR = mk3(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
    f3 tbot = mul3(invR, sub3(mk3(0.f, 0.f, 0.f), org));
    f3 ttop = mul3(invR, sub3(box_max, org));
    f3 tmin = mk3(fminf(ttop.x, tbot.x), fminf(ttop.y, tbot.y), fminf(ttop.z, tbot.z));
    f3 tmax = mk3(fmaxf(ttop.x, tbot.x), fmaxf(ttop.y, tbot.y), fmaxf(ttop.z, tbot.z));
    *tnear = fmaxf(fmaxf(tmin.x, tmin.y), fmaxf(tmin.x, tmin.z));
    *tfar = fminf(fminf(tmax.x, tmax.y), fminf(tmax.x, tmax.z));
}

#define ORC_RC_NO_EVENT 0xffffffffu
typedef struct { uint32_t key; int hit; float t_hit; f3 p_curr, p_next, org, dir; uint32_t n_steps; } rc_hit;

/* One ray, stage 1 (march): the event key is (step index k << 1) | kind, kind 1 = +->- hit, 0 = -->+ break;
 * ORC_RC_NO_EVENT if no event on a step this slab owns.  A step is owned when the nearest-voxel plane of its
 * `curr` sample lies in [z_own0, z_own0+z_own_n).                                                      */
static inline rc_hit ray_march(const rc_ctx *c, const float aff[12], const float reproj[4], int x, int y)
{
    rc_hit h; h.key = ORC_RC_NO_EVENT; h.hit = 0; h.t_hit = 0.f; h.n_steps = 0;
    h.org = mk3(aff[9], aff[10], aff[11]);
    /* device.hpp:43-48 : x = z*(u-cx)*finvx with z = 1.f */
    f3 rp = mk3(1.f * ((float)x - reproj[2]) * reproj[0], 1.f * ((float)y - reproj[3]) * reproj[1], 1.f);
    h.dir = normalized3(mat3_mul(aff, rp));                                    /* :354 */
    h.p_curr = h.org; h.p_next = h.org;
    const f3 org = h.org, dir = h.dir;
    f3 box_max = sub3(c->volume_size, c->vs);                                  /* :359 */
    float tmin, tmax; intersect(org, dir, box_max, &tmin, &tmax);
    tmin = fmaxf(0.f, tmin);                                                   /* :364-365 */
    if (!(tmin < tmax)) return h;                                              /* :366 */
    tmax -= c->time_step;                                                      /* :369 */
    f3 vstep = scale3(dir, c->time_step);
    f3 next = add3(org, scale3(dir, tmin));
    int zn; float tsdf_next = fetch_tsdf(c, next, &zn);                        /* :373 */
    uint32_t k = 0;
    for (float tcurr = tmin; tcurr < tmax; tcurr += c->time_step, ++k) {       /* :374 */
        float tsdf_curr = tsdf_next; f3 curr = next; int zc = zn;
        next = add3(next, vstep);
        tsdf_next = fetch_tsdf(c, next, &zn);                                  /* :380 */
        if (zc < c->s.z_own0 || zc >= c->s.z_own0 + c->s.z_own_n) continue;   /* not this slab's step */
        ++h.n_steps;
        if (tsdf_curr < 0.f && tsdf_next > 0.f) { h.key = (k << 1) | 0u; return h; }   /* :381 */
        if (tsdf_curr > 0.f && tsdf_next < 0.f) {                              /* :384 */
            h.key = (k << 1) | 1u; h.hit = 1; h.t_hit = tcurr; h.p_curr = curr; h.p_next = next;
            return h;
        }
    }
    return h;
}
/* stage 2 (locate), :386-391 : the refined ray parameter Ts and the vertex org + dir * Ts in the volume frame.  Ts may
 * extrapolate far beyond [curr, next]. */
static inline float ray_locate_ts(const rc_ctx *c, const rc_hit *h)
{
    float Ft = interpolate(c, mul3(h->p_curr, c->vsi));
    float Ftdt = interpolate(c, mul3(h->p_next, c->vsi));
    return h->t_hit - (c->time_step * Ft) / (Ftdt - Ft);                       /* :389 */
}
static inline f3 ray_vertex(f3 org, f3 dir, float Ts) { return add3(org, scale3(dir, Ts)); }   /* :390 */
static inline f3 ray_locate(const rc_ctx *c, const rc_hit *h) { return ray_vertex(h->org, h->dir, ray_locate_ts(c, h)); }
/* the ray of pixel (x, y): :353-354 (the first lines of ray_march) */
static inline void ray_of_pixel(const float aff[12], const float reproj[4], int x, int y, f3 *org, f3 *dir)
{
    *org = mk3(aff[9], aff[10], aff[11]);
    f3 rp = mk3(1.f * ((float)x - reproj[2]) * reproj[0], 1.f * ((float)y - reproj[3]) * reproj[1], 1.f);
    *dir = normalized3(mat3_mul(aff, rp));
}
/* stage 3 (shade), :392-401 : normal at the vertex; camera-frame outputs if the normal is finite. */
static inline int ray_shade(const rc_ctx *c, const float aff[12], const float Rinv[9], f3 vertex, f3 *vertex_out, f3 *normal_out)
{
    f3 normal = compute_normal(c, vertex);
    if (isnan(normal.x * normal.y * normal.z)) return 0;                       /* :394 */
    *normal_out = mat3_mul(Rinv, normal);
    *vertex_out = mat3_mul(Rinv, sub3(vertex, mk3(aff[9], aff[10], aff[11])));
    return 1;
}
static inline uint32_t cast_ray(const rc_ctx *c, const float aff[12], const float Rinv[9], const float reproj[4],
                                int x, int y, f3 *vertex_out, f3 *normal_out, int *valid, uint32_t *n_steps)
{
    rc_hit h = ray_march(c, aff, reproj, x, y);
    *valid = 0;
    if (n_steps) *n_steps = h.n_steps;
    if (h.hit) *valid = ray_shade(c, aff, Rinv, ray_locate(c, &h), vertex_out, normal_out);
    return h.key;
}

static void rc_setup(rc_ctx *c, const OrcVolume *v, const OrcSlab *slab, float step_factor, float delta_factor)
{
    c->data = (const uint16_t *)v->data; c->X = v->dims[0]; c->Y = v->dims[1]; c->Z = v->dims[2];
    slab_or_full(v, slab, &c->s);
    c->vs = mk3(v->voxel_size[0], v->voxel_size[1], v->voxel_size[2]);
    c->volume_size = mk3(c->vs.x * (float)c->X, c->vs.y * (float)c->Y, c->vs.z * (float)c->Z);   /* :464 */
    c->time_step = v->trunc_dist * step_factor;                                                  /* :465 */
    c->gd = scale3(c->vs, delta_factor);                                                         /* :466 */
    c->vsi = mk3(1.f / c->vs.x, 1.f / c->vs.y, 1.f / c->vs.z);                                   /* :467 */
}

/* Points variant, :340-405.  points/normals are float4 rows with byte pitches; misses = all-NaN.
 * keys (optional, cols*rows uint32) receives the per-pixel event key for sharded merging.
 * stats (optional): [0] = sum of owned steps, [1] = number of valid hits.                     */
ORC_API void orc_raycast_points(OrcVolume v, const OrcSlab *slab, const float cam2vol[12], const float Rinv[9],
                                const float reproj[4] /* finvx finvy cx cy */, float *points, size_t ppitch,
                                float *normals, size_t npitch, int cols, int rows, float step_factor,
                                float delta_factor, uint32_t *keys, uint64_t *stats)
{
    rc_ctx c; rc_setup(&c, &v, slab, step_factor, delta_factor);
    uint64_t steps = 0, hits = 0;
    const float qn = qnanf();
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : steps, hits)
    for (int y = 0; y < rows; ++y) {
        float *prow = (float *)((char *)points + (size_t)y * ppitch);
        float *nrow = (float *)((char *)normals + (size_t)y * npitch);
        for (int x = 0; x < cols; ++x) {
            for (int i = 0; i < 4; ++i) { prow[4 * x + i] = qn; nrow[4 * x + i] = qn; }   /* :351 */
            f3 vtx, nrm; int valid; uint32_t ns;
            uint32_t key = cast_ray(&c, cam2vol, Rinv, reproj, x, y, &vtx, &nrm, &valid, &ns);
            if (keys) keys[(size_t)y * cols + x] = key;
            steps += ns;
            if (valid) {
                ++hits;
                nrow[4 * x] = nrm.x; nrow[4 * x + 1] = nrm.y; nrow[4 * x + 2] = nrm.z; nrow[4 * x + 3] = 0.f;
                prow[4 * x] = vtx.x; prow[4 * x + 1] = vtx.y; prow[4 * x + 2] = vtx.z; prow[4 * x + 3] = 0.f;
            }
        }
    }
    if (stats) { stats[0] = steps; stats[1] = hits; }
}

/* Depth variant, :272-338 : depth = (ushort)(vertex.z*1000), zero-filled; normals NaN-filled. */
ORC_API void orc_raycast_depth(OrcVolume v, const OrcSlab *slab, const float cam2vol[12], const float Rinv[9],
                               const float reproj[4], uint16_t *depth, size_t dpitch, float *normals, size_t npitch,
                               int cols, int rows, float step_factor, float delta_factor)
{
    rc_ctx c; rc_setup(&c, &v, slab, step_factor, delta_factor);
    const float qn = qnanf();
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < rows; ++y) {
        uint16_t *drow = (uint16_t *)((char *)depth + (size_t)y * dpitch);
        float *nrow = (float *)((char *)normals + (size_t)y * npitch);
        for (int x = 0; x < cols; ++x) {
            drow[x] = 0;                                                                  /* :283 */
            for (int i = 0; i < 4; ++i) nrow[4 * x + i] = qn;
            f3 vtx, nrm; int valid;
            cast_ray(&c, cam2vol, Rinv, reproj, x, y, &vtx, &nrm, &valid, 0);
            if (valid) {
                nrow[4 * x] = nrm.x; nrow[4 * x + 1] = nrm.y; nrow[4 * x + 2] = nrm.z; nrow[4 * x + 3] = 0.f;
                float mm = vtx.z * 1000;                                                  /* :333 */
                drow[x] = (uint16_t)(mm <= 0.f ? 0 : (mm >= 65535.f ? 65535 : (int)mm));
            }
        }
    }
}

/* ---- Z-slab (multi-GPU) cast in two stages, mirroring dfusion_raycast_march / dfusion_raycast_shade.
 * march: keys[cols*rows] (event key, ORC_RC_NO_EVENT = none) + ts[cols*rows]: the refined ray parameter Ts of a hit (:389),
 * 0 otherwise.  The vertex of a hit is org + dir * Ts -- recomputable by anyone who knows the pixel and Ts, which is what lets the
 * sharded merge carry Ts inside the key instead of exchanging vertices.                                                        */
ORC_API void orc_raycast_march(OrcVolume v, const OrcSlab *slab, const float cam2vol[12], const float reproj[4],
                               int cols, int rows, float step_factor, uint32_t *keys, float *ts)
{
    rc_ctx c; rc_setup(&c, &v, slab, step_factor, 0.5f);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            rc_hit h = ray_march(&c, cam2vol, reproj, x, y);
            ts[(size_t)y * cols + x] = h.hit ? ray_locate_ts(&c, &h) : 0.f;
            keys[(size_t)y * cols + x] = h.key;
        }
}
/* shade: given the MERGED keys and the winners' Ts, the slab owning the vertex' nearest plane writes the
 * final point/normal (NaN if the normal is not finite); the slab owning plane 0 writes the NaN fill of misses;
 * everything else is all-zero bits so that integer-summing the slabs' outputs reproduces the unsharded cast. */
ORC_API void orc_raycast_shade(OrcVolume v, const OrcSlab *slab, const float cam2vol[12], const float Rinv[9], const float reproj[4],
                               const float *ts, const uint32_t *merged_keys, float *points, size_t ppitch,
                               float *normals, size_t npitch, int cols, int rows, float delta_factor)
{
    rc_ctx c; rc_setup(&c, &v, slab, 0.75f, delta_factor);
    const float qn = qnanf();
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < rows; ++y) {
        float *prow = (float *)((char *)points + (size_t)y * ppitch);
        float *nrow = (float *)((char *)normals + (size_t)y * npitch);
        for (int x = 0; x < cols; ++x) {
            uint32_t key = merged_keys[(size_t)y * cols + x];
            float fill = 0.f;
            int resolved = 0; f3 vtx, nrm;
            if (key != ORC_RC_NO_EVENT && (key & 1u)) {
                f3 org, dir; ray_of_pixel(cam2vol, reproj, x, y, &org, &dir);
                const f3 vv = ray_vertex(org, dir, ts[(size_t)y * cols + x]);
                float zf = rintf(vv.z * c.vsi.z);
                int pz = (zf == zf) ? (int)fminf(fmaxf(zf, 0.f), (float)(c.Z - 1)) : 0;
                if (pz >= c.s.z_own0 && pz < c.s.z_own0 + c.s.z_own_n) {
                    fill = qn;
                    resolved = ray_shade(&c, cam2vol, Rinv, vv, &vtx, &nrm);
                }
            } else if (c.s.z_own0 == 0) fill = qn;
            for (int i = 0; i < 4; ++i) { prow[4 * x + i] = fill; nrow[4 * x + i] = fill; }
            if (resolved) {
                nrow[4 * x] = nrm.x; nrow[4 * x + 1] = nrm.y; nrow[4 * x + 2] = nrm.z; nrow[4 * x + 3] = 0.f;
                prow[4 * x] = vtx.x; prow[4 * x + 1] = vtx.y; prow[4 * x + 2] = vtx.z; prow[4 * x + 3] = 0.f;
            }
        }
    }
}

/* Stage 3 of the sharded cast (dfusion_raycast_points_of_keys): the camera-frame points from the merged events, their Ts and the
 * summed normals -- Rinv * (vertex - origin) as ray_shade writes it (:396), for the hits whose normal stands (4th component 0;
 * the NaN fill has a NaN there).  No volume involved.                                                                            */
ORC_API void orc_raycast_points_of_keys(const float cam2vol[12], const float Rinv[9], const float reproj[4], const float *ts,
                                        const uint32_t *merged_keys, const float *normals, size_t npitch, float *points, size_t ppitch,
                                        int cols, int rows)
{
    const float qn = qnanf();
    for (int y = 0; y < rows; ++y) {
        float *prow = (float *)((char *)points + (size_t)y * ppitch);
        const float *nrow = (const float *)((const char *)normals + (size_t)y * npitch);
        for (int x = 0; x < cols; ++x) {
            const uint32_t key = merged_keys[(size_t)y * cols + x];
            for (int i = 0; i < 4; ++i) prow[4 * x + i] = qn;
            if (key != ORC_RC_NO_EVENT && (key & 1u) && nrow[4 * x + 3] == nrow[4 * x + 3]) {
                f3 org, dir; ray_of_pixel(cam2vol, reproj, x, y, &org, &dir);
                const f3 vv = ray_vertex(org, dir, ts[(size_t)y * cols + x]);
                const f3 v = mat3_mul(Rinv, sub3(vv, mk3(cam2vol[9], cam2vol[10], cam2vol[11])));
                prow[4 * x] = v.x; prow[4 * x + 1] = v.y; prow[4 * x + 2] = v.z; prow[4 * x + 3] = 0.f;
            }
        }
    }
}

/* ================================================================ cloud / normal extraction
 * tsdf_volume.cu:511-710 FullScan6: every voxel with W != 0 && F != 1 is compared with its +x, +y, +z neighbour;
 * a sign change emits the linearly interpolated crossing, in voxel-CORNER convention ((i+0.5)*vs, :549-550,566 -- the
 * reference is inconsistent with integrate's centre convention; reproduced), transformed by aff (= pose_).
 * Output order in the reference depends on atomics; here z-major scan order -- compare as sets.  Returns the number of
 * crossings found (may exceed capacity; only the first `capacity` are written).  A slab scans its own planes and
 * needs plane z_own_end as a halo for the +z neighbour.                                                         */
ORC_API uint64_t orc_extract_cloud(OrcVolume v, const OrcSlab *slab, const float aff[12], float *points /* float4 */,
                                   uint64_t capacity)
{
    OrcSlab s; slab_or_full(&v, slab, &s);
    const int X = v.dims[0], Y = v.dims[1], Z = v.dims[2];
    const uint16_t *base = (const uint16_t *)v.data;
    const float vsx = v.voxel_size[0], vsy = v.voxel_size[1], vsz = v.voxel_size[2];
    uint64_t n = 0;
    const int z_end = s.z_own0 + s.z_own_n < Z - 1 ? s.z_own0 + s.z_own_n : Z - 1;          /* :538 z < dims.z - 1 */
    for (int z = s.z_own0; z < z_end; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
#define VOX(xx, yy, zz) (base + 2 * ((size_t)(xx) + (size_t)(yy) * X + (size_t)((zz) - s.z_store0) * X * Y))
                const uint16_t *c = VOX(x, y, z);
                int W = c[1]; float F = h2f(c[0]);
                if (W == 0 || F == 1.f) continue;                                              /* :548 */
                f3 V = mk3(((float)x + 0.5f) * vsx, ((float)y + 0.5f) * vsy, ((float)z + 0.5f) * vsz);
                for (int axis = 0; axis < 3; ++axis) {
                    if (axis == 0 && !(x + 1 < X)) continue;                                   /* :553 */
                    if (axis == 1 && !(y + 1 < Y)) continue;                                   /* :574 */
                    const uint16_t *nb = axis == 0 ? VOX(x + 1, y, z) : axis == 1 ? VOX(x, y + 1, z) : VOX(x, y, z + 1);
                    int Wn = nb[1]; float Fn = h2f(nb[0]);
                    if (Wn == 0 || Fn == 1.f) continue;
                    if (!((F > 0 && Fn < 0) || (F < 0 && Fn > 0))) continue;                   /* :559 */
                    float d_inv = 1.f / (fabsf(F) + fabsf(Fn));                                /* :567 */
                    f3 p = V;
                    if (axis == 0) { float Vn = V.x + vsx; p.x = (V.x * fabsf(Fn) + Vn * fabsf(F)) * d_inv; }
                    if (axis == 1) { float Vn = V.y + vsy; p.y = (V.y * fabsf(Fn) + Vn * fabsf(F)) * d_inv; }
                    if (axis == 2) { float Vn = V.z + vsz; p.z = (V.z * fabsf(Fn) + Vn * fabsf(F)) * d_inv; }
                    f3 q = aff_mul(aff, p);                                                    /* :570 */
                    if (n < capacity) { float *o = points + 4 * n; o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = 0.f; }
                    ++n;
                }
#undef VOX
            }
    return n;
}

/* tsdf_volume.cu:714-795 ExtractNormals: gradient at each extracted point (Rinv*(p - aff.t) back to the volume frame),
 * NaN unless the nearest voxel is at least 2 from every face; result normalized(aff.R * n), w = 0.                */
ORC_API void orc_extract_normals(OrcVolume v, const OrcSlab *slab, const float aff[12], const float Rinv[9],
                                 const float *points /* float4 */, uint64_t n, float gradient_delta_factor,
                                 float *normals /* float4 */)
{
    rc_ctx c; rc_setup(&c, &v, slab, 0.75f, gradient_delta_factor);
    const f3 t = mk3(aff[9], aff[10], aff[11]);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        const float *pp = points + 4 * i;
        float qn = qnanf();
        f3 nrm = mk3(qn, qn, qn);
        f3 point = mat3_mul(Rinv, sub3(mk3(pp[0], pp[1], pp[2]), t));                         /* :749 */
        int gx = (int)lrintf(point.x * c.vsi.x), gy = (int)lrintf(point.y * c.vsi.y), gz = (int)lrintf(point.z * c.vsi.z);
        if (gx > 1 && gy > 1 && gz > 1 && gx < c.X - 2 && gy < c.Y - 2 && gz < c.Z - 2) {     /* :752 */
            f3 g;
            g.x = (interpolate(&c, mul3(mk3(point.x + c.gd.x, point.y, point.z), c.vsi)) -
                   interpolate(&c, mul3(mk3(point.x - c.gd.x, point.y, point.z), c.vsi))) / c.gd.x;
            g.y = (interpolate(&c, mul3(mk3(point.x, point.y + c.gd.y, point.z), c.vsi)) -
                   interpolate(&c, mul3(mk3(point.x, point.y - c.gd.y, point.z), c.vsi))) / c.gd.y;
            g.z = (interpolate(&c, mul3(mk3(point.x, point.y, point.z + c.gd.z), c.vsi)) -
                   interpolate(&c, mul3(mk3(point.x, point.y, point.z - c.gd.z), c.vsi))) / c.gd.z;
            nrm = normalized3(mat3_mul(aff, g));                                                /* :789 */
        }
        float *o = normals + 4 * i; o[0] = nrm.x; o[1] = nrm.y; o[2] = nrm.z; o[3] = 0.f;
    }
}

/* ---------------------------------------------------------------- project_and_remove / psdf
 * project_kernel, tsdf_volume.cu:113-139 (device::project_and_remove :163-176, device::project :179-192 -- the two
 * launch the same kernel) and the host arithmetic of TsdfVolume::psdf, tsdf_volume.cpp:266-292.
 * For every non-NaN point: coo = proj(point) (device.hpp:35-36); outside the image -> (qnan,qnan,qnan,0); otherwise
 * Dp = dists(coo) (point filter), the dists pixel is zeroed ("removed") and the point becomes (coo.x*Dp, coo.y*Dp, Dp, 0).
 * ro[i] (nullable) = (K^-1 * new_point)[2] - old_point.z, psdf :284-290.  K^-1 is cv::Matx33f::inv(DECOMP_LU) == OpenCV's
 * closed-form 3x3 inverse (third-party, not in the reference tree: opencv2/core/operations.hpp Matx_FastInvOp<_Tp,3>):
 * d = 1/fl(fx*fy); third row = (0, 0, fl(fl(fx*fy)*d)); the product accumulates 0*X + 0*Y + b22*Z in f32.
 * Deviations (header): the guard `x<cols || y<rows` (:119) is dropped -- points are a flat list here; NaN image
 * coordinates count as outside; the reference reads dists through a texture while other threads zero the same image
 * (racy: a second point landing on an already-removed pixel may read 0 or the old value) -- here reads come from the
 * immutable `dists_in` and the zeros go to `dists_out` (nullable; must not alias dists_in).
 * Returns the number of points that landed inside the image.                                                     */
ORC_API uint64_t orc_project_and_remove(const uint16_t *dists_in, size_t in_pitch, uint16_t *dists_out, size_t out_pitch,
                                        int cols, int rows, float *points /* float4 */, uint64_t n, const float proj[4],
                                        float *ro /* nullable */)
{
    const float P = proj[0] * proj[1];
    const float dinv = 1.f / P;
    const float b22 = P * dinv;
    uint64_t inside = 0;
    for (uint64_t i = 0; i < n; ++i) {
        float *p = points + 4 * i;
        const float px = p[0], py = p[1], pz = p[2];
        if (isnan(px) || isnan(py) || isnan(pz)) {                 /* :121 */
            if (ro) ro[i] = qnanf();                               /* (K^-1 * NaN)[2] - z */
            continue;
        }
        const float u = fmaf(proj[0], px / pz, proj[2]);
        const float w = fmaf(proj[1], py / pz, proj[3]);
        if (!(u >= 0 && w >= 0 && w < (float)rows && u < (float)cols)) {   /* :125 */
            p[0] = p[1] = p[2] = qnanf(); p[3] = 0.f;
            if (ro) ro[i] = qnanf();
            continue;
        }
        const uint16_t *row = (const uint16_t *)((const char *)dists_in + (size_t)(int)w * in_pitch);
        const float Dp = h2f(row[(int)u]);                         /* :131 */
        if (dists_out) ((uint16_t *)((char *)dists_out + (size_t)(int)w * out_pitch))[(int)u] = 0;   /* :132 */
        p[0] = u * Dp; p[1] = w * Dp; p[2] = Dp; p[3] = 0.f;        /* :133 */
        if (ro) { float s = 0.f; s = s + 0.f * p[0]; s = s + 0.f * p[1]; s = s + b22 * p[2]; ro[i] = s - pz; }
        ++inside;
    }
    return inside;
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
