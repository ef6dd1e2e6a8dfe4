// ref_glue.cpp -- builds oracle/_ref/libdfref.so FROM THE REFERENCE'S OWN SOURCES where they lie
// under /root/reference (never copied): the vendored nanoflann.hpp (v1.2.3), quaternion.hpp,
// dual_quaternion.hpp and knn_point_cloud.hpp are #included unmodified against oracle/cv_shim.
// TEST INFRASTRUCTURE ONLY.  Used to (1) validate oracle/dfusion_oracle.c's restatement of
// k-NN / weighting / DQB / transform bit-for-bit, (2) run the reference tests' known-answer
// vectors, (3) generate tests/golden/*.npz (tests/golden/make_golden.py).
//
// kfusion/src/warp_field.cpp itself cannot be compiled (it drags in Ceres / Opt / CUDA headers,
// warp_field.cpp:5-8), so the ~40 lines of WarpField::{KNN, weighting, getWeightsAndUpdateKNN,
// DQB, warp} (warp_field.cpp:180-251) are restated here ON TOP of the reference's classes.
#include <cmath>      // NB: <cmath> only, never <math.h>: keeps `exp(float)` on the double overload,
                      // as on the reference's platform (gcc 5 / Ubuntu 16.04), see static_assert below.
#include <cstring>
#include <cstdint>
#include <cstddef>
#include <immintrin.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <type_traits>
#include <vector>
#include <kfusion/types.hpp>            // oracle/cv_shim
#include <quaternion.hpp>               // /root/reference/kfusion/src/utils
#include <dual_quaternion.hpp>
#include <knn_point_cloud.hpp>
#include <nanoflann/nanoflann.hpp>      // /root/reference/kfusion/include

using namespace kfusion;

typedef nanoflann::KDTreeSingleIndexAdaptor<
        nanoflann::L2_Simple_Adaptor<float, utils::PointCloud>,
        utils::PointCloud, 3> kd_tree_t;                                   // warp_field.hpp:13-17

typedef utils::DualQuaternion<float> DQ;
static_assert(std::is_standard_layout<utils::Quaternion<float> >::value, "layout");
static_assert(sizeof(utils::Quaternion<float>) == 16, "Quaternion<float> is (w,x,y,z)");

// DualQuaternion keeps rotation_/translation_ private; they are its first 32 bytes.
static DQ dq_from_raw(const float* raw8) { DQ d; std::memcpy((void*)&d, raw8, 32); return d; }
static void dq_to_raw(const DQ& d, float* raw8) { std::memcpy(raw8, (const void*)&d, 32); }

namespace
{
    struct Node { Vec3f vertex; DQ transform; float weight; };            // warp_field.hpp:35-40

    struct Warp
    {
        std::vector<Node> nodes;
        utils::PointCloud cloud;
        kd_tree_t* index;
        int k;
        std::vector<size_t> ret_index;
        std::vector<float> out_dist_sqr;

        Warp(const float* pos, const float* dq, const float* sigma, int M, int k_) : index(0), k(k_), ret_index(k_), out_dist_sqr(k_)
        {
            nodes.resize(M);
            cloud.pts.resize(M);
            for (int i = 0; i < M; ++i) {
                nodes[i].vertex = Vec3f(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
                if (dq) nodes[i].transform = dq_from_raw(dq + 8 * i);
                nodes[i].weight = sigma ? sigma[i] : 0.f;
                cloud.pts[i] = nodes[i].vertex;                             // buildKDTree warp_field.cpp:275-282
            }
            index = new kd_tree_t(3, cloud, nanoflann::KDTreeSingleIndexAdaptorParams(10));   // warp_field.cpp:20
            index->buildIndex();
        }
        ~Warp() { delete index; }

        void KNN(Vec3f point)                                               // warp_field.cpp:247-251
        {
            nanoflann::KNNResultSet<float> rs(k);
            rs.init(&ret_index[0], &out_dist_sqr[0]);
            index->findNeighbors(rs, point.val, nanoflann::SearchParams(10));
        }
        float weighting(float squared_dist, float weight) const             // warp_field.cpp:238-241
        {
            static_assert(std::is_same<decltype(exp(-squared_dist / (2 * weight * weight))), double>::value,
                          "exp(float) must resolve to the double overload (gcc5/<cmath> semantics)");
            return (float) exp(-squared_dist / (2 * weight * weight));
        }
        DQ DQB(const Vec3f& vertex)                                         // warp_field.cpp:203-217
        {
            KNN(vertex);
            utils::Quaternion<float> translation_sum(0, 0, 0, 0);
            utils::Quaternion<float> rotation_sum(0, 0, 0, 0);
            for (int i = 0; i < k; i++) {
                float w = weighting(out_dist_sqr[i], nodes[ret_index[i]].weight);
                translation_sum += w * nodes[ret_index[i]].transform.getTranslation();
                rotation_sum += w * nodes[ret_index[i]].transform.getRotation();
            }
            rotation_sum.normalize();
            return DQ(translation_sum, rotation_sum);
        }
    };
}

extern "C" {

// exact k-NN through the reference's nanoflann tree: idx_out[N*k] (int32), d2_out[N*k]
void ref_knn(const float* pos, int M, const float* queries, int N, int k, int* idx_out, float* d2_out)
{
    Warp w(pos, 0, 0, M, k);
    for (int i = 0; i < N; ++i) {
        w.KNN(Vec3f(queries[3 * i], queries[3 * i + 1], queries[3 * i + 2]));
        for (int j = 0; j < k; ++j) { idx_out[(size_t)i * k + j] = (int)w.ret_index[j]; d2_out[(size_t)i * k + j] = w.out_dist_sqr[j]; }
    }
}

// DQB(p) for each point: out_dq[N*8] = {rotation_, translation_}
void ref_dqb(const float* pos, const float* dq, const float* sigma, int M, int k, const float* points, int N, float* out_dq)
{
    Warp w(pos, dq, sigma, M, k);
    for (int i = 0; i < N; ++i) {
        DQ d = w.DQB(Vec3f(points[3 * i], points[3 * i + 1], points[3 * i + 2]));
        dq_to_raw(d, out_dq + 8 * (size_t)i);
    }
}

// WarpField::warp (warp_field.cpp:180-195) with warp_to_live_ = identity and the index-drift
// bug fixed (indexed by position).  points/normals N x 3, in place.
void ref_warp_points(const float* pos, const float* dq, const float* sigma, int M, int k, float* points, float* normals, int N)
{
    Warp w(pos, dq, sigma, M, k);
    for (int i = 0; i < N; ++i) {
        Vec3f point(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
        if (std::isnan(point[0]) || (normals && std::isnan(normals[3 * i]))) continue;
        DQ dqb = w.DQB(point);
        dqb.transform(point);
        points[3 * i] = point[0]; points[3 * i + 1] = point[1]; points[3 * i + 2] = point[2];
        if (normals) {
            Vec3f n(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
            dqb.transform(n);
            normals[3 * i] = n[0]; normals[3 * i + 1] = n[1]; normals[3 * i + 2] = n[2];
        }
    }
}

// ---- known-answer helpers mirroring tests/utils/test_quaternion.cc / test_dual_quaternion.cc
void ref_quat_encode_rotation(float theta, float x, float y, float z, float out[4])
{ utils::Quaternion<float> q; q.encodeRotation(theta, x, y, z); std::memcpy(out, &q, 16); }

void ref_quat_rotate_xyz(const float q[4], float v[3])
{ utils::Quaternion<float> a(q[0], q[1], q[2], q[3]); a.rotate(v[0], v[1], v[2]); }

void ref_quat_mul(const float a[4], const float b[4], float out[4])
{ utils::Quaternion<float> x(a[0], a[1], a[2], a[3]), y(b[0], b[1], b[2], b[3]); utils::Quaternion<float> r = x * y; std::memcpy(out, &r, 16); }

float ref_quat_dot(const float a[4], const float b[4])
{ utils::Quaternion<float> x(a[0], a[1], a[2], a[3]), y(b[0], b[1], b[2], b[3]); return x.dotProduct(y); }

void ref_quat_normalize(const float a[4], float out[4])
{ utils::Quaternion<float> x(a[0], a[1], a[2], a[3]); x.normalize(); std::memcpy(out, &x, 16); }

// DualQuaternion(x,y,z,roll,pitch,yaw): out = rotation(4), getTranslation()(4)
void ref_dq_euler(float x, float y, float z, float roll, float pitch, float yaw, float out_rot[4], float out_trans[4])
{
    DQ d(x, y, z, roll, pitch, yaw);
    utils::Quaternion<float> r = d.getRotation(), t = d.getTranslation();
    std::memcpy(out_rot, &r, 16); std::memcpy(out_trans, &t, 16);
}

void ref_dq_from_twist(const float r[3], const float t[3], float dq_out[8])
{ DQ d; d.from_twist(r[0], r[1], r[2], t[0], t[1], t[2]); dq_to_raw(d, dq_out); }

void ref_dq_get_translation(const float dq[8], float out[4])
{ DQ d = dq_from_raw(dq); utils::Quaternion<float> t = d.getTranslation(); std::memcpy(out, &t, 16); }

void ref_dq_transform(const float dq[8], float p[3])
{ DQ d = dq_from_raw(dq); Vec3f v(p[0], p[1], p[2]); d.transform(v); p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; }

int ref_nanoflann_version(void) { return NANOFLANN_VERSION; }

// ---- the reference-style CPU path for the north-star kernel: per-voxel warped integrate with the reference's OWN
// nanoflann k-NN + DQB + DualQuaternion::transform (one Warp -- tree, result set, scratch vectors -- PER THREAD: the reference's
// WarpField keeps them in globals, warp_field.cpp:11-15, and is not re-entrant), OpenMP over Z planes.  The TSDF update after the
// warp is the ~20 lines of TsdfIntegrator::operator() from the projection on (tsdf_volume.cu:77-104) in plain C++ (that kernel
// is pinned separately, oracle/ref_cu_glue.cpp).  Planes [z0, z0 + zn) of a volume whose blob starts at plane z_store0.
// threads <= 0: all host cores.  Returns the number of updated voxels.
unsigned long long ref_integrate_warped(const unsigned short* dists, size_t pitch, int cols, int rows, unsigned short* vol,
                                        int X, int Y, int z_store0, int z0, int zn, const float vs[3], float trunc, int max_weight,
                                        const float vol2world[12], const float world2cam[12], const float proj[4],
                                        const float* pos, const float* dq, const float* sigma, int M, int k, int threads,
                                        int* threads_used)
{
    const float trunc_inv = 1.f / trunc;
    unsigned long long n_upd = 0;
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel num_threads(threads) reduction(+ : n_upd)
#endif
    {
        Warp w(pos, dq, sigma, M, k);
#ifdef _OPENMP
#pragma omp single
        used = omp_get_num_threads();
#pragma omp for collapse(2) schedule(dynamic, 4)
#endif
        for (int z = z0; z < z0 + zn; ++z)
            for (int y = 0; y < Y; ++y)
                for (int x = 0; x < X; ++x) {
                    const float vx = (float)x * vs[0], vy = (float)y * vs[1], vz = (float)z * vs[2];
                    // Aff3f * v (device.hpp:71-74): three fma-nested dots + t
                    const float* A = vol2world;
                    Vec3f p(fmaf(A[0], vx, fmaf(A[1], vy, A[2] * vz)) + A[9], fmaf(A[3], vx, fmaf(A[4], vy, A[5] * vz)) + A[10],
                            fmaf(A[6], vx, fmaf(A[7], vy, A[8] * vz)) + A[11]);
                    DQ d = w.DQB(p);                                            // warp_field.cpp:203-217 (nanoflann inside)
                    d.transform(p);                                             // warp_field.cpp:188
                    const float* C = world2cam;
                    const float cx_ = fmaf(C[0], p[0], fmaf(C[1], p[1], C[2] * p[2])) + C[9];
                    const float cy_ = fmaf(C[3], p[0], fmaf(C[4], p[1], C[5] * p[2])) + C[10];
                    const float cz_ = fmaf(C[6], p[0], fmaf(C[7], p[1], C[8] * p[2])) + C[11];
                    const float u = fmaf(proj[0], cx_ / cz_, proj[2]);          // device.hpp:35
                    const float v = fmaf(proj[1], cy_ / cz_, proj[3]);          // device.hpp:36
                    if (!(u >= 0 && v >= 0 && u < (float)cols && v < (float)rows)) continue;   // tsdf_volume.cu:82
                    const unsigned short* row = (const unsigned short*)((const char*)dists + (size_t)(int)v * pitch);
                    const float Dp = _cvtsh_ss(row[(int)u]);                    // :85
                    if (Dp == 0 || cz_ <= 0) continue;                          // :86
                    const float sdf = Dp - sqrtf(fmaf(cx_, cx_, fmaf(cy_, cy_, cz_ * cz_)));   // :89
                    if (sdf >= -trunc) {                                        // :91
                        const float tsdf = fminf(1.f, sdf * trunc_inv);
                        unsigned short* vox = vol + 2 * ((size_t)x + (size_t)y * X + (size_t)(z - z_store0) * X * Y);
                        const int wp = vox[1];
                        const float fp = _cvtsh_ss(vox[0]);
                        const float fn = fmaf(fp, (float)wp, tsdf) / (float)(wp + 1);           // :99
                        vox[0] = _cvtss_sh(fn, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
                        vox[1] = (unsigned short)(wp + 1 < max_weight ? wp + 1 : max_weight);
                        ++n_upd;
                    }
                }
    }
    if (threads_used) *threads_used = used;
    return n_upd;
}

}  // extern "C"
