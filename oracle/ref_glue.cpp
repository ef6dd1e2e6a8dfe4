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

}  // extern "C"
