// kfusion/types.hpp -- value types named by the hot-path API of the reference
// (/root/reference/kfusion/include/kfusion/types.hpp:11-63).
// With -DKFUSION_USE_OPENCV (and the OpenCV headers on the include path) the reference's own typedefs are used -- cv::Vec3f,
// cv::Affine3f, cv::Mat returns, cv::Ptr<KinFu> -- so that the reference's apps/demo.cpp compiles against these headers
// UNMODIFIED (tests/test_demo_ref.py does that, against a test-side stand-in for OpenCV, which this image does not have).
// Without it the minimal stand-ins below are used; they are what the headless harnesses and the GPU tests exercise.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <kfusion/cuda/device_array.hpp>

#if defined(KFUSION_USE_OPENCV) && __has_include(<opencv2/core/affine.hpp>)
#include <opencv2/core/core.hpp>
#include <opencv2/core/affine.hpp>
namespace kfusion
{
    typedef cv::Matx33f Mat3f;
    typedef cv::Vec3f Vec3f;
    typedef cv::Vec3i Vec3i;
    typedef cv::Affine3f Affine3f;
}
#else
namespace kfusion
{
    template <typename T, int N> struct VecN
    {
        T val[N];
        VecN() { for (int i = 0; i < N; ++i) val[i] = T(0); }
        VecN(T a, T b, T c) { static_assert(N == 3, "3-vector ctor"); val[0] = a; val[1] = b; val[2] = c; }
        static VecN all(T v) { VecN r; for (int i = 0; i < N; ++i) r.val[i] = v; return r; }
        T& operator[](int i) { return val[i]; }
        const T& operator[](int i) const { return val[i]; }
    };
    typedef VecN<float, 3> Vec3f;
    typedef VecN<int, 3> Vec3i;
    inline Vec3f operator+(const Vec3f& a, const Vec3f& b) { return Vec3f(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
    inline Vec3f operator-(const Vec3f& a, const Vec3f& b) { return Vec3f(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
    inline Vec3f operator*(const Vec3f& a, float s) { return Vec3f(a[0] * s, a[1] * s, a[2] * s); }

    struct Mat3f
    {
        float val[9];                                    // row-major, like cv::Matx33f
        Mat3f() { std::memset(val, 0, sizeof(val)); val[0] = val[4] = val[8] = 1.f; }
        float& operator()(int r, int c) { return val[3 * r + c]; }
        float operator()(int r, int c) const { return val[3 * r + c]; }
        Mat3f inv() const                                // cv::Matx33f::inv(DECOMP_SVD) stand-in: adjugate in double
        {
            const float* m = val; double d[9];
            d[0] = (double)m[4] * m[8] - (double)m[5] * m[7]; d[1] = (double)m[2] * m[7] - (double)m[1] * m[8]; d[2] = (double)m[1] * m[5] - (double)m[2] * m[4];
            d[3] = (double)m[5] * m[6] - (double)m[3] * m[8]; d[4] = (double)m[0] * m[8] - (double)m[2] * m[6]; d[5] = (double)m[2] * m[3] - (double)m[0] * m[5];
            d[6] = (double)m[3] * m[7] - (double)m[4] * m[6]; d[7] = (double)m[1] * m[6] - (double)m[0] * m[7]; d[8] = (double)m[0] * m[4] - (double)m[1] * m[3];
            const double det = m[0] * d[0] + m[1] * d[3] + m[2] * d[6];
            Mat3f r; for (int i = 0; i < 9; ++i) r.val[i] = (float)(d[i] / det);
            return r;
        }
    };

    // cv::Affine3f subset: rotation(), translation(), inv(), operator*, translate(), Identity()
    struct Affine3f
    {
        Mat3f R; Vec3f t;
        Affine3f() {}
        Affine3f(const Mat3f& R_, const Vec3f& t_) : R(R_), t(t_) {}
        /// cv::Affine3f(rvec, t): Rodrigues rotation (opencv2/core/affine.hpp Affine3<T>::rotation(const Vec3&)), in double
        Affine3f(const Vec3f& rvec, const Vec3f& t_) : t(t_)
        {
            const double rx = rvec[0], ry = rvec[1], rz = rvec[2];
            const double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
            if (theta >= 2.220446049250313e-16) {
                const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, it = 1. / theta;
                const double k[3] = {rx * it, ry * it, rz * it};
                const double K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j)
                        R(i, j) = (float)(c * (i == j ? 1. : 0.) + c1 * k[i] * k[j] + s * K[3 * i + j]);
            }
        }
        static Affine3f Identity() { return Affine3f(); }
        Mat3f rotation() const { return R; }
        Vec3f translation() const { return t; }
        Affine3f translate(const Vec3f& d) const { Affine3f r(*this); r.t = r.t + d; return r; }
        Affine3f inv(int /*method*/ = 0) const
        {
            Affine3f r; r.R = R.inv();
            for (int i = 0; i < 3; ++i)
                r.t[i] = (float)-((double)r.R(i, 0) * t[0] + (double)r.R(i, 1) * t[1] + (double)r.R(i, 2) * t[2]);
            return r;
        }
    };
    inline Affine3f operator*(const Affine3f& a, const Affine3f& b)       // a applied after b
    {
        Affine3f r;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j)
                r.R(i, j) = (float)((double)a.R(i, 0) * b.R(0, j) + (double)a.R(i, 1) * b.R(1, j) + (double)a.R(i, 2) * b.R(2, j));
            r.t[i] = (float)((double)a.R(i, 0) * b.t[0] + (double)a.R(i, 1) * b.t[1] + (double)a.R(i, 2) * b.t[2] + (double)a.t[i]);
        }
        return r;
    }
    inline Vec3f operator*(const Affine3f& a, const Vec3f& v)             // cv::Affine3f * Vec3f: left-associated floats
    {
        return Vec3f(a.R(0, 0) * v[0] + a.R(0, 1) * v[1] + a.R(0, 2) * v[2] + a.t[0],
                     a.R(1, 0) * v[0] + a.R(1, 1) * v[1] + a.R(1, 2) * v[2] + a.t[1],
                     a.R(2, 0) * v[0] + a.R(2, 1) * v[1] + a.R(2, 2) * v[2] + a.t[2]);
    }
}
#endif

namespace kfusion
{
    struct Intr                                          // types.hpp:20-27
    {
        float fx, fy, cx, cy;
        Intr() : fx(0), fy(0), cx(0), cy(0) {}
        Intr(float fx_, float fy_, float cx_, float cy_) : fx(fx_), fy(fy_), cx(cx_), cy(cy_) {}
        Intr operator()(int level_index) const { int d = 1 << level_index; return Intr(fx / d, fy / d, cx / d, cy / d); }
    };
    struct Point { union { float data[4]; struct { float x, y, z; }; }; };   // types.hpp:31-38
    typedef Point Normal;

    namespace cuda
    {
        typedef DeviceMemory CudaData;                   // types.hpp:58-63
        typedef DeviceArray2D<unsigned short> Depth;
        typedef DeviceArray2D<unsigned short> Dists;
        typedef DeviceArray2D<Normal> Normals;
        typedef DeviceArray2D<Point> Cloud;
        struct RGB { unsigned char b, g, r, a; };          // types.hpp:42-50 (rendering itself is out of scope)
        typedef DeviceArray2D<RGB> Image;
        struct Frame                                     // types.hpp:65-72
        {
            bool use_points;
            std::vector<Depth> depth_pyr;
            std::vector<Cloud> points_pyr;
            std::vector<Normals> normals_pyr;
        };
    }
    inline float deg2rad(float alpha) { return alpha * 0.017453293f; }   // types.hpp:75

    // row-major R[9] then t[3]: device::Aff3f as the C-ABI wants it (precomp.hpp:19-28 device_cast)
    inline void affine_to_aff12(const Affine3f& a, float out[12])
    {
        const Mat3f R = a.rotation(); const Vec3f t = a.translation();
        for (int i = 0; i < 9; ++i) out[i] = R.val[i];
        for (int i = 0; i < 3; ++i) out[9 + i] = t[i];
    }
    inline Affine3f aff12_to_affine(const float in[12])                 // (through the (R, t) constructor: what cv::Affine3f offers too)
    {
        Mat3f R; Vec3f t;
        for (int i = 0; i < 9; ++i) R.val[i] = in[i];
        for (int i = 0; i < 3; ++i) t[i] = in[9 + i];
        return Affine3f(R, t);
    }
}
