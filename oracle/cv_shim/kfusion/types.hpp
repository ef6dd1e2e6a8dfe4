// Stand-in for the reference's kfusion/types.hpp (which needs OpenCV, absent here) so that the
// reference's OWN headers -- kfusion/src/utils/quaternion.hpp, dual_quaternion.hpp,
// knn_point_cloud.hpp and the vendored nanoflann.hpp -- compile UNMODIFIED from where they lie
// under /root/reference.  TEST INFRASTRUCTURE ONLY (oracle/_ref build); never shipped.
//
// Only the cv::Vec3f surface those headers touch is provided, with OpenCV's float semantics:
// every operator is a component-wise single-precision op, cross() is
// (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)   [opencv2/core/matx.hpp Vec<float,3>::cross].
#pragma once
#include <cmath>
#include <cassert>   // OpenCV pulls this in for quaternion.hpp:224-226
#include <cstddef>
#include <vector>

namespace cv
{
    struct Vec3f
    {
        float val[3];
        Vec3f() { val[0] = val[1] = val[2] = 0.f; }
        Vec3f(float a, float b, float c) { val[0] = a; val[1] = b; val[2] = c; }
        float& operator[](int i) { return val[i]; }
        const float& operator[](int i) const { return val[i]; }
        Vec3f cross(const Vec3f& v) const
        {
            return Vec3f(val[1] * v.val[2] - val[2] * v.val[1],
                         val[2] * v.val[0] - val[0] * v.val[2],
                         val[0] * v.val[1] - val[1] * v.val[0]);
        }
        float dot(const Vec3f& v) const { return val[0] * v.val[0] + val[1] * v.val[1] + val[2] * v.val[2]; }
        Vec3f& operator+=(const Vec3f& v) { val[0] += v.val[0]; val[1] += v.val[1]; val[2] += v.val[2]; return *this; }
    };
    inline Vec3f operator+(const Vec3f& a, const Vec3f& b) { return Vec3f(a.val[0] + b.val[0], a.val[1] + b.val[1], a.val[2] + b.val[2]); }
    inline Vec3f operator*(const Vec3f& a, float s) { return Vec3f(a.val[0] * s, a.val[1] * s, a.val[2] * s); }
    inline bool operator!=(const Vec3f& a, const Vec3f& b) { return a.val[0] != b.val[0] || a.val[1] != b.val[1] || a.val[2] != b.val[2]; }
    inline Vec3f normalize(const Vec3f& v)
    {
        float n = std::sqrt(v.dot(v));
        return n > 0 ? v * (1.f / n) : v;
    }
    // Quaternion(const Vec3f& normal) names cv::Mat3f (quaternion.hpp:47-55); never executed here.
    struct Mat3f
    {
        std::vector<Vec3f> rows_;
        void push_back(const Vec3f& r) { rows_.push_back(r); }
        template <typename T> T at(int i, int j) const { return rows_[i].val[j]; }
    };
}

namespace kfusion
{
    typedef cv::Vec3f Vec3f;
}
