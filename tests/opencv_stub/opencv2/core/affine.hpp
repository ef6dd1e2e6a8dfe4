// tests/opencv_stub/opencv2/core/affine.hpp -- TEST INFRASTRUCTURE (see core.hpp): cv::Affine3<T>, arithmetic of kfusion/types.hpp
#pragma once
#include <opencv2/core/core.hpp>
namespace cv
{
    template <typename T> struct Affine3
    {
        typedef Matx33<T> Mat3; typedef Vec<T, 3> Vec3;
        Mat3 R; Vec3 t;
        Affine3() {}
        Affine3(const Mat3& R_, const Vec3& t_ = Vec3()) : R(R_), t(t_) {}
        Affine3(const Vec3& rvec, const Vec3& t_) : t(t_)            // Rodrigues, in double
        {
            const double rx = rvec[0], ry = rvec[1], rz = rvec[2];
            const double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
            if (theta >= 2.220446049250313e-16) {
                const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, it = 1. / theta;
                const double k[3] = {rx * it, ry * it, rz * it};
                const double K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j)
                        R(i, j) = (T)(c * (i == j ? 1. : 0.) + c1 * k[i] * k[j] + s * K[3 * i + j]);
            }
        }
        template <typename U> Affine3(const Affine3<U>& o) : R(o.R), t(o.t) {}          // Affine3f <-> Affine3d (demo.cpp:31, 48, 104)
        static Affine3 Identity() { return Affine3(); }
        Mat3 rotation() const { return R; }
        Vec3 translation() const { return t; }
        Affine3 translate(const Vec3& d) const { Affine3 r(*this); r.t = r.t + d; return r; }
        Affine3 inv(int /*method*/ = DECOMP_SVD) const
        {
            Affine3 r; r.R = R.inv();
            for (int i = 0; i < 3; ++i)
                r.t[i] = (T)-((double)r.R(i, 0) * t[0] + (double)r.R(i, 1) * t[1] + (double)r.R(i, 2) * t[2]);
            return r;
        }
    };
    template <typename T> inline Affine3<T> operator*(const Affine3<T>& a, const Affine3<T>& b)
    {
        Affine3<T> r;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j)
                r.R(i, j) = (T)((double)a.R(i, 0) * b.R(0, j) + (double)a.R(i, 1) * b.R(1, j) + (double)a.R(i, 2) * b.R(2, j));
            r.t[i] = (T)((double)a.R(i, 0) * b.t[0] + (double)a.R(i, 1) * b.t[1] + (double)a.R(i, 2) * b.t[2] + (double)a.t[i]);
        }
        return r;
    }
    template <typename T> inline Vec<T, 3> operator*(const Affine3<T>& a, const Vec<T, 3>& v)   // left-associated, in T
    {
        return Vec<T, 3>(a.R(0, 0) * v[0] + a.R(0, 1) * v[1] + a.R(0, 2) * v[2] + a.t[0],
                         a.R(1, 0) * v[0] + a.R(1, 1) * v[1] + a.R(1, 2) * v[2] + a.t[1],
                         a.R(2, 0) * v[0] + a.R(2, 1) * v[1] + a.R(2, 2) * v[2] + a.t[2]);
    }
    typedef Affine3<float> Affine3f;
    typedef Affine3<double> Affine3d;
}
