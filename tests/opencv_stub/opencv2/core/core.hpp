// tests/opencv_stub/opencv2/core/core.hpp -- TEST INFRASTRUCTURE, not OpenCV.
// The reference's demo (/root/reference/apps/demo.cpp) and its public headers name OpenCV types; OpenCV is not installed in this
// image.  This stub supplies just the names that file and the kfusion interface use -- cv::Vec, cv::Matx33f, cv::Affine3<T>, cv::Mat,
// cv::Ptr, cv::String, cv::glob, the highgui calls and cv::viz -- so that the reference's demo.cpp can be COMPILED UNMODIFIED and
// LINKED against this repository's kfusion mirror built with -DKFUSION_USE_OPENCV (tests/test_demo_ref.py).  The arithmetic of the
// small value types is the one of the mirror's own stand-ins (kfusion/types.hpp), so both builds of the mirror give the same bits;
// windows are files, images on disk are raw arrays with a 16-byte header (see imread in highgui.hpp).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_16U 2
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn) - 1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(CV_8U, 4)
#define CV_16UC1 CV_MAKETYPE(CV_16U, 1)
#define CV_32FC3 CV_MAKETYPE(CV_32F, 3)
#define CV_32FC4 CV_MAKETYPE(CV_32F, 4)

namespace cv
{
    typedef std::string String;
    enum { DECOMP_LU = 0, DECOMP_SVD = 1 };

    template <typename T, int N> struct Vec
    {
        T val[N];
        Vec() { for (int i = 0; i < N; ++i) val[i] = T(0); }
        Vec(T a, T b, T c) { static_assert(N == 3, "3-vector ctor"); val[0] = a; val[1] = b; val[2] = c; }
        Vec(T a, T b, T c, T d) { static_assert(N == 4, "4-vector ctor"); val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
        // (names the reference's quaternion.hpp uses, tests/ref_host_bridge: plain T arithmetic, left to right)
        T dot(const Vec& o) const { T s = T(0); for (int i = 0; i < N; ++i) s += val[i] * o.val[i]; return s; }
        Vec cross(const Vec& o) const
        {
            static_assert(N == 3, "cross product of 3-vectors");
            return Vec(val[1] * o.val[2] - val[2] * o.val[1], val[2] * o.val[0] - val[0] * o.val[2], val[0] * o.val[1] - val[1] * o.val[0]);
        }
        template <typename U> Vec(const Vec<U, N>& o) { for (int i = 0; i < N; ++i) val[i] = (T)o.val[i]; }
        static Vec all(T v) { Vec r; for (int i = 0; i < N; ++i) r.val[i] = v; return r; }
        T& operator[](int i) { return val[i]; }
        const T& operator[](int i) const { return val[i]; }
        Vec& operator+=(const Vec& o) { for (int i = 0; i < N; ++i) val[i] += o.val[i]; return *this; }
    };
    template <typename T, int N> inline Vec<T, N> operator+(const Vec<T, N>& a, const Vec<T, N>& b) { Vec<T, N> r; for (int i = 0; i < N; ++i) r[i] = a[i] + b[i]; return r; }
    template <typename T, int N> inline Vec<T, N> operator-(const Vec<T, N>& a, const Vec<T, N>& b) { Vec<T, N> r; for (int i = 0; i < N; ++i) r[i] = a[i] - b[i]; return r; }
    template <typename T, int N> inline Vec<T, N> operator*(const Vec<T, N>& a, T s) { Vec<T, N> r; for (int i = 0; i < N; ++i) r[i] = a[i] * s; return r; }
    typedef Vec<float, 3> Vec3f;
    typedef Vec<float, 4> Vec4f;
    typedef Vec<double, 3> Vec3d;
    typedef Vec<int, 3> Vec3i;
    template <typename T, int N> inline Vec<T, N> normalize(const Vec<T, N>& v)
    {
        double s = 0; for (int i = 0; i < N; ++i) s += (double)v[i] * v[i];
        const double n = std::sqrt(s); Vec<T, N> r; for (int i = 0; i < N; ++i) r[i] = (T)(n > 0 ? v[i] / n : 0); return r;
    }
    struct Matx44f { float val[16]; Matx44f() { std::memset(val, 0, sizeof(val)); val[0] = val[5] = val[10] = val[15] = 1.f; } };

    template <typename T> struct Matx33
    {
        T val[9];                                        // row-major
        Matx33() { std::memset(val, 0, sizeof(val)); val[0] = val[4] = val[8] = T(1); }
        template <typename U> Matx33(const Matx33<U>& o) { for (int i = 0; i < 9; ++i) val[i] = (T)o.val[i]; }
        Matx33(T a0, T a1, T a2, T a3, T a4, T a5, T a6, T a7, T a8) { const T v[9] = {a0, a1, a2, a3, a4, a5, a6, a7, a8}; std::memcpy(val, v, sizeof(val)); }
        T& operator()(int r, int c) { return val[3 * r + c]; }
        T operator()(int r, int c) const { return val[3 * r + c]; }
        Matx33 inv(int /*method*/ = DECOMP_LU) const     // adjugate in double (kfusion/types.hpp Mat3f::inv)
        {
            const T* m = val; double d[9];
            d[0] = (double)m[4] * m[8] - (double)m[5] * m[7]; d[1] = (double)m[2] * m[7] - (double)m[1] * m[8]; d[2] = (double)m[1] * m[5] - (double)m[2] * m[4];
            d[3] = (double)m[5] * m[6] - (double)m[3] * m[8]; d[4] = (double)m[0] * m[8] - (double)m[2] * m[6]; d[5] = (double)m[2] * m[3] - (double)m[0] * m[5];
            d[6] = (double)m[3] * m[7] - (double)m[4] * m[6]; d[7] = (double)m[1] * m[6] - (double)m[0] * m[7]; d[8] = (double)m[0] * m[4] - (double)m[1] * m[3];
            const double det = m[0] * d[0] + m[1] * d[3] + m[2] * d[6];
            Matx33 r; for (int i = 0; i < 9; ++i) r.val[i] = (T)(d[i] / det);
            return r;
        }
    };
    typedef Matx33<float> Matx33f;
    typedef Matx33<double> Matx33d;
    template <typename T> inline Vec<T, 3> operator*(const Matx33<T>& m, const Vec<T, 3>& v)      // (Matx * Vec: in T, left to right)
    {
        return Vec<T, 3>(m(0, 0) * v[0] + m(0, 1) * v[1] + m(0, 2) * v[2], m(1, 0) * v[0] + m(1, 1) * v[1] + m(1, 2) * v[2],
                         m(2, 0) * v[0] + m(2, 1) * v[1] + m(2, 2) * v[2]);
    }

    // reference-counted owner (cv::Ptr<T>(new T) as demo.cpp:27 uses it)
    template <typename T> struct Ptr : std::shared_ptr<T>
    {
        Ptr() {}
        Ptr(T* p) : std::shared_ptr<T>(p) {}
        Ptr(const std::shared_ptr<T>& p) : std::shared_ptr<T>(p) {}
    };

    // dense 2-D array: rows x cols elements of CV_MAKETYPE(depth, channels), row pitch `step` bytes, shared buffer
    class Mat
    {
    public:
        int rows, cols; size_t step; unsigned char* data;
        Mat() : rows(0), cols(0), step(0), data(nullptr), type_(0) {}
        Mat(int r, int c, int type) : rows(0), cols(0), step(0), data(nullptr), type_(0) { create(r, c, type); }
        void create(int r, int c, int type)
        {
            if (r == rows && c == cols && type == type_ && data) return;
            rows = r; cols = c; type_ = type; step = (size_t)c * elemSize();
            buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * step);
            data = buf_->data();
        }
        int type() const { return type_; }
        int depth() const { return type_ & 7; }
        int channels() const { return (type_ >> 3) + 1; }
        size_t elemSize() const { static const int sz[8] = {1, 1, 2, 2, 4, 4, 8, 2}; return (size_t)sz[depth()] * channels(); }
        size_t total() const { return (size_t)rows * cols; }
        bool empty() const { return data == nullptr || total() == 0; }
        template <typename T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
        template <typename T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
        template <typename T> T& at(int i) { return ((T*)data)[i]; }                       // single-row / continuous matrices
        template <typename T> const T& at(int i) const { return ((const T*)data)[i]; }
        template <typename T> T& at(int r, int c) { return ptr<T>(r)[c]; }
        template <typename T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }     // (as cv::Mat: warp_field.cpp:53 reads a const Mat)
        void convertTo(Mat& m, int rtype, double alpha = 1, double beta = 0) const         // u16 / u8 -> u8 only (demo.cpp:39)
        {
            m.create(rows, cols, CV_MAKETYPE(rtype & 7, channels()));
            for (int r = 0; r < rows; ++r)
                for (int c = 0; c < cols * channels(); ++c) {
                    const double v = (depth() == CV_16U ? (double)ptr<unsigned short>(r)[c] : (double)ptr<unsigned char>(r)[c]) * alpha + beta;
                    m.ptr<unsigned char>(r)[c] = (unsigned char)std::min(255.0, std::max(0.0, std::nearbyint(v)));
                }
        }
    private:
        int type_;
        std::shared_ptr<std::vector<unsigned char>> buf_;
    };

    struct Mat3f
    {
        std::vector<Vec3f> rows_;
        void push_back(const Vec3f& v) { rows_.push_back(v); }
        template <typename T> T at(int r, int c) const { return (T)rows_[(size_t)r][c]; }
    };

    void glob(String pattern, std::vector<String>& result, bool recursive = false);   // every entry of the directory `pattern`
}
