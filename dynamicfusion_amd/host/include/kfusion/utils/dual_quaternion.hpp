// kfusion/utils/dual_quaternion.hpp -- host-side Quaternion<T> / DualQuaternion<T> with the member names the
// hot-path API mentions (deformation_node::transform, warp_field.hpp:35-40).  Independent implementation; the
// arithmetic follows /root/reference/kfusion/src/utils/quaternion.hpp and dual_quaternion.hpp operation for
// operation where results feed the GPU path (product :186-194, normalize :220-228, getTranslation :120-125,
// from_twist :212-229).  The first 32 bytes of DualQuaternion<float> are {rotation_, translation_} (w,x,y,z each)
// -- exactly the dq[8] layout dfusion_warp_set_nodes takes.
#pragma once
#include <cmath>

namespace kfusion
{
    namespace utils
    {
        template <typename T> struct Quaternion
        {
            T w_, x_, y_, z_;
            Quaternion() : w_(1), x_(0), y_(0), z_(0) {}
            Quaternion(T w, T x, T y, T z) : w_(w), x_(x), y_(y), z_(z) {}
            Quaternion conjugate() const { return Quaternion(w_, -x_, -y_, -z_); }
            T norm() const { return (T)std::sqrt((w_ * w_) + (x_ * x_) + (y_ * y_) + (z_ * z_)); }
            void normalize()                                       // scalar is double, one rounding per component
            {
                const double inv = 1.0 / (double)norm();
                w_ = (T)(inv * (double)w_); x_ = (T)(inv * (double)x_); y_ = (T)(inv * (double)y_); z_ = (T)(inv * (double)z_);
            }
            Quaternion operator*(const Quaternion& o) const
            {
                return Quaternion(((w_ * o.w_) - (x_ * o.x_) - (y_ * o.y_) - (z_ * o.z_)),
                                  ((w_ * o.x_) + (x_ * o.w_) + (y_ * o.z_) - (z_ * o.y_)),
                                  ((w_ * o.y_) - (x_ * o.z_) + (y_ * o.w_) + (z_ * o.x_)),
                                  ((w_ * o.z_) + (x_ * o.y_) - (y_ * o.x_) + (z_ * o.w_)));
            }
            Quaternion operator+(const Quaternion& o) const { return Quaternion(w_ + o.w_, x_ + o.x_, y_ + o.y_, z_ + o.z_); }
            bool operator==(const Quaternion& o) const { return w_ == o.w_ && x_ == o.x_ && y_ == o.y_ && z_ == o.z_; }
        };
        template <typename T> inline Quaternion<T> operator*(T s, const Quaternion<T>& q) { return Quaternion<T>(s * q.w_, s * q.x_, s * q.y_, s * q.z_); }

        template <typename T> class DualQuaternion
        {
        public:
            DualQuaternion() {}                                    // both quaternions (1,0,0,0), as the reference's default
            DualQuaternion(const Quaternion<T>& translation, const Quaternion<T>& rotation)
                : rotation_(rotation), translation_((T(0.5) * translation) * rotation) {}
            Quaternion<T> getRotation() const { return rotation_; }
            Quaternion<T> getDual() const { return translation_; }
            Quaternion<T> getTranslation() const
            {
                Quaternion<T> rot = rotation_;
                rot.normalize();
                return (T(2) * translation_) * rot.conjugate();
            }
            void getTranslation(T& x, T& y, T& z) const { const Quaternion<T> t = getTranslation(); x = t.x_; y = t.y_; z = t.z_; }
            void from_twist(T r0, T r1, T r2, T x, T y, T z)
            {
                const T norm = (T)std::sqrt((double)(r0 * r0 + r1 * r1 + r2 * r2));
                Quaternion<T> rot;
                if (norm > T(1e-6)) {
                    T c = (T)std::cos((double)norm);
                    const T sign = (T)((c > 0) - (c < 0));
                    c *= sign;
                    const T s = (T)((double)sign * std::sin((double)norm) / (double)norm);
                    rot = Quaternion<T>(c, r0 * s, r1 * s, r2 * s);
                }
                *this = DualQuaternion(Quaternion<T>(0, x, y, z), rot);
            }
            const T* raw() const { return &rotation_.w_; }         // 8 contiguous values: rotation_, translation_
        private:
            Quaternion<T> rotation_;
            Quaternion<T> translation_;
        };
        static_assert(sizeof(DualQuaternion<float>) == 32, "DualQuaternion<float> must be {rotation_, translation_}");
    }
}
