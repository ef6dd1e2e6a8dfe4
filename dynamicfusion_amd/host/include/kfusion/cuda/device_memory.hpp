// kfusion/cuda/device_memory.hpp -- HIP-backed device blobs with the reference's names and semantics
// (/root/reference/kfusion/include/kfusion/cuda/device_memory.hpp): reference counted, copy = share,
// create() reallocates only on size change, user-pointer wrapping disables ownership.  Plumbing for the drop-in
// boundary, not a feature: ownership is a std::shared_ptr instead of the CV_XADD counter.
#pragma once
#include <cstddef>
#include <memory>

namespace kfusion
{
    namespace cuda
    {
        /// Error hook: every failed HIP / dfusion call lands here.  The reference prints "KinFu2 error: ..." and
        /// exit(0)s (device_memory.cpp:7-11); this one prints the same line and exits with status 1.
        void error(const char* error_string, const char* file, const int line, const char* func = "");

        class DeviceMemory
        {
        public:
            DeviceMemory() : data_(nullptr), sizeBytes_(0) {}
            explicit DeviceMemory(size_t sizeBytes) : data_(nullptr), sizeBytes_(0) { create(sizeBytes); }
            DeviceMemory(void* ptr, size_t sizeBytes) : data_(ptr), sizeBytes_(sizeBytes) {}   // not owned
            void create(size_t sizeBytes);
            void release() { owner_.reset(); data_ = nullptr; sizeBytes_ = 0; }
            void copyTo(DeviceMemory& other) const;
            void upload(const void* host_ptr, size_t sizeBytes);
            void download(void* host_ptr) const;
            void swap(DeviceMemory& other) { owner_.swap(other.owner_); std::swap(data_, other.data_); std::swap(sizeBytes_, other.sizeBytes_); }
            template <class T> T* ptr() { return static_cast<T*>(data_); }
            template <class T> const T* ptr() const { return static_cast<const T*>(data_); }
            bool empty() const { return !data_; }
            size_t sizeBytes() const { return sizeBytes_; }
        private:
            std::shared_ptr<void> owner_;
            void* data_;
            size_t sizeBytes_;
        };

        class DeviceMemory2D
        {
        public:
            DeviceMemory2D() : data_(nullptr), step_(0), colsBytes_(0), rows_(0) {}
            DeviceMemory2D(int rows, int colsBytes) : data_(nullptr), step_(0), colsBytes_(0), rows_(0) { create(rows, colsBytes); }
            DeviceMemory2D(int rows, int colsBytes, void* data, size_t stepBytes) : data_(data), step_(stepBytes), colsBytes_(colsBytes), rows_(rows) {}
            void create(int rows, int colsBytes);                 // hipMallocPitch
            void release() { owner_.reset(); data_ = nullptr; step_ = 0; colsBytes_ = rows_ = 0; }
            void copyTo(DeviceMemory2D& other) const;
            void upload(const void* host_ptr, size_t host_step, int rows, int colsBytes);
            void download(void* host_ptr, size_t host_step) const;
            void swap(DeviceMemory2D& o) { owner_.swap(o.owner_); std::swap(data_, o.data_); std::swap(step_, o.step_); std::swap(colsBytes_, o.colsBytes_); std::swap(rows_, o.rows_); }
            template <class T> T* ptr(int y = 0) { return reinterpret_cast<T*>(static_cast<char*>(data_) + (size_t)y * step_); }
            template <class T> const T* ptr(int y = 0) const { return reinterpret_cast<const T*>(static_cast<const char*>(data_) + (size_t)y * step_); }
            bool empty() const { return !data_; }
            int colsBytes() const { return colsBytes_; }
            int rows() const { return rows_; }
            size_t step() const { return step_; }
        private:
            std::shared_ptr<void> owner_;
            void* data_;
            size_t step_;
            int colsBytes_, rows_;
        };
    }
}
