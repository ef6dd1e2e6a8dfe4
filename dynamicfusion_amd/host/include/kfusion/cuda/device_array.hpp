// kfusion/cuda/device_array.hpp -- typed views over DeviceMemory / DeviceMemory2D with the reference's names
// (/root/reference/kfusion/include/kfusion/cuda/device_array.hpp:19-232).
#pragma once
#include <vector>
#include <kfusion/cuda/device_memory.hpp>

namespace kfusion
{
    namespace cuda
    {
        template <class T>
        class DeviceArray : public DeviceMemory
        {
        public:
            typedef T type;
            enum { elem_size = sizeof(T) };
            DeviceArray() {}
            explicit DeviceArray(size_t size) : DeviceMemory(size * elem_size) {}
            DeviceArray(T* ptr, size_t size) : DeviceMemory(ptr, size * elem_size) {}
            void create(size_t size) { DeviceMemory::create(size * elem_size); }
            void upload(const T* host_ptr, size_t size) { DeviceMemory::upload(host_ptr, size * elem_size); }
            void download(T* host_ptr) const { DeviceMemory::download(host_ptr); }
            template <class A> void upload(const std::vector<T, A>& data) { upload(data.data(), data.size()); }
            template <class A> void download(std::vector<T, A>& data) const { data.resize(size()); if (!data.empty()) download(data.data()); }
            T* ptr() { return DeviceMemory::ptr<T>(); }
            const T* ptr() const { return DeviceMemory::ptr<T>(); }
            operator T*() { return ptr(); }
            operator const T*() const { return ptr(); }
            size_t size() const { return sizeBytes() / elem_size; }
        };

        template <class T>
        class DeviceArray2D : public DeviceMemory2D
        {
        public:
            typedef T type;
            enum { elem_size = sizeof(T) };
            DeviceArray2D() {}
            DeviceArray2D(int rows, int cols) : DeviceMemory2D(rows, cols * elem_size) {}
            DeviceArray2D(int rows, int cols, void* data, size_t stepBytes) : DeviceMemory2D(rows, cols * elem_size, data, stepBytes) {}
            void create(int rows, int cols) { DeviceMemory2D::create(rows, cols * elem_size); }
            void upload(const void* host_ptr, size_t host_step, int rows, int cols) { DeviceMemory2D::upload(host_ptr, host_step, rows, cols * elem_size); }
            void download(void* host_ptr, size_t host_step) const { DeviceMemory2D::download(host_ptr, host_step); }
            template <class A> void upload(const std::vector<T, A>& data, int cols) { upload(data.data(), cols * elem_size, (int)(data.size() / cols), cols); }
            template <class A> void download(std::vector<T, A>& data, int& elem_step) const
            { elem_step = cols(); data.resize((size_t)cols() * rows()); if (!data.empty()) download(data.data(), (size_t)cols() * elem_size); }
            T* ptr(int y = 0) { return DeviceMemory2D::ptr<T>(y); }
            const T* ptr(int y = 0) const { return DeviceMemory2D::ptr<T>(y); }
            operator T*() { return ptr(); }
            operator const T*() const { return ptr(); }
            int cols() const { return colsBytes() / elem_size; }
            size_t elem_step() const { return step() / elem_size; }
        };
    }
}
