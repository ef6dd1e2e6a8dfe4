// CUDA-on-the-host shim: just enough of the CUDA language and runtime for the reference's OWN kernel
// sources -- kfusion/src/cuda/tsdf_volume.cu, imgproc.cu, proj_icp.cu with device.hpp, temp_utils.hpp,
// texture_binder.hpp, internal.hpp, kfusion/cuda/*.hpp and kfusion/src/device_memory.cpp -- to be
// compiled by g++ from where they lie under /root/reference and executed on the CPU, one host loop
// iteration per CUDA thread.  TEST INFRASTRUCTURE ONLY (oracle/_ref build, see oracle/Makefile);
// never shipped, never loaded by the product.
//
// What stands in for the GPU:
//   * "device memory" is host memory (cudaMalloc = calloc); pitched allocations get a 512-byte aligned
//     pitch plus slack rows, because several reference kernels guard with `x < cols || y < rows`.
//   * a kernel launch `k<<<grid, block>>>(args)` is rewritten by the Makefile's sed line into
//     SHIM_LAUNCH(k, (grid, block), (args)), which calls the kernel body once per (block, thread) with
//     blockIdx / threadIdx / blockDim / gridDim set.  Kernels that use __syncthreads or warp intrinsics run
//     their block's threads as ucontext fibers (cuda_shim::launch, see below).
//   * intrinsics map to their IEEE meaning: __fmaf_rn = fmaf, __fsqrt_rn = sqrtf, __float2half_rn = F16C
//     RN-even conversion, __float2int_rn = round-half-even, __float2int_rd = floor.  The *approximate*
//     NVIDIA intrinsics have no CPU equivalent and get the correctly rounded operation (the same stand-ins
//     the restatement in oracle/dfusion_oracle.c uses): __fdividef(a,b) = a/b, rsqrt(x) = 1/sqrtf(x),
//     __expf(x) = (float)exp((double)x), __sinf/__cosf = sinf/cosf.
//   * texture<T,2>: point filtering, unnormalised coordinates: texel (floor(x), floor(y)); border mode
//     returns 0 outside, clamp mode clamps; a half channel descriptor reads IEEE binary16.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <stdint.h>
#include <stddef.h>
#include <immintrin.h>
#include <functional>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __shared__ static thread_local
#define __constant__ static
// PTX inline asm (ld.global.cs / %laneid) only sits behind __CUDA_ARCH__ tests or in helpers the Makefile patches
#define asm(...) ((void)0)

// ---------------------------------------------------------------------------------------- vector types
struct char2 { signed char x, y; };
struct uchar2 { unsigned char x, y; };
struct uchar3 { unsigned char x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
struct short2 { short x, y; };
struct ushort2 { unsigned short x, y; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct uint3 { unsigned int x, y, z; };
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct dim3
{
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};

static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { uchar4 v = {x, y, z, w}; return v; }
static inline ushort2 make_ushort2(unsigned short x, unsigned short y) { ushort2 v = {x, y}; return v; }
static inline int2 make_int2(int x, int y) { int2 v = {x, y}; return v; }
static inline int3 make_int3(int x, int y, int z) { int3 v = {x, y, z}; return v; }
static inline float2 make_float2(float x, float y) { float2 v = {x, y}; return v; }
static inline float3 make_float3(float x, float y, float z) { float3 v = {x, y, z}; return v; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 v = {x, y, z, w}; return v; }

// ---------------------------------------------------------------------------------------- thread geometry
namespace cuda_shim
{
    extern thread_local uint3 tl_threadIdx, tl_blockIdx;
    extern thread_local dim3 tl_blockDim, tl_gridDim;
}
#define threadIdx (::cuda_shim::tl_threadIdx)
#define blockIdx  (::cuda_shim::tl_blockIdx)
#define blockDim  (::cuda_shim::tl_blockDim)
#define gridDim   (::cuda_shim::tl_gridDim)

// ---------------------------------------------------------------------------------------- intrinsics
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float __fdividef(float a, float b) { return a / b; }          // stand-in: IEEE division
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }             // stand-in: correctly rounded 1/sqrt
static inline float rsqrt(float a) { return 1.0f / sqrtf(a); }
// (glibc's <math.h> declares __expf/__sinf/__cosf/__powf as its own internal aliases, hence the macros)
static inline float cuda_shim_expf(float a) { return (float)exp((double)a); }   // stand-in
static inline float cuda_shim_sinf(float a) { return sinf(a); }
static inline float cuda_shim_cosf(float a) { return cosf(a); }
static inline float cuda_shim_powf(float a, float b) { return (float)pow((double)a, (double)b); }   // stand-in
#define __expf(a) cuda_shim_expf(a)
#define __sinf(a) cuda_shim_sinf(a)
#define __cosf(a) cuda_shim_cosf(a)
#define __powf(a, b) cuda_shim_powf(a, b)
static inline float __saturatef(float a) { return a != a ? 0.f : (a < 0.f ? 0.f : (a > 1.f ? 1.f : a)); }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned short __float2half_rn(float f) { return (unsigned short)_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
static inline float __half2float(unsigned short h) { return _cvtsh_ss(h); }
static inline int __float2int_rn(float f) { return (int)lrintf(f); }        // default rounding mode = nearest even
static inline int __float2int_rd(float f) { return (int)floorf(f); }
static inline int __float2int_rz(float f) { return (int)f; }
static inline int __popc(unsigned int v) { return __builtin_popcount(v); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
static inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

// ---------------------------------------------------------------------------------------- runtime
typedef int cudaError_t;
enum { cudaSuccess = 0 };
typedef void* cudaStream_t;
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
static inline const char* cudaGetErrorString(cudaError_t) { return "cuda_shim error"; }
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
enum cudaFuncCache { cudaFuncCachePreferNone, cudaFuncCachePreferShared, cudaFuncCachePreferL1 };
template <class F> static inline cudaError_t cudaFuncSetCacheConfig(F, cudaFuncCache) { return cudaSuccess; }
static inline cudaError_t cudaMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? cudaSuccess : 2; }
static inline cudaError_t cudaMallocHost(void** p, size_t n) { return cudaMalloc(p, n); }
static inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaFreeHost(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaMallocPitch(void** p, size_t* pitch, size_t width_bytes, size_t height)
{
    *pitch = ((width_bytes + 511) / 512) * 512 + 512;       // slack columns for `||`-guarded kernels
    *p = calloc(*pitch * (height + 8), 1);                   // slack rows likewise
    return *p ? cudaSuccess : 2;
}
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind, cudaStream_t = 0) { memcpy(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, cudaMemcpyKind)
{
    for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
    return cudaSuccess;
}
#define cudaMemcpyFromSymbol(dst, sym, n) (memcpy((dst), &(sym), (n)), cudaSuccess)
#define cudaMemcpyToSymbol(sym, src, n) (memcpy(&(sym), (src), (n)), cudaSuccess)

// ---------------------------------------------------------------------------------------- textures
enum cudaTextureReadMode { cudaReadModeElementType = 0, cudaReadModeNormalizedFloat = 1 };
enum cudaTextureFilterMode { cudaFilterModePoint = 0, cudaFilterModeLinear = 1 };
enum cudaTextureAddressMode { cudaAddressModeWrap = 0, cudaAddressModeClamp = 1, cudaAddressModeMirror = 2, cudaAddressModeBorder = 3 };
enum cudaChannelFormatKind { cudaChannelFormatKindSigned = 0, cudaChannelFormatKindUnsigned = 1, cudaChannelFormatKindFloat = 2 };
struct cudaChannelFormatDesc { int x, y, z, w; cudaChannelFormatKind f; };
static inline cudaChannelFormatDesc cudaCreateChannelDesc(int x, int y, int z, int w, cudaChannelFormatKind f) { cudaChannelFormatDesc d = {x, y, z, w, f}; return d; }
static inline cudaChannelFormatDesc cudaCreateChannelDescHalf() { return cudaCreateChannelDesc(16, 0, 0, 0, cudaChannelFormatKindFloat); }
template <class T> static inline cudaChannelFormatDesc cudaCreateChannelDesc() { return cudaCreateChannelDesc((int)sizeof(T) * 8, 0, 0, 0, cudaChannelFormatKindSigned); }
template <> inline cudaChannelFormatDesc cudaCreateChannelDesc<float>() { return cudaCreateChannelDesc(32, 0, 0, 0, cudaChannelFormatKindFloat); }
template <> inline cudaChannelFormatDesc cudaCreateChannelDesc<unsigned short>() { return cudaCreateChannelDesc(16, 0, 0, 0, cudaChannelFormatKindUnsigned); }
template <> inline cudaChannelFormatDesc cudaCreateChannelDesc<float4>() { return cudaCreateChannelDesc(32, 32, 32, 32, cudaChannelFormatKindFloat); }

struct textureReference
{
    int normalized;
    cudaTextureFilterMode filterMode;
    cudaTextureAddressMode addressMode[3];
    cudaChannelFormatDesc channelDesc;
    // binding
    mutable const void* bound_ptr;
    mutable size_t bound_cols, bound_rows, bound_pitch;
    mutable cudaChannelFormatDesc bound_desc;
};
template <class T, int dim = 1, cudaTextureReadMode mode = cudaReadModeElementType>
struct texture : public textureReference
{
    texture(int norm = 0, cudaTextureFilterMode fm = cudaFilterModePoint, cudaTextureAddressMode am = cudaAddressModeClamp)
    {
        normalized = norm; filterMode = fm; addressMode[0] = addressMode[1] = addressMode[2] = am;
        channelDesc = cudaCreateChannelDesc<T>(); bound_ptr = 0; bound_cols = bound_rows = bound_pitch = 0; bound_desc = channelDesc;
    }
    texture(int norm, cudaTextureFilterMode fm, cudaTextureAddressMode am, cudaChannelFormatDesc desc)
    {
        normalized = norm; filterMode = fm; addressMode[0] = addressMode[1] = addressMode[2] = am;
        channelDesc = desc; bound_ptr = 0; bound_cols = bound_rows = bound_pitch = 0; bound_desc = desc;
    }
};
static inline cudaError_t cudaBindTexture2D(size_t* offset, const textureReference& tex, const void* ptr, const cudaChannelFormatDesc& desc,
                                            size_t cols, size_t rows, size_t pitch)
{
    if (offset) *offset = 0;
    tex.bound_ptr = ptr; tex.bound_cols = cols; tex.bound_rows = rows; tex.bound_pitch = pitch; tex.bound_desc = desc;
    return cudaSuccess;
}
static inline cudaError_t cudaBindTexture(size_t* offset, const textureReference& tex, const void* ptr, const cudaChannelFormatDesc& desc, size_t bytes)
{
    if (offset) *offset = 0;
    tex.bound_ptr = ptr; tex.bound_cols = bytes; tex.bound_rows = 1; tex.bound_pitch = bytes; tex.bound_desc = desc;
    return cudaSuccess;
}
static inline cudaError_t cudaUnbindTexture(const textureReference* tex) { tex->bound_ptr = 0; return cudaSuccess; }

namespace cuda_shim
{
    // point-filtered, unnormalised fetch: texel index = floor(coordinate)
    template <class T> static inline bool texel_address(const textureReference& t, float x, float y, const T*& out)
    {
        if (!(x == x) || !(y == y)) { if (t.addressMode[0] == cudaAddressModeBorder) return false; x = y = 0.f; }
        float fx = floorf(x), fy = floorf(y);
        long ix, iy;
        if (t.addressMode[0] == cudaAddressModeBorder)
        {
            if (fx < 0.f || fy < 0.f || fx >= (float)t.bound_cols || fy >= (float)t.bound_rows) return false;
            ix = (long)fx; iy = (long)fy;
        }
        else
        {
            ix = fx < 0.f ? 0 : (fx >= (float)t.bound_cols ? (long)t.bound_cols - 1 : (long)fx);
            iy = fy < 0.f ? 0 : (fy >= (float)t.bound_rows ? (long)t.bound_rows - 1 : (long)fy);
        }
        out = (const T*)((const char*)t.bound_ptr + iy * t.bound_pitch) + ix;
        return true;
    }
}
template <cudaTextureReadMode m> static inline float tex2D(const texture<float, 2, m>& t, float x, float y)
{
    if (t.bound_desc.x == 16)   // half channel: IEEE binary16 texels read as float
    {
        const unsigned short* p;
        return cuda_shim::texel_address(t, x, y, p) ? _cvtsh_ss(*p) : 0.f;
    }
    const float* p;
    return cuda_shim::texel_address(t, x, y, p) ? *p : 0.f;
}
template <cudaTextureReadMode m> static inline unsigned short tex2D(const texture<unsigned short, 2, m>& t, float x, float y)
{
    const unsigned short* p;
    return cuda_shim::texel_address(t, x, y, p) ? *p : (unsigned short)0;
}
template <cudaTextureReadMode m> static inline float4 tex2D(const texture<float4, 2, m>& t, float x, float y)
{
    const float4* p;
    if (cuda_shim::texel_address(t, x, y, p)) return *p;
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------------------- launches
namespace cuda_shim
{
    struct LaunchCfg
    {
        dim3 grid, block;
        LaunchCfg(dim3 g, dim3 b, size_t = 0, cudaStream_t = 0) : grid(g), block(b) {}
    };
    // mode for the NEXT launches (set by the glue): 0 = plain nested loops (kernels whose threads are independent),
    // 1 = one fiber per thread of a block, scheduled in thread order, switching at __syncthreads and the
    //     warp-level primitives below (kernels with barriers / warp-synchronous code).
    extern int g_fiber_mode;
    extern int g_parallel_blocks;        // 1: OpenMP over blocks in mode 0 (independent threads only)
    void launch(const LaunchCfg& cfg, const std::function<void()>& body);
    void sync_threads();                 // block-wide barrier (fiber mode)
    void sync_warp();                    // barrier over the 32 fibers of the calling thread's warp
    unsigned int warp_ballot(int pred);  // bit i = predicate of lane i
    unsigned int lane_id();
}
#define SHIM_LAUNCH(kern, cfg, args) ::cuda_shim::launch(::cuda_shim::LaunchCfg cfg, [&]() { kern args; })
static inline void __syncthreads() { ::cuda_shim::sync_threads(); }
static inline unsigned int __ballot(int pred) { return ::cuda_shim::warp_ballot(pred); }
static inline int __all(int pred) { return ::cuda_shim::warp_ballot(pred) == 0xffffffffu; }
static inline int __any(int pred) { return ::cuda_shim::warp_ballot(pred) != 0u; }
// sequential emulation: blocks/warps never run concurrently in fiber mode, so plain read-modify-write is atomic
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned int atomicInc(unsigned int* p, unsigned int lim) { unsigned int o = *p; *p = (o >= lim) ? 0 : o + 1; return o; }
