// tests/ref_host_bridge/shim/cuda_runtime_api.h -- TEST INFRASTRUCTURE.  The reference's HOST sources (kfusion/src/safe_call.hpp,
// device_memory.cpp, precomp.cpp, tsdf_volume.cpp, imgproc.cpp) name a dozen CUDA runtime entry points and the CUDA vector types.
// This header gives those names their HIP equivalents so that the files compile UNMODIFIED, where they lie under /root/reference, with
// the host compiler against ROCm -- which is all a maintainer's port of the host layer amounts to.  It is
// never part of the product library.
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaStream_t hipStream_t
#define cudaMalloc hipMalloc
#define cudaMallocPitch hipMallocPitch
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemcpy2D hipMemcpy2D
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice
#define cudaDeviceSynchronize hipDeviceSynchronize
#define cudaGetLastError hipGetLastError
#define cudaStreamSynchronize hipStreamSynchronize
