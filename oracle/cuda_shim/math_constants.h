// part of the CUDA-on-the-host shim (TEST INFRASTRUCTURE ONLY): everything lives in cuda_runtime_api.h
#pragma once
#include "cuda_runtime_api.h"
#define CUDART_NAN_F __int_as_float(0x7fffffff)
#define CUDART_INF_F __int_as_float(0x7f800000)
