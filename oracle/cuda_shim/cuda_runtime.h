// part of the CUDA-on-the-host shim (TEST INFRASTRUCTURE ONLY): everything lives in cuda_runtime_api.h
#pragma once
#include "cuda_runtime_api.h"
