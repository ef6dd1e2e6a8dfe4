// TEST INFRASTRUCTURE (see cuda_runtime_api.h): make_float2 & co come with the HIP vector types.
#pragma once
#include <hip/hip_vector_types.h>
