// kfusion/cuda/imgproc.hpp -- the one image op on the hot path (the rest of imgproc is out of scope).
#pragma once
#include <kfusion/types.hpp>
namespace kfusion
{
    namespace cuda
    {
        /// depth mm -> ray length metres as half bits (/root/reference/kfusion/src/imgproc.cpp:87-91)
        void computeDists(const Depth& depth, Dists& dists, const Intr& intr);
        /// cudaDeviceSynchronize stand-in (imgproc.cpp:41-44)
        void waitAllDefaultStream();
    }
}
