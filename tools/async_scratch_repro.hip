// Minimal repro attempt of round 3's incident: dfusion_integrate took its launch-plan scratch from the runtime's
// stream-ordered allocator (hipMallocAsync / hipFreeAsync per call); in a process that ALSO hipMalloc'ed / hipFree'd between the calls,
// one integrate in ~30 updated a different set of voxels.  Same shape here: a "plan" kernel writes a pattern into stream-allocated
// scratch, a "sweep" kernel reads it back a few launches later, the scratch is freed on the stream, and the host interleaves
// synchronous hipMalloc / hipMemset / hipFree of varying sizes (hipFree synchronises the device and lets the pool trim).
//   hipcc --offload-arch=gfx950 -O2 tools/async_scratch_repro.hip -o build/async_scratch_repro && build/async_scratch_repro [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s failed: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)

__global__ void plan_kernel(unsigned* s, size_t n, unsigned tag)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s[i] = (unsigned)i * 2654435761u ^ tag;
}
__global__ void spin_kernel(unsigned* sink, int rounds)                  // something else on the stream between plan and sweep
{
    unsigned v = threadIdx.x;
    for (int i = 0; i < rounds; ++i) v = v * 1664525u + 1013904223u;
    if (v == 12345u) *sink = v;
}
__global__ void sweep_kernel(const unsigned* s, size_t n, unsigned tag, unsigned long long* bad)
{
    unsigned long long b = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b += s[i] != ((unsigned)i * 2654435761u ^ tag);
    if (b) atomicAdd(bad, b);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned long long* bad; CK(hipMalloc((void**)&bad, 8)); CK(hipMemset(bad, 0, 8));
    unsigned* sink; CK(hipMalloc((void**)&sink, 4));
    srand(1);
    unsigned long long total_bad = 0; int bad_iters = 0;
    for (int it = 0; it < iters; ++it) {
        const size_t n = (size_t)(16 + rand() % 48) << 18;               // 16-64 MiB of plan, as dfusion_integrate's at 512^3
        unsigned* s = nullptr;
        CK(hipMallocAsync((void**)&s, n * 4, st));
        plan_kernel<<<1024, 256, 0, st>>>(s, n, (unsigned)it);
        spin_kernel<<<64, 256, 0, st>>>(sink, 2000);
        // the host's own allocations between the enqueue of the plan and of the sweep, and after: synchronous, on the null stream
        void* h = nullptr; const size_t hb = (size_t)(1 + rand() % 64) << 20;
        if (it % 3 != 2) { CK(hipMalloc(&h, hb)); CK(hipMemset(h, 0xAB, hb)); }
        sweep_kernel<<<1024, 256, 0, st>>>(s, n, (unsigned)it, bad);
        CK(hipFreeAsync(s, st));
        if (h) CK(hipFree(h));                                           // device-synchronising: the pool may trim here
        if (it % 5 == 4) { void* g; CK(hipMalloc(&g, n * 4)); CK(hipMemset(g, 0, n * 4)); CK(hipFree(g)); }   // same size as the scratch just freed
        if (it % 50 == 49) {
            CK(hipStreamSynchronize(st));
            unsigned long long b; CK(hipMemcpy(&b, bad, 8, hipMemcpyDeviceToHost));
            if (b) { ++bad_iters; total_bad += b; printf("iterations %d-%d: %llu words of the plan read back wrong\n", it - 49, it, b); CK(hipMemset(bad, 0, 8)); }
        }
    }
    CK(hipStreamSynchronize(st));
    printf("%d iterations: %llu wrong words in %d of %d checked windows -> %s\n", iters, total_bad, bad_iters, iters / 50,
           total_bad ? "REPRODUCED: stream-ordered scratch is not safe beside synchronous hipMalloc / hipFree on this runtime"
                     : "not reproduced: the stream-ordered allocation itself held");
    return total_bad ? 1 : 0;
}
