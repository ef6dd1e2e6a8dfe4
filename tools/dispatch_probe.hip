// How fast does the GPU start the workgroups of a launch?  tools/dispatch_probe.hip: every workgroup stamps the 100 MHz clock when its
// first wave starts and then spins for ~60 us (so that none retires during the ramp); the host prints when the 1st, 128th, 256th, 512th
// ... workgroup started, for workgroup sizes 256 / 512 / 1024 and dynamic LDS of 0 / 32 / 64 KiB -- the warped sweep launches 2000
// workgroups of 512 threads with 64 KiB of LDS each, two to a CU, and its per-wave timeline shows a 30-40 us ramp.
//   hipcc --offload-arch=gfx950 -O3 tools/dispatch_probe.hip -o build/dispatch_probe && build/dispatch_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned long long* start, unsigned spin_ticks, int touch_lds)
{
    extern __shared__ unsigned int lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    if (touch_lds) lds[threadIdx.x] = threadIdx.x;
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (touch_lds && lds[(threadIdx.x + 1) % blockDim.x] == 0xffffffffu) start[blockIdx.x] = 0;
}
// the same with the registers of the sweep kernel: 113 VGPRs (v112 touched) -- 4 waves per SIMD, two 512-thread workgroups per CU at most
__global__ __launch_bounds__(512) void probe_vgpr(unsigned long long* start, unsigned spin_ticks, int touch_lds)
{
    extern __shared__ unsigned int lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    asm volatile("v_mov_b32 v112, 0" ::: "v112");
    if (touch_lds) lds[threadIdx.x] = threadIdx.x;
    while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (touch_lds && lds[(threadIdx.x + 1) % blockDim.x] == 0xffffffffu) start[blockIdx.x] = 0;
}
int main()
{
    const int n_wg = 2048;
    unsigned long long* d; hipMalloc(&d, n_wg * 8);
    std::vector<unsigned long long> h(n_wg);
    for (int threads : {256, 512, 1024})
        for (int lds_kb : {0, 32, 64}) {
            hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(d, 0, n_wg * 8);
                hipLaunchKernelGGL(probe, dim3(n_wg), dim3(threads), lds_kb * 1024, 0, d, 6000u, lds_kb ? 1 : 0);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), d, n_wg * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("threads %4d  LDS %2d KiB: start of workgroup #1 / 128 / 256 / 512 / 1024 / 2048 at", threads, lds_kb);
            for (int k : {1, 128, 256, 512, 1024, 2048}) printf(" %7.2f", (double)(h[k - 1] - h[0]) * 0.01);
            printf(" us\n");
        }
    hipFuncSetAttribute((const void*)probe_vgpr, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int lds_b : {0, 64000}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d, 0, n_wg * 8);
            hipLaunchKernelGGL(probe_vgpr, dim3(n_wg), dim3(512), lds_b, 0, d, 6000u, lds_b ? 1 : 0);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d, n_wg * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("113 VGPRs, threads 512, LDS %5d B: start of workgroup #1 / 128 / 256 / 400 / 471 / 512 / 513 at", lds_b);
        for (int k : {1, 128, 256, 400, 471, 512, 513}) printf(" %7.2f", (double)(h[k - 1] - h[0]) * 0.01);
        printf(" us\n");
    }
    return 0;
}
