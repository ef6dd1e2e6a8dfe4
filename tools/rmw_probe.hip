// tools/rmw_probe.hip -- what the ACCESS PATTERN of the rigid sweep costs, without its arithmetic.
// The sweep read-modify-writes 4-byte voxels of a 512^3 volume; a wave owns a patch of 64 columns (PX wide, 64 / PX rows) and walks
// planes (1 MiB apart), a workgroup of W waves owns W patches side by side in x.  Per plane a workgroup therefore touches 64 / PX
// rows x (W * PX * 4) contiguous bytes, and at any instant thousands of workgroups are at unrelated places of the volume.  Round 3
// found the sweep stuck at ~100 us whatever its instruction count or the number of voxels in flight; this probe measures the rate of
// exactly that traffic for each patch shape / workgroup width, RMW of every voxel of a fraction `fill` of the patches' planes.
//   hipcc --offload-arch=gfx950 -O3 tools/rmw_probe.hip -o build/rmw_probe && build/rmw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// item = (workgroup tile, chunk of ZC planes); tiles are taken in a scrambled order (the sweep's plan sorts by work, not by place)
template <int PX, int W, int U>
__global__ __launch_bounds__(W * 64) void rmw_kernel(unsigned* __restrict__ vol, int X, int Y, int Z, int ZC, unsigned n_items, unsigned mul)
{
    constexpr int PY = 64 / PX;
    const unsigned item = (unsigned)(((unsigned long long)blockIdx.x * mul) % n_items);
    const int tiles_x = X / (PX * W), tiles_y = Y / PY;
    const int tile = item % (tiles_x * tiles_y), chunk = item / (tiles_x * tiles_y);
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int x = (tx * W + wave) * PX + (lane % PX), y = ty * PY + lane / PX;
    const size_t plane = (size_t)X * Y;
    unsigned* p = vol + (size_t)chunk * ZC * plane + (size_t)y * X + x;
    for (int z = 0; z < ZC; z += U) {
        unsigned v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(size_t)u * plane];
#pragma unroll
        for (int u = 0; u < U; ++u) p[(size_t)u * plane] = v[u] + 0x10000u;
        p += (size_t)U * plane;
    }
}

template <int PX, int W, int U>
static void run(unsigned* vol, int N, int ZC, double frac)
{
    constexpr int PY = 64 / PX;
    const unsigned tiles = (unsigned)((N / (PX * W)) * (N / PY)), n_items = tiles * (unsigned)(N / ZC);
    const unsigned launch = (unsigned)(n_items * frac);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const unsigned mul = 2654435761u % n_items | 1u;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((rmw_kernel<PX, W, U>), dim3(launch), dim3(W * 64), 0, 0, vol, N, N, N, ZC, n_items, mul);
    CK(hipEventRecord(a, 0));
    const int iters = 10;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((rmw_kernel<PX, W, U>), dim3(launch), dim3(W * 64), 0, 0, vol, N, N, N, ZC, n_items, mul);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
    const double bytes = 8.0 * (double)launch * W * 64 * ZC;
    printf("  patch %2d x %d, %2d waves/WG (row run %5d B), U=%d, chunk %3d: %.3f ms  %.0f GB/s (r+w)\n", PX, PY, W, PX * W * 4, U, ZC, ms, bytes / ms / 1e6);
}

int main()
{
    const int N = 512;
    unsigned* vol; CK(hipMalloc(&vol, (size_t)N * N * N * 4)); CK(hipMemset(vol, 0, (size_t)N * N * N * 4));
    const double frac = 0.3;      // the sweep updates ~23 % of the volume
    printf("RMW of %.0f %% of a %d^3 volume of 4-byte voxels, scattered (patch, chunk) items\n", frac * 100, N);
    run<16, 1, 2>(vol, N, 64, frac); run<16, 4, 2>(vol, N, 64, frac); run<16, 8, 2>(vol, N, 64, frac); run<16, 16, 2>(vol, N, 64, frac);
    run<32, 1, 2>(vol, N, 64, frac); run<32, 4, 2>(vol, N, 64, frac); run<32, 8, 2>(vol, N, 64, frac); run<32, 16, 2>(vol, N, 64, frac);
    run<64, 1, 2>(vol, N, 64, frac); run<64, 2, 2>(vol, N, 64, frac); run<64, 4, 2>(vol, N, 64, frac); run<64, 8, 2>(vol, N, 64, frac);
    printf("more planes in flight per lane\n");
    run<16, 1, 4>(vol, N, 64, frac); run<16, 1, 8>(vol, N, 64, frac); run<32, 1, 4>(vol, N, 64, frac); run<32, 1, 8>(vol, N, 64, frac);
    run<64, 1, 4>(vol, N, 64, frac); run<64, 1, 8>(vol, N, 64, frac); run<64, 8, 4>(vol, N, 64, frac); run<64, 8, 8>(vol, N, 64, frac);
    run<16, 8, 4>(vol, N, 64, frac); run<16, 8, 8>(vol, N, 64, frac);
    printf("shorter / longer chunks\n");
    run<16, 1, 2>(vol, N, 16, frac); run<16, 1, 2>(vol, N, 32, frac); run<16, 1, 2>(vol, N, 128, frac);
    run<64, 8, 2>(vol, N, 16, frac); run<64, 8, 2>(vol, N, 32, frac); run<64, 8, 2>(vol, N, 128, frac);
    printf("whole volume (frac 1)\n");
    run<16, 1, 2>(vol, N, 64, 1.0); run<64, 8, 2>(vol, N, 64, 1.0); run<64, 8, 8>(vol, N, 64, 1.0);
    return 0;
}
