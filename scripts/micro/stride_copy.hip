// Micro-benchmark: what does the HBM/L2 system deliver for the column pass' access pattern?
// A [H][WS] array of 16-byte elements is read and written back in place by workgroups that each own TC
// adjacent columns (all H rows): the pattern of pass B without any arithmetic. Also a contiguous copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int TC, int THREADS, bool XCD>
__global__ void __launch_bounds__(THREADS) col_copy(double2* __restrict__ ws, int H, int wstride, double scale) {
    constexpr int T = THREADS / TC;           // threads per column
    const int c = threadIdx.x % TC, t = threadIdx.x / TC;
    int grp = blockIdx.x;
    if (XCD) { const int ng = gridDim.x; grp = (blockIdx.x & 7) * (ng >> 3) + (blockIdx.x >> 3); }
    const int jp = grp * TC + c;
    const int E = H / T;                      // 16
    double2 v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = ws[(unsigned)(t + T * m) * (unsigned)wstride + jp];
#pragma unroll
    for (int m = 0; m < 16; ++m) { v[m].x *= scale; v[m].y *= scale; }
#pragma unroll
    for (int m = 0; m < 16; ++m) ws[(unsigned)(t + T * m) * (unsigned)wstride + jp] = v[m];
    (void)E;
}

__global__ void __launch_bounds__(256) lin_copy(double2* __restrict__ ws, size_t n, double scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double2 v = ws[i]; v.x *= scale; v.y *= scale; ws[i] = v;
    }
}

template <int TC, int THREADS, bool XCD> void run(double2* ws, int H, int NC, int wstride, const char* name) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = NC / TC;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((col_copy<TC, THREADS, XCD>), dim3(grid), dim3(THREADS), 0, 0, ws, H, wstride, 1.0);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((col_copy<TC, THREADS, XCD>), dim3(grid), dim3(THREADS), 0, 0, ws, H, wstride, 1.0);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 2.0 * H * (double)NC * 16.0;
    printf("%-28s TC=%2d threads=%4d xcd=%d : %8.1f us  %7.1f GB/s\n", name, TC, THREADS, (int)XCD, 1e3 * ms / reps, bytes / (1e-3 * ms / reps) / 1e9);
}

// same measurement, but the Infinity Cache is flushed (1 GiB streamed) before every launch
template <int TC, int THREADS, bool XCD> void run_cold(double2* ws, double2* big, size_t nbig, int H, int NC, int wstride) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = NC / TC;
    double tot = 0;
    const int reps = 6;
    for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(lin_copy, dim3(2048), dim3(256), 0, 0, big, nbig, 1.0);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((col_copy<TC, THREADS, XCD>), dim3(grid), dim3(THREADS), 0, 0, ws, H, wstride, 1.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i > 0) tot += ms;
    }
    const double bytes = 2.0 * H * (double)NC * 16.0, sec = 1e-3 * tot / (reps - 1);
    printf("COLD (cache flushed)         TC=%2d threads=%4d xcd=%d : %8.1f us  %7.1f GB/s\n", TC, THREADS, (int)XCD, 1e6 * sec, bytes / sec / 1e9);
}

int main() {
    const int H = 4096, NC = 2048, wstride = NC + 8;
    double2* ws; CK(hipMalloc(&ws, sizeof(double2) * (size_t)H * wstride));
    CK(hipMemset(ws, 0, sizeof(double2) * (size_t)H * wstride));
    // contiguous reference
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const size_t n = (size_t)H * wstride;
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(lin_copy, dim3(2048), dim3(256), 0, 0, ws, n, 1.0);
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(lin_copy, dim3(2048), dim3(256), 0, 0, ws, n, 1.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("contiguous in-place copy: %8.1f us  %7.1f GB/s\n", 1e3 * ms / 20, 2.0 * n * 16 / (1e-3 * ms / 20) / 1e9);
    }
    run<1, 256, true>(ws, H, NC, wstride, "col tile");
    run<2, 512, true>(ws, H, NC, wstride, "col tile");
    run<2, 512, false>(ws, H, NC, wstride, "col tile");
    run<4, 1024, true>(ws, H, NC, wstride, "col tile");
    run<4, 1024, false>(ws, H, NC, wstride, "col tile");
    run<8, 1024, true>(ws, H, NC, wstride, "col tile (E=32 not 16!)");
    double2* big; const size_t nbig = (size_t)64 << 20; // 1 GiB
    CK(hipMalloc(&big, sizeof(double2) * nbig)); CK(hipMemset(big, 0, sizeof(double2) * nbig));
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(lin_copy, dim3(2048), dim3(256), 0, 0, big, nbig, 1.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("contiguous in-place copy of 1 GiB (HBM): %8.1f us  %7.1f GB/s\n", 1e3 * ms / 5, 2.0 * nbig * 16 / (1e-3 * ms / 5) / 1e9);
        // cold contiguous copy of the workspace
        double tot = 0;
        for (int i = 0; i < 6; ++i) {
            hipLaunchKernelGGL(lin_copy, dim3(2048), dim3(256), 0, 0, big, nbig, 1.0);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(lin_copy, dim3(2048), dim3(256), 0, 0, ws, (size_t)H * wstride, 1.0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (i > 0) tot += ms;
        }
        printf("COLD contiguous copy of the 128 MiB workspace: %8.1f us  %7.1f GB/s\n", 1e3 * tot / 5, 2.0 * H * wstride * 16 / (1e-3 * tot / 5) / 1e9);
    }
    run_cold<1, 256, true>(ws, big, nbig, H, NC, wstride);
    run_cold<2, 512, true>(ws, big, nbig, H, NC, wstride);
    run_cold<4, 1024, true>(ws, big, nbig, H, NC, wstride);
    run_cold<4, 1024, false>(ws, big, nbig, H, NC, wstride);
    return 0;
}
