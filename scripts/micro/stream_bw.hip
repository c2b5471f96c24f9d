// Micro-benchmark: what this box's memory system delivers to plain streaming kernels -- read-only, write-only, copy,
// with ordinary and non-temporal accesses, on buffers far larger than the Infinity Cache (2 GiB) and on a buffer that
// fits it (96 MiB: read or copy repeated, so it is served on-die). 16 bytes per lane, grid-stride.
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/stream_bw.hip -o scripts/micro/stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ d2 ld(const d2* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(d2* p, d2 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT> __global__ void __launch_bounds__(256) k_read(const d2* __restrict__ a, size_t n, double* sink) {
    d2 acc = {0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += ld<NT>(a + i);
    if (acc.x == 1.2345e300) sink[0] = acc.y;
}
template <bool NT> __global__ void __launch_bounds__(256) k_write(d2* __restrict__ a, size_t n, double v) {
    const d2 x = {v, v};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) st<NT>(a + i, x);
}
template <bool NTL, bool NTS> __global__ void __launch_bounds__(256) k_copy(const d2* __restrict__ a, d2* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) st<NTS>(b + i, ld<NTL>(a + i));
}
// 4 independent 16-byte loads in flight per thread
template <bool NTL, bool NTS> __global__ void __launch_bounds__(256) k_copy4(const d2* __restrict__ a, d2* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
        const d2 v0 = ld<NTL>(a + i), v1 = ld<NTL>(a + i + stride), v2 = ld<NTL>(a + i + 2 * stride), v3 = ld<NTL>(a + i + 3 * stride);
        st<NTS>(b + i, v0); st<NTS>(b + i + stride, v1); st<NTS>(b + i + 2 * stride, v2); st<NTS>(b + i + 3 * stride, v3);
    }
}

// Persistent grid, block-contiguous chunks, NL 16-byte loads in flight per lane before the first store: each workgroup
// walks chunks of 256 * NL * 16 bytes (whole DRAM pages stay with one workgroup instead of being shared by 64).
template <int NL, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k_copy_chunk(const d2* __restrict__ a, d2* __restrict__ b, size_t n) {
    const size_t chunk = (size_t)256 * NL;
    for (size_t c = blockIdx.x; (c + 1) * chunk <= n; c += gridDim.x) {
        const d2* s = a + c * chunk + threadIdx.x;
        d2* d = b + c * chunk + threadIdx.x;
        d2 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) v[k] = ld<NTL>(s + 256 * k);
#pragma unroll
        for (int k = 0; k < NL; ++k) st<NTS>(d + 256 * k, v[k]);
    }
}
// The access pattern of the column pass: tiles of 8 complex columns (one 128-byte line per row) x 2048 rows out of a
// [rows][ld] complex array, read into registers and written back in place; 512 threads per tile, 32 loads per lane.
template <bool NT> __global__ void __launch_bounds__(512) k_tile_rmw(d2* __restrict__ a, int lda, int ntq, int ntiles) {
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int q = tile / ntq, ct = tile - q * ntq;
        const int c = threadIdx.x & 7, tp = threadIdx.x >> 3; // 64 row positions x 8 columns
        d2* base = a + (size_t)(q * 2048 + tp) * lda + ct * 8 + c;
        d2 v[32];
#pragma unroll
        for (int m = 0; m < 32; ++m) v[m] = ld<NT>(base + (size_t)(64 * m) * lda);
#pragma unroll
        for (int m = 0; m < 32; ++m) st<NT>(base + (size_t)(64 * m) * lda, v[m] + 1.0); // (a bare copy in place is dropped)
    }
}

template <typename F> void timeit(const char* name, double bytes, int reps, F launch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s %9.1f us  %7.1f GB/s\n", name, 1e3 * ms / reps, bytes / (1e-3 * ms / reps) / 1e9);
}

int main() {
    const size_t big = (size_t)2 << 30, small = (size_t)96 << 20;
    d2 *a, *b; double* sink;
    CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 0, big)); CK(hipMemset(b, 0, big));
    const size_t nb = big / 16, ns = small / 16;
    for (int grid : {2048, 8192, 65536}) {
        printf("-- grid %d x 256 threads, 2 GiB buffers (HBM)\n", grid);
        timeit("read", (double)big, 10, [&] { hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, a, nb, sink); });
        timeit("read nt", (double)big, 10, [&] { hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, a, nb, sink); });
        timeit("write", (double)big, 10, [&] { hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, a, nb, 1.0); });
        timeit("write nt", (double)big, 10, [&] { hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, a, nb, 1.0); });
        timeit("copy (read + write bytes)", 2.0 * big, 10, [&] { hipLaunchKernelGGL((k_copy<false, false>), dim3(grid), dim3(256), 0, 0, a, b, nb); });
        timeit("copy nt loads", 2.0 * big, 10, [&] { hipLaunchKernelGGL((k_copy<true, false>), dim3(grid), dim3(256), 0, 0, a, b, nb); });
        timeit("copy nt stores", 2.0 * big, 10, [&] { hipLaunchKernelGGL((k_copy<false, true>), dim3(grid), dim3(256), 0, 0, a, b, nb); });
        timeit("copy nt both", 2.0 * big, 10, [&] { hipLaunchKernelGGL((k_copy<true, true>), dim3(grid), dim3(256), 0, 0, a, b, nb); });
        timeit("copy4 nt both (4 loads in flight)", 2.0 * big, 10, [&] { hipLaunchKernelGGL((k_copy4<true, true>), dim3(grid), dim3(256), 0, 0, a, b, nb); });
    }
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        printf("-- persistent grid %d x 256 threads, block-contiguous chunks, 2 GiB buffers (HBM)\n", grid);
        timeit("copy chunk 4 loads in flight", 2.0 * big, 10, [&] { hipLaunchKernelGGL((k_copy_chunk<4, false, false>), dim3(grid), dim3(256), 0, 0, a, b, nb); });
        timeit("copy chunk 8 loads in flight", 2.0 * big, 10, [&] { hipLaunchKernelGGL((k_copy_chunk<8, false, false>), dim3(grid), dim3(256), 0, 0, a, b, nb); });
        timeit("copy chunk 8 loads in flight, nt both", 2.0 * big, 10, [&] { hipLaunchKernelGGL((k_copy_chunk<8, true, true>), dim3(grid), dim3(256), 0, 0, a, b, nb); });
        timeit("copy chunk 16 loads in flight, nt both", 2.0 * big, 10, [&] { hipLaunchKernelGGL((k_copy_chunk<16, true, true>), dim3(grid), dim3(256), 0, 0, a, b, nb); });
    }
    {
        // 16 lattices' worth of packed spectrum: [16 * 4096][2056] complex = 2.16 GB inside the two buffers (a, b are
        // separate allocations: use a alone, 2 GiB = 15.9 lattices -> 15 lattices)
        const int ld = 2056, rows = 15 * 4096, ntq = 256, ntiles = (rows / 2048) * ntq;
        const double bytes = 2.0 * (double)ntiles * 8 * 2048 * 16;
        printf("-- column-pass pattern: 8-column x 2048-row tiles of a [%d][%d] complex array, in place\n", rows, ld);
        for (int grid : {256, 512}) {
            char nm[96];
            snprintf(nm, sizeof nm, "tile rmw, grid %d x 512", grid);
            timeit(nm, bytes, 10, [&] { hipLaunchKernelGGL(k_tile_rmw<false>, dim3(grid), dim3(512), 0, 0, a, ld, ntq, ntiles); });
            snprintf(nm, sizeof nm, "tile rmw nt, grid %d x 512", grid);
            timeit(nm, bytes, 10, [&] { hipLaunchKernelGGL(k_tile_rmw<true>, dim3(grid), dim3(512), 0, 0, a, ld, ntq, ntiles); });
        }
    }
    printf("-- 96 MiB buffer (Infinity Cache resident), grid 8192\n");
    timeit("read 96 MiB, repeated", (double)small, 40, [&] { hipLaunchKernelGGL(k_read<false>, dim3(8192), dim3(256), 0, 0, a, ns, sink); });
    timeit("write 96 MiB, repeated", (double)small, 40, [&] { hipLaunchKernelGGL(k_write<false>, dim3(8192), dim3(256), 0, 0, a, ns, 1.0); });
    timeit("copy 96 -> 96 MiB, repeated (r + w bytes)", 2.0 * small, 40, [&] { hipLaunchKernelGGL((k_copy<false, false>), dim3(8192), dim3(256), 0, 0, a, b, ns); });
    timeit("in-place rmw 96 MiB, repeated (r + w bytes)", 2.0 * small, 40, [&] { hipLaunchKernelGGL((k_copy<false, false>), dim3(8192), dim3(256), 0, 0, a, a, ns); });
    // read resident 96 MiB + write 96 MiB stream to HBM (nt): pass C with a resident workspace
    timeit("read resident 96 MiB -> nt write over 2 GiB", 2.0 * small, 20, [&] {
        static size_t off = 0; off = (off + ns) % (nb - ns);
        hipLaunchKernelGGL((k_copy<false, true>), dim3(8192), dim3(256), 0, 0, a, b + off, ns); });
    return 0;
}
