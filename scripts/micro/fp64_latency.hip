// Micro-benchmark: issue cadence of dependent and independent v_fma_f64 / v_add_f64 / v_mul_f64 chains on one wave
// (and on 2 or 4 waves of one SIMD), in core clocks per instruction (s_memtime: 100 MHz constant clock, converted with
// the measured time of a known v_mov chain is avoided: we report ns per instruction and the clock from hipDeviceProp).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/fp64_latency.hip -o scripts/micro/fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int CHAINS, int KIND> __global__ void k(double* out, int iters, double a, double b) {
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = a + c + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (KIND == 0) x[c] = __builtin_fma(x[c], a, b);
                else if (KIND == 1) x[c] = x[c] + b;
                else x[c] = x[c] * a;
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS, int KIND> void run(const char* name, int threads, double* out, double ghz) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(1), dim3(threads), 0, 0, out, 100, 1.0000001, 1e-9);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(1), dim3(threads), 0, 0, out, iters, 1.0000001, 1e-9);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ninst = (double)iters * 16 * CHAINS; // per wave
    const int waves_per_simd = (threads / 64 + 3) / 4;
    printf("%-10s chains=%d waves/SIMD=%d : %6.2f ns per instruction per wave = %5.1f clk ; SIMD issue interval %5.1f clk\n", name, CHAINS,
           waves_per_simd, 1e6 * ms / ninst, 1e6 * ms / ninst * ghz, 1e6 * ms / ninst * ghz / waves_per_simd);
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    printf("clockRate %.3f GHz\n", ghz);
    double* out; CK(hipMalloc(&out, 1 << 20));
    for (int threads : {64, 256, 512, 1024}) { // 1 wave on one SIMD; 1, 2, 4 waves per SIMD
        run<1, 0>("fma", threads, out, ghz); run<2, 0>("fma", threads, out, ghz); run<4, 0>("fma", threads, out, ghz); run<8, 0>("fma", threads, out, ghz);
        run<1, 1>("add", threads, out, ghz); run<4, 1>("add", threads, out, ghz);
        run<1, 2>("mul", threads, out, ghz); run<4, 2>("mul", threads, out, ghz);
    }
    return 0;
}
