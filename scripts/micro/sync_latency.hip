// Micro-benchmark: what one "launch a short kernel and wait for it" round trip costs on this box, by the way the host
// waits -- the floor under every host-pointer call on a small lattice.
//   hipcc -O2 --offload-arch=gfx950 scripts/micro/sync_latency.hip -o scripts/micro/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
// the kernel itself publishes its completion: a system-scope store into host-mapped memory
__global__ void k_flag(volatile unsigned long long* flag, unsigned long long v) {
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store((unsigned long long*)flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == 1) CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
    if (argc > 1 && atoi(argv[1]) == 2) CK(hipSetDeviceFlags(hipDeviceScheduleBlockingSync));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned long long* flag;
    CK(hipHostMalloc((void**)&flag, 64, hipHostMallocMapped | hipHostMallocPortable));
    unsigned long long* dflag;
    CK(hipHostGetDevicePointer((void**)&dflag, flag, 0));
    *flag = 0;
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int iters = 3000;
    unsigned long long seq = 0;
    auto run = [&](const char* name, auto body) {
        for (int i = 0; i < 50; ++i) body();
        const double t0 = now();
        for (int i = 0; i < iters; ++i) body();
        printf("%-64s %7.2f us per round trip\n", name, (now() - t0) / iters * 1e6);
    };
    run("empty kernel + hipStreamSynchronize", [&] {
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (int*)nullptr);
        CK(hipStreamSynchronize(st));
    });
    run("empty kernel + hipEventRecord + hipEventSynchronize", [&] {
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (int*)nullptr);
        CK(hipEventRecord(ev, st));
        CK(hipEventSynchronize(ev));
    });
    run("empty kernel + hipEventRecord + spin on hipEventQuery", [&] {
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (int*)nullptr);
        CK(hipEventRecord(ev, st));
        while (hipEventQuery(ev) == hipErrorNotReady) {}
    });
    run("empty kernel + spin on hipStreamQuery", [&] {
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (int*)nullptr);
        while (hipStreamQuery(st) == hipErrorNotReady) {}
    });
    run("kernel stores a flag in host-mapped memory + host spins on it", [&] {
        ++seq;
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, dflag, seq);
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
    });
    run("empty kernel + flag kernel + host spins", [&] {
        ++seq;
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (int*)nullptr);
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, st, dflag, seq);
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
    });
    {
        const hipError_t e = hipStreamWriteValue64(st, dflag, 1, 0);
        if (e == hipSuccess) {
            CK(hipStreamSynchronize(st));
            run("empty kernel + hipStreamWriteValue64 + host spins", [&] {
                ++seq;
                hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (int*)nullptr);
                CK(hipStreamWriteValue64(st, dflag, seq, 0));
                while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {}
            });
        } else {
            printf("hipStreamWriteValue64: %s\n", hipGetErrorString(e));
            (void)hipGetLastError();
        }
    }
    CK(hipStreamSynchronize(st));
    return 0;
}
