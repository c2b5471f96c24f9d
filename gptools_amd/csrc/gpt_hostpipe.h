// Pinned staging ring of the host-pointer entry points (gpt_api_host.hip); owned by a context, freed by gpt_destroy.
#pragma once
#include "../../include/gptools_hip.h"
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

// Staging of the host-pointer entry points: the caller's arrays are pageable (Stan's arena, NumPy), which the runtime
// copies at ~20 GB/s. Large arrays go through a ring of pinned bounce buffers instead: a few worker threads copy chunk
// k + 1 between the caller's memory and a pinned buffer while the DMA engine moves chunk k, so a transfer runs at what
// PCIe gives. Owned by a context (one calling thread at a time); small arrays keep the plain path.
struct HostPipe {
    static constexpr size_t CHUNK = (size_t)8 << 20;
    static constexpr int NBUF = 4;
    static constexpr size_t MIN_BYTES = (size_t)2 << 20; // below this the plain copy is as fast
    char* pin[NBUF] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[NBUF] = {nullptr, nullptr, nullptr, nullptr};
    bool busy[NBUF] = {false, false, false, false};
    int next = 0;
    // worker threads: each call of run() hands every worker one slice of a memcpy
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    std::function<void(int)> job;
    unsigned long long generation = 0;
    int pending = 0;
    bool stop = false;

    bool ready() const { return pin[0] != nullptr; }
    int init() {
        if (ready()) return 0;
        stop = false;
        for (int i = 0; i < NBUF; ++i) {
            void* p = nullptr;
            if (hipHostMalloc(&p, CHUNK, hipHostMallocDefault) != hipSuccess) return -1;
            pin[i] = static_cast<char*>(p);
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return -1;
        }
        unsigned hc = std::thread::hardware_concurrency();
        int nt = 6; // enough to outrun PCIe; the GPU boxes give a job 16 cores
        if (const char* e = getenv("GPTOOLS_HIP_COPY_THREADS")) nt = atoi(e);
        if (hc && (int)hc < nt) nt = (int)hc;
        if (nt < 1) nt = 1;
        for (int t = 0; t < nt - 1; ++t) workers.emplace_back([this, t] { loop(t + 1); });
        return 0;
    }
    void loop(int id) {
        unsigned long long seen = 0;
        for (;;) {
            std::function<void(int)> j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_go.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                j = job;
            }
            j(id);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    // dst <- src, n bytes, split over the workers and the calling thread
    void parallel_copy(char* dst, const char* src, size_t n) {
        const int nt = (int)workers.size() + 1;
        if (nt == 1 || n < ((size_t)1 << 20)) {
            memcpy(dst, src, n);
            return;
        }
        auto part = [=](int id) {
            size_t lo = 0, len = 0;
            gpt_host_copy_slice(n, nt, id, &lo, &len);
            if (len) memcpy(dst + lo, src + lo, len);
        };
        {
            std::lock_guard<std::mutex> lk(m);
            job = part;
            pending = (int)workers.size();
            ++generation;
        }
        cv_go.notify_all();
        part(0);
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    // host -> device, asynchronous on `st` once the last chunk has been queued
    hipError_t up(void* dev, const void* host, size_t bytes, hipStream_t st) {
        const char* h = static_cast<const char*>(host);
        char* d = static_cast<char*>(dev);
        for (size_t off = 0; off < bytes; off += CHUNK) {
            const size_t n = bytes - off < CHUNK ? bytes - off : CHUNK;
            const int b = next;
            next = (next + 1) % NBUF;
            hipError_t e;
            if (busy[b] && (e = hipEventSynchronize(ev[b])) != hipSuccess) return e; // its last DMA has left the buffer
            parallel_copy(pin[b], h + off, n);
            if ((e = hipMemcpyAsync(d + off, pin[b], n, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
            if ((e = hipEventRecord(ev[b], st)) != hipSuccess) return e;
            busy[b] = true;
        }
        return hipSuccess;
    }
    // device -> host; returns when the caller's array is complete
    hipError_t down(void* host, const void* dev, size_t bytes, hipStream_t st) {
        char* h = static_cast<char*>(host);
        const char* d = static_cast<const char*>(dev);
        const size_t nchunks = (bytes + CHUNK - 1) / CHUNK;
        int slot[NBUF];
        size_t issued = 0, drained = 0;
        hipError_t e;
        while (drained < nchunks) {
            while (issued < nchunks && issued - drained < (size_t)NBUF) { // keep the DMA engine NBUF chunks ahead
                const int b = next;
                next = (next + 1) % NBUF;
                if (busy[b] && (e = hipEventSynchronize(ev[b])) != hipSuccess) return e;
                const size_t off = issued * CHUNK, n = bytes - off < CHUNK ? bytes - off : CHUNK;
                if ((e = hipMemcpyAsync(pin[b], d + off, n, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
                if ((e = hipEventRecord(ev[b], st)) != hipSuccess) return e;
                busy[b] = true;
                slot[issued % NBUF] = b;
                ++issued;
            }
            const int b = slot[drained % NBUF];
            if ((e = hipEventSynchronize(ev[b])) != hipSuccess) return e;
            const size_t off = drained * CHUNK, n = bytes - off < CHUNK ? bytes - off : CHUNK;
            parallel_copy(h + off, pin[b], n);
            busy[b] = false; // (the buffer is free again: nothing in flight reads or writes it)
            ++drained;
        }
        return hipSuccess;
    }
    void destroy() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv_go.notify_all();
        for (auto& w : workers) w.join();
        workers.clear();
        for (int i = 0; i < NBUF; ++i) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (pin[i]) (void)hipHostFree(pin[i]);
            ev[i] = nullptr;
            pin[i] = nullptr;
        }
    }
};
