// C ABI of libgptools_hip.so (see include/gptools_hip.h): context, plan cache, path dispatch, and the
// host-pointer staging entry points that the Stan user header binds.
#include "../../include/gptools_hip.h"
#include "gpt_common.h"
#include "gpt_host.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// Staging of the host-pointer entry points: the caller's arrays are pageable (Stan's arena, NumPy), which the runtime
// copies at ~20 GB/s. Large arrays go through a ring of pinned bounce buffers instead: a few worker threads copy chunk
// k + 1 between the caller's memory and a pinned buffer while the DMA engine moves chunk k, so a transfer runs at what
// PCIe gives. Owned by a context (one calling thread at a time); small arrays keep the plain path.
// Slice `id` of `nt` of an n-byte copy: ceil(n / nt) rounded up to a page, the last slices shortened or empty, so that
// the slices tile [0, n) exactly for every n and nt (round 2 rounded floor(n / nt) up and lost the last n % nt bytes
// when that was already a multiple of 4096 and nt >= 9). Exported for its unit test (tests/test_host_copy_slices.py).
extern "C" void gpt_host_copy_slice(size_t n, int nt, int id, size_t* lo, size_t* len) {
    if (nt < 1) nt = 1;
    const size_t slice = (((n + (size_t)nt - 1) / (size_t)nt) + 4095) & ~(size_t)4095;
    const size_t a = (size_t)id * slice;
    *lo = a < n ? a : n;
    *len = a < n ? (a + slice <= n ? slice : n - a) : 0;
}

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

struct gpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::map<int, cplx*> tw; // twiddle tables w_n^m, keyed by n
    cplx* ws = nullptr;      // packed-spectrum workspace of the three-pass paths
    size_t ws_bytes = 0;
    double* partials = nullptr;
    size_t partials_bytes = 0;
    double* sink = nullptr; // write-only dump for gradients the caller did not ask for
    size_t sink_bytes = 0;
    double* kscr = nullptr; // kernel-fused log density: per-lattice sums, or the spectrum + d/dcov of the fallback
    size_t kscr_bytes = 0;
    double* oscr = nullptr; // observation models: f, d/df and per-lattice sums when the caller does not keep them
    size_t oscr_bytes = 0;
    char* arena = nullptr; // device staging arena of the host-pointer entry points
    size_t arena_bytes = 0;
    double* pinned = nullptr; // small pinned scratch for scalar read-backs
    std::string err;
    double last_min_eig = NAN;
    bool profiling = false; // record events around the passes of the three-pass paths
    struct ProfRec {
        hipEvent_t t0, t1;
        int pass; // 0 = A, 1 = B, 3 = C (slot 2 is the separate finalize of older builds: always 0 now)
    };
    std::vector<ProfRec> prof;
    int total_cus = 0;
    HostPipe* pipe = nullptr; // pinned staging ring of the host-pointer entry points (created on first large transfer)
    unsigned char* scratch = nullptr; // this device's copy of the built-in scratch array (gpt_gp_rfft2_lpdf_vjp_noalloc)
    gpt_anyn* anyn = nullptr; // plans, tables and scratch of the any-length path (gpt_anyn.hip), created on first use
    // Pinned, device-mapped bounce area of the host-pointer entry points for transfers too small for the ring (Stager):
    // small calls let the kernels read and write it directly, mid-size ones stage their asynchronous copies in it.
    char* bounce = nullptr;     // host view
    char* bounce_dev = nullptr; // device view of the same memory
    size_t bounce_bytes = 0;
};

// Built-in scratch of the allocation-free entry point (gpt_gp_rfft2_lpdf_vjp_noalloc): twiddle tables, per-workgroup
// partials and the packed-spectrum workspace of the lattices of one unit. Every GPU the library runs on holds its own
// copy at its own address until the process ends. In host code the name below is only a handle for
// hipGetSymbolAddress (gpt_create); kernels and copies get the per-device address kept in gpt_ctx::scratch.
#define GPT_NOALLOC_SCRATCH_BYTES ((size_t)40 << 20)
static __device__ __align__(256) unsigned char scratch[GPT_NOALLOC_SCRATCH_BYTES];

// w_n^m = exp(-2 pi i m / n), m = 0 .. n-1, generated on the device (sincospi is exact at the octants)
static __global__ void noalloc_twiddle_kernel(cplx* __restrict__ tw, int n) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    double sn, cs;
    sincospi(2.0 * (double)m / (double)n, &sn, &cs);
    tw[m] = make_double2(cs, -sn);
}

namespace {

int fail(gpt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}
int hip_fail(gpt_ctx* c, hipError_t e, const char* what) {
    return fail(c, e == hipErrorOutOfMemory ? GPT_ERR_ALLOC : GPT_ERR_HIP,
                std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(c, expr)                                                                                       \
    do {                                                                                                       \
        hipError_t e__ = (expr);                                                                               \
        if (e__ != hipSuccess) return hip_fail((c), e__, #expr);                                               \
    } while (0)

template <typename T> int ensure(gpt_ctx* c, T*& ptr, size_t& have, size_t need) {
    if (need <= have) return GPT_OK;
    if (ptr) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipFree(ptr));
        ptr = nullptr;
        have = 0;
    }
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, need));
    ptr = reinterpret_cast<T*>(p);
    have = need;
    return GPT_OK;
}

int get_twiddles(gpt_ctx* c, int n, const cplx** out) {
    auto it = c->tw.find(n);
    if (it != c->tw.end()) {
        *out = it->second;
        return GPT_OK;
    }
    std::vector<cplx> h((size_t)n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int m = 0; m < n; ++m) {
        // exact octant values where they exist, long-double elsewhere
        const long double ang = two_pi * (long double)m / (long double)n;
        long double cr = cosl(ang), si = -sinl(ang);
        if ((4LL * m) % n == 0) {
            const int q = (int)((4LL * m) / n) & 3;
            cr = (q == 0) ? 1.0L : (q == 2) ? -1.0L : 0.0L;
            si = (q == 1) ? -1.0L : (q == 3) ? 1.0L : 0.0L;
        }
        h[(size_t)m] = make_double2((double)cr, (double)si);
    }
    void* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, sizeof(cplx) * (size_t)n));
    HIP_TRY(c, hipMemcpy(d, h.data(), sizeof(cplx) * (size_t)n, hipMemcpyHostToDevice));
    c->tw[n] = reinterpret_cast<cplx*>(d);
    *out = reinterpret_cast<cplx*>(d);
    return GPT_OK;
}

// Stockham radices of a line of n points (host arithmetic only)
// wide: radix-16 / radix-8 stages first (the generic three-pass path: fewer passes through LDS); the one-workgroup
// kernels keep radix 4 and 2 (their register budget)
int plan_radices(gpt_ctx* c, int n, LinePlan& p, bool wide = false) {
    p.n = n;
    p.ns = 0;
    p.tw = nullptr;
    int m = n;
    auto push = [&](int r) {
        if (p.ns < GPT_MAX_STAGES) p.r[p.ns] = r;
        p.ns++;
    };
    if (wide) {
        while (m % 16 == 0) { push(16); m /= 16; }
        if (m % 8 == 0) { push(8); m /= 8; }
    }
    while (m % 4 == 0) { push(4); m /= 4; }
    while (m % 2 == 0) { push(2); m /= 2; }
    for (int f = 3; (int64_t)f * f <= m; f += 2)
        while (m % f == 0) { push(f); m /= f; }
    if (m > 1) push(m);
    if (p.ns > GPT_MAX_STAGES) return fail(c, GPT_ERR_UNSUPPORTED, "too many FFT stages");
    return GPT_OK;
}

int make_plan(gpt_ctx* c, int n, LinePlan& p, bool wide = false) {
    const int rc = plan_radices(c, n, p, wide);
    return rc != GPT_OK ? rc : get_twiddles(c, n, &p.tw);
}

int path_for_shape(int64_t H, int64_t W) {
    if (H < 1 || W < 1 || H * W > 2147483647LL) return -1;
    if (gpt_small_fits((int)H, (int)W, nullptr, nullptr)) return 0;
    if (gpt_pow2_supported((int)H, (int)W)) return 1;
    if (gpt_generic_supported((int)H, (int)W) && !gpt_anyn_preferred((int)H, (int)W)) return 2;
    if (gpt_anyn_supported((int)H, (int)W)) return 3;
    return -1;
}

int check_shape(gpt_ctx* c, int64_t H, int64_t W, int64_t batch, int dim1) {
    if (!c) return GPT_ERR_SHAPE;
    if (H < 1 || W < 1 || batch < 0) return fail(c, GPT_ERR_SHAPE, "height, width >= 1 and batch >= 0 required");
    if (dim1 && H != 1) return fail(c, GPT_ERR_SHAPE, "1-D functions take height == 1");
    // n = height * width is a 32-bit Stan int in the reference (fft2.stan:15)
    if (H * W > 2147483647LL) return fail(c, GPT_ERR_SHAPE, "height * width exceeds the reference's 32-bit int");
    return GPT_OK;
}

// The three passes over `a.batch` lattices in units of `chunk` lattices that share the workspace `ws` (chunk times
// the path's per-lattice workspace) and `partials`, one unit after the other on the context's stream.
int run_units(gpt_ctx* c, int op, const LatArgs& a, int path, int64_t chunk, cplx* ws, double* partials, bool may_profile) {
    const int64_t hw = (int64_t)a.H * a.W, hwf = (int64_t)a.H * (a.W / 2 + 1);
    for (int64_t b0 = 0; b0 < a.batch; b0 += chunk) {
        LatArgs s = a;
        s.batch = (a.batch - b0 < chunk) ? a.batch - b0 : chunk;
        if (s.x) s.x += b0 * a.x_bs;
        if (s.loc) s.loc += b0 * a.loc_bs;
        if (s.cov) s.cov += b0 * a.cov_bs;
        if (s.aux) s.aux += b0 * a.aux_bs;
        if (s.cin) s.cin += b0 * a.cin_bs;
        if (s.out) s.out += b0 * hw;
        if (s.out_cov) s.out_cov += b0 * hwf;
        if (s.out_c) s.out_c += b0 * hwf;
        if (s.out_lpdf) s.out_lpdf += b0;
        if (s.min_eig) s.min_eig += b0;
        if (s.kpar) s.kpar += b0 * 5;
        if (s.ksums) s.ksums += b0 * 4;
        if (s.kbar) s.kbar += b0 * 5;
        if (path == 3) {
            HIP_TRY(c, gpt_launch_anyn(c->anyn, op, s, c->stream));
            continue;
        }
        if (path != 1) {
            HIP_TRY(c, gpt_launch_generic(op, s, ws, partials, c->stream));
            continue;
        }
        for (int pass = 0; pass < 3; ++pass) {
            // each pass bracketed by timing events in profiling mode
            const bool present = pass == 1 || (pass == 0 ? op_has_fwd(op) : op_has_inv(op));
            gpt_ctx::ProfRec pr = {nullptr, nullptr, pass == 2 ? 3 : pass};
            if (may_profile && c->profiling && present) {
                HIP_TRY(c, hipEventCreate(&pr.t0));
                HIP_TRY(c, hipEventCreate(&pr.t1));
                HIP_TRY(c, hipEventRecord(pr.t0, c->stream));
            }
            HIP_TRY(c, gpt_pow2_pass(pass, op, s, ws, partials, c->total_cus, c->stream));
            if (pr.t0) {
                HIP_TRY(c, hipEventRecord(pr.t1, c->stream));
                c->prof.push_back(pr);
            }
        }
    }
    return GPT_OK;
}

int run_lattice(gpt_ctx* c, int op, LatArgs a) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (a.batch == 0) return GPT_OK;
    int rc;
    const int path = path_for_shape(a.H, a.W);
    if (path == 3) { // any-length path: its own plans (four-step / Bluestein), tables and scratch
        if (!c->anyn) c->anyn = gpt_anyn_create();
        return run_units(c, op, a, path, gpt_anyn_chunk(a.H, a.W), nullptr, nullptr, false);
    }
    if ((rc = make_plan(c, a.W, a.pw, path == 2)) != GPT_OK) return rc;
    if ((rc = make_plan(c, a.H, a.ph, path == 2)) != GPT_OK) return rc;
    if (path == 0) {
        HIP_TRY(c, gpt_launch_small(op, a, c->stream));
        return GPT_OK;
    }
    if (path < 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "no kernel for a %d x %d lattice in this build", a.H, a.W);
        return fail(c, GPT_ERR_UNSUPPORTED, buf);
    }
    // Three-pass paths: lattices are processed in units (chunks of lattices that share one packed-spectrum
    // workspace of <= ~128 MiB, so that it can live in the 256 MiB Infinity Cache between the passes), one
    // unit after the other on the context's stream. (Overlapping pass B of one lattice with the row passes of
    // its neighbours on a second stream, with and without CU masks, was measured and is slower: pass B has no
    // idle CU resources to share and slows down in proportion to the CUs it gives up -- DESIGN.md section 6.)
    if (path == 1) {
        const int R = gpt_pow2_sub_radix(op, a.H, a.W);
        if (R && (rc = get_twiddles(c, a.H / R, &a.tw_sub)) != GPT_OK) return rc;
    }
    const size_t per = path == 1 ? gpt_pow2_workspace_bytes(a.H, a.W) : gpt_generic_workspace_bytes(a.H, a.W);
    // Ops with a forward row pass (power-of-two path) take units of up to GPT_UNIT_MIB (default 2560) of workspace
    // instead: that pass numbers its workgroups lattice-fastest, so the rows of a loc shared by the batch are read
    // from HBM once per unit, and launches of many lattices have no tails. At 4096^2 with 16 lattices per unit the
    // row pass takes 52 instead of 76 us per lattice, the column pass (its workspace now comes from HBM) 125 instead
    // of 117, the inverse row pass 54 instead of 50: +3 % overall; units of 2..9 lattices lose.
    static const size_t unit_budget = [] {
        const char* e = getenv("GPT_UNIT_MIB");
        const long v = e ? atol(e) : 0;
        return (size_t)(v > 0 ? v : 2560) << 20;
    }();
    const size_t small_budget = (size_t)128 << 20;
    int64_t chunk = (int64_t)(small_budget / per);
    if (chunk < 1) chunk = 1;
    if (path == 1 && op_has_fwd(op) && chunk < 4) {
        // (lattices so large that a cache-sized unit holds only a few: launch tails matter.) A unit that is neither
        // cache-sized nor many lattices is the worst of both: below 1.5 GiB of workspace the small units stay.
        int64_t big = (int64_t)(unit_budget / per);
        if (big < 1) big = 1;
        const int64_t nunits = (a.batch + big - 1) / big;
        big = (a.batch + nunits - 1) / nunits; // units of equal size
        static const size_t unit_min = [] { // GPT_UNIT_MIN_MIB: smallest unit worth giving up cache residency for
            const char* e = getenv("GPT_UNIT_MIN_MIB");
            const long v = e ? atol(e) : 0;
            return (size_t)(v > 0 ? v : 1536) << 20;
        }();
        if (per * (size_t)big >= unit_min) chunk = big;
    }
    if (chunk > a.batch) chunk = a.batch;
    if (chunk > 32768) chunk = 32768;
    if (c->total_cus == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
        c->total_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int nblk = path == 1 ? gpt_pow2_col_blocks(a.H, a.W, c->total_cus) : gpt_generic_col_blocks(a.H, a.W);
    if ((rc = ensure(c, c->ws, c->ws_bytes, per * (size_t)chunk)) != GPT_OK) return rc;
    if ((rc = ensure(c, c->partials, c->partials_bytes, sizeof(double) * GPT_PART * (size_t)nblk * (size_t)chunk)) !=
        GPT_OK)
        return rc;
    return run_units(c, op, a, path, chunk, c->ws, c->partials, true);
}

LatArgs base_args(int64_t H, int64_t W, int64_t batch, int dim1) {
    LatArgs a;
    memset(&a, 0, sizeof a);
    a.H = (int)H;
    a.W = (int)W;
    a.batch = batch;
    a.dim1 = dim1 ? 1 : 0;
    return a;
}

int run_elem(gpt_ctx* c, int op, ElemArgs e, bool reduces, int K, double add, double* out_sum, double* out_min) {
    HIP_TRY(c, hipSetDevice(c->device));
    if (e.batch == 0) return GPT_OK;
    const int nblk = gpt_elem_blocks(e.H, e.W);
    const int64_t chunk_max = 32768;
    int rc;
    if (reduces) {
        const int64_t chunk = e.batch < chunk_max ? e.batch : chunk_max;
        if ((rc = ensure(c, c->partials, c->partials_bytes, sizeof(double) * GPT_PART * (size_t)nblk * (size_t)chunk)) !=
            GPT_OK)
            return rc;
    }
    const int64_t hw = (int64_t)e.H * e.W, hwf = (int64_t)e.H * (e.W / 2 + 1);
    const int npar = e.dim1 ? 3 : 5;
    for (int64_t b0 = 0; b0 < e.batch; b0 += chunk_max) {
        ElemArgs s = e;
        s.batch = (e.batch - b0 < chunk_max) ? e.batch - b0 : chunk_max;
        if (s.in) s.in += b0 * e.in_bs;
        if (s.cin) s.cin += b0 * e.cin_bs;
        if (s.par) s.par += b0 * npar;
        if (s.out) {
            if (op == EL_UNPACK) s.out += b0 * hw;
            else s.out += b0 * hwf;
        }
        if (s.cout) s.cout += b0 * (op == EL_EXPAND ? (int64_t)e.W : hwf);
        HIP_TRY(c, gpt_launch_elem(op, s, c->partials, c->stream));
        if (reduces)
            HIP_TRY(c, gpt_launch_finalize(c->partials, nblk, s.batch, K, add, out_sum ? out_sum + b0 * K : nullptr,
                                           out_min ? out_min + b0 : nullptr, c->stream));
    }
    return GPT_OK;
}

ElemArgs elem_args(int64_t H, int64_t W, int64_t batch, int dim1) {
    ElemArgs e;
    memset(&e, 0, sizeof e);
    e.H = (int)H;
    e.W = (int)W;
    e.batch = batch;
    e.dim1 = dim1 ? 1 : 0;
    return e;
}

// Fused route of gpt_gp_inv_rfft2_obs_lpdf_vjp on the power-of-two path: column pass of the inverse transform, one
// row kernel (inverse rows -> observation model -> forward rows, in place on the workspace), column pass of the
// reverse mode. sums: [batch][4] on the device.
int run_obs_fused(gpt_ctx* c, int family, const double* z, int64_t z_bs, const double* loc, int64_t loc_bs,
                  const double* cov, int64_t cov_bs, const double* obs, int64_t obs_bs, int oh, int ow,
                  const double* theta, int H, int W, int64_t batch, double* f_out, double* fbar_out, double* grad_z,
                  double* grad_cov, double* min_eig, double* sums, double* psum) {
    int rc;
    LatArgs a = base_args(H, W, batch, 0);
    if ((rc = make_plan(c, W, a.pw)) != GPT_OK) return rc;
    if ((rc = make_plan(c, H, a.ph)) != GPT_OK) return rc;
    const int R = gpt_pow2_sub_radix(OP_GP_INV_RFFT2, H, W);
    if (R && (rc = get_twiddles(c, H / R, &a.tw_sub)) != GPT_OK) return rc;
    if (c->total_cus == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
        c->total_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const size_t per = gpt_pow2_workspace_bytes(H, W);
    int64_t chunk = (int64_t)((size_t)(128u << 20) / per);
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if (chunk > 32768) chunk = 32768;
    const int nblk = gpt_pow2_col_blocks(H, W, c->total_cus), nobs = gpt_pow2_obs_blocks(H, W);
    if ((rc = ensure(c, c->ws, c->ws_bytes, per * (size_t)chunk)) != GPT_OK) return rc;
    if ((rc = ensure(c, c->partials, c->partials_bytes, sizeof(double) * GPT_PART * (size_t)(nblk + nobs) * (size_t)chunk)) !=
        GPT_OK)
        return rc;
    double* obs_partials = c->partials + GPT_PART * (size_t)nblk * (size_t)chunk;
    const int64_t hw = (int64_t)H * W, hwf = (int64_t)H * (W / 2 + 1);
    if (!grad_cov) {
        if ((rc = ensure(c, c->sink, c->sink_bytes, sizeof(double) * (size_t)chunk * hwf))) return rc;
    }
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = batch - b0 < chunk ? batch - b0 : chunk;
        LatArgs si = a; // gp_inv_rfft2
        si.batch = nb;
        si.x = z + b0 * z_bs; si.x_bs = z_bs;
        si.loc = loc ? loc + b0 * loc_bs : nullptr; si.loc_bs = loc_bs;
        si.cov = cov + b0 * cov_bs; si.cov_bs = cov_bs;
        si.min_eig = min_eig ? min_eig + b0 : nullptr;
        HIP_TRY(c, gpt_pow2_pass(1, OP_GP_INV_RFFT2, si, c->ws, c->partials, c->total_cus, c->stream));
        ObsRowArgs o;
        memset(&o, 0, sizeof o);
        o.family = family; o.oh = oh; o.ow = ow;
        o.obs = obs + b0 * obs_bs; o.obs_bs = obs_bs;
        o.theta = theta ? theta + b0 : nullptr;
        o.f_out = f_out ? f_out + b0 * hw : nullptr;
        o.fbar_out = fbar_out ? fbar_out + b0 * hw : nullptr;
        o.partials = obs_partials;
        HIP_TRY(c, gpt_pow2_obs_rows(si, c->ws, c->partials, c->total_cus, o, c->stream));
        HIP_TRY(c, gpt_launch_finalize(obs_partials, nobs, nb, 4, 0.0, sums + 4 * b0, nullptr, c->stream));
        LatArgs sv = a; // its reverse mode: the workspace holds the row transforms of d lp / d f
        sv.batch = nb;
        sv.x = si.x; sv.x_bs = z_bs;
        sv.cov = si.cov; sv.cov_bs = cov_bs;
        sv.out = grad_z + b0 * hw;
        sv.out_cov = grad_cov ? grad_cov + b0 * hwf : c->sink;
        sv.zprior = psum ? 1 : 0;
        sv.out_lpdf = psum ? psum + b0 : nullptr;
        HIP_TRY(c, gpt_pow2_pass(1, OP_INV_VJP, sv, c->ws, c->partials, c->total_cus, c->stream));
    }
    return GPT_OK;
}

} // namespace

// ===================================================================================================
extern "C" {

const char* gpt_version(void) { return "gptools_hip 0.1 (gfx950)"; }

int gpt_create(int device, void* stream, gpt_ctx** out) {
    if (!out) return GPT_ERR_SHAPE;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return GPT_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return GPT_ERR_HIP;
    gpt_ctx* c = new gpt_ctx();
    c->device = device;
    c->stream = reinterpret_cast<hipStream_t>(stream); // NULL = HIP's default stream
    void* p = nullptr;
    if (hipHostMalloc(&p, 4096, hipHostMallocDefault) != hipSuccess) {
        delete c;
        return GPT_ERR_ALLOC;
    }
    c->pinned = reinterpret_cast<double*>(p);
    // this device's copy of the built-in scratch array (`device` is current) and its CU count: nothing after this
    // point of an allocation-free call has to ask the runtime for either
    void* sp = nullptr;
    if (hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)) != hipSuccess || !sp) {
        (void)hipHostFree(p);
        delete c;
        return GPT_ERR_HIP;
    }
    c->scratch = reinterpret_cast<unsigned char*>(sp);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        c->total_cus = prop.multiProcessorCount;
    *out = c;
    return GPT_OK;
}

int gpt_destroy(gpt_ctx* c) {
    if (!c) return GPT_OK;
    // (may run on whichever thread collects the context, e.g. a finalizer: leave that thread's current device alone)
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& r : c->prof) {
        (void)hipEventDestroy(r.t0);
        (void)hipEventDestroy(r.t1);
    }
    for (auto& kv : c->tw) (void)hipFree(kv.second);
    if (c->ws) (void)hipFree(c->ws);
    if (c->partials) (void)hipFree(c->partials);
    if (c->arena) (void)hipFree(c->arena);
    if (c->kscr) (void)hipFree(c->kscr);
    if (c->oscr) (void)hipFree(c->oscr);
    if (c->sink) (void)hipFree(c->sink);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pipe) {
        c->pipe->destroy();
        delete c->pipe;
    }
    if (c->bounce) (void)hipHostFree(c->bounce);
    gpt_anyn_destroy(c->anyn);
    delete c;
    if (prev >= 0) (void)hipSetDevice(prev);
    return GPT_OK;
}

int gpt_set_stream(gpt_ctx* c, void* stream) {
    if (!c) return GPT_ERR_SHAPE;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = reinterpret_cast<hipStream_t>(stream);
    return GPT_OK;
}

int gpt_synchronize(gpt_ctx* c) {
    if (!c) return GPT_ERR_SHAPE;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GPT_OK;
}

int gpt_profile_begin(gpt_ctx* c) {
    if (!c) return GPT_ERR_SHAPE;
    for (auto& r : c->prof) {
        (void)hipEventDestroy(r.t0);
        (void)hipEventDestroy(r.t1);
    }
    c->prof.clear();
    c->profiling = true;
    return GPT_OK;
}

int gpt_profile_end(gpt_ctx* c, double* pass_ms, int64_t* launches) {
    if (!c) return GPT_ERR_SHAPE;
    c->profiling = false;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double acc[4] = {0, 0, 0, 0};
    int64_t nb = 0;
    for (auto& r : c->prof) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, r.t0, r.t1));
        acc[r.pass] += ms;
        if (r.pass == 1) ++nb;
    }
    for (auto& r : c->prof) {
        (void)hipEventDestroy(r.t0);
        (void)hipEventDestroy(r.t1);
    }
    c->prof.clear();
    if (pass_ms)
        for (int k = 0; k < 4; ++k) pass_ms[k] = acc[k];
    if (launches) *launches = nb;
    return GPT_OK;
}

const char* gpt_last_error(const gpt_ctx* c) { return c ? c->err.c_str() : "null context"; }
double gpt_last_min_eig(const gpt_ctx* c) { return c ? c->last_min_eig : NAN; }
int gpt_path_for_shape(int64_t H, int64_t W) { return path_for_shape(H, W); }

// ---- slab decomposition: one lattice over several GPUs ----------------------------------------------
// (no counterpart in the reference, which is single-process: SURVEY 8(e)). Each rank owns `rows` consecutive
// rows of y / loc / d/dy and `ncols` consecutive packed columns of the half spectrum; the exchange between
// the two layouts is the caller's (gptools_amd/slab.py: all-to-all over RCCL).
static int slab_check(gpt_ctx* c, int64_t H, int64_t W) {
    if (!c) return GPT_ERR_SHAPE;
    if (H < 1 || W < 1 || H * W > 2147483647LL || !gpt_slab_supported((int)H, (int)W))
        return fail(c, GPT_ERR_UNSUPPORTED, "slab passes need a power-of-two height and width in [256, 32768]");
    return GPT_OK;
}
// twiddles of the W / 2-point lines of long rows (a no-op for rows whose two-for-one line one workgroup holds)
static int slab_long_rows(gpt_ctx* c, LatArgs& a) {
    if (gpt_long_row_split_radix(a.W) <= 0) return GPT_OK;
    return get_twiddles(c, a.W / 2, &a.tw_sub);
}
static int slab_cus(gpt_ctx* c) {
    if (c->total_cus == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return 256;
        c->total_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return c->total_cus;
}

// block_cols == 0: spec[rows][ld]. block_cols > 0 (a power of two dividing width / 2): the layout of the exchange,
// spec[width / 2 / block_cols][rows][block_cols] -- block p holds this rank's rows of the columns rank p owns.
static int slab_blocking(gpt_ctx* c, LatArgs& a, int64_t rows, int64_t W, int64_t block_cols, int64_t* ld) {
    if (block_cols == 0) return GPT_OK;
    if (block_cols < 1 || (block_cols & (block_cols - 1)) || (W / 2) % block_cols || rows * block_cols > 2147483647LL)
        return fail(c, GPT_ERR_SHAPE, "block_cols must be a power of two that divides width / 2");
    int sh = 0;
    while (((int64_t)1 << sh) < block_cols) ++sh;
    if (sh == 0) return fail(c, GPT_ERR_SHAPE, "block_cols must be at least 2");
    a.spec_cshift = sh;
    a.spec_blk = (int)(rows * block_cols);
    *ld = block_cols;
    return GPT_OK;
}

static int slab_rows_fwd(gpt_ctx* c, const double* y, const double* loc, int64_t rows, int64_t H, int64_t W, double* spec,
                         int64_t ld, int64_t block_cols) {
    int rc = slab_check(c, H, W);
    if (rc) return rc;
    if (rows < 2 || rows % 2 || rows > H || (!block_cols && ld < W / 2) || !y || !spec)
        return fail(c, GPT_ERR_SHAPE, "rows must be even, 2 <= rows <= height, and ld >= width / 2");
    HIP_TRY(c, hipSetDevice(c->device));
    LatArgs a = base_args(rows, W, 1, 0);
    if ((rc = make_plan(c, a.W, a.pw)) != GPT_OK) return rc;
    a.x = y;
    a.loc = loc;
    if ((rc = slab_blocking(c, a, rows, W, block_cols, &ld)) != GPT_OK) return rc;
    if ((rc = slab_long_rows(c, a)) != GPT_OK) return rc;
    HIP_TRY(c, gpt_pow2_slab_pass(0, OP_LPDF_VJP, a, reinterpret_cast<cplx*>(spec), (int)ld, 1.0, nullptr,
                                  slab_cus(c), nullptr, c->stream));
    return GPT_OK;
}
int gpt_slab_rows_fwd(gpt_ctx* c, const double* y, const double* loc, int64_t rows, int64_t H, int64_t W,
                      double* spec, int64_t ld) {
    return slab_rows_fwd(c, y, loc, rows, H, W, spec, ld, 0);
}
int gpt_slab_rows_fwd_blocked(gpt_ctx* c, const double* y, const double* loc, int64_t rows, int64_t H, int64_t W,
                              double* spec, int64_t block_cols) {
    if (block_cols < 1) return fail(c, GPT_ERR_SHAPE, "block_cols must be a power of two that divides width / 2");
    return slab_rows_fwd(c, y, loc, rows, H, W, spec, 0, block_cols);
}

// cov / out_cov: arrays of row stride cov_ld whose column 0 is lattice column cov_col0 (the lattice's own H x (W/2+1)
// arrays: cov_ld = W/2+1, cov_col0 = 0). edge (may be null): 4 H doubles, see gpt_pow2_slab_pass.
static int slab_cols(gpt_ctx* c, double* spec, int64_t ld, const double* cov, double* out_cov, int64_t cov_ld,
                     int64_t cov_col0, double* edge, int64_t H, int64_t W, int64_t col0, int64_t ncols,
                     double* lpdf_partial, double* min_partial) {
    int rc = slab_check(c, H, W);
    if (rc) return rc;
    const int tile = gpt_slab_col_tile((int)H);
    if (col0 < 0 || ncols < 1 || col0 + ncols > W / 2 || ncols % tile || col0 % tile || ld < ncols || !spec || !cov ||
        !lpdf_partial) {
        char buf[200];
        snprintf(buf, sizeof buf, "column slab [%lld, %lld) must lie in [0, width/2) and be aligned to %d columns",
                 (long long)col0, (long long)(col0 + ncols), tile);
        return fail(c, GPT_ERR_SHAPE, buf);
    }
    HIP_TRY(c, hipSetDevice(c->device));
    LatArgs a = base_args(H, W, 1, 0);
    if ((rc = make_plan(c, a.H, a.ph)) != GPT_OK) return rc;
    a.col0 = (int)col0;
    a.ncols = (int)ncols;
    a.out_lpdf = lpdf_partial;
    a.min_eig = min_partial;
    const int cus = slab_cus(c);
    const int nblk = gpt_slab_col_blocks(a.H, a.ncols, cus);
    if ((rc = ensure(c, c->partials, c->partials_bytes, sizeof(double) * GPT_PART * (size_t)nblk)) != GPT_OK) return rc;
    if (const int R = gpt_long_col_split_radix(a.H); R > 0)
        if ((rc = get_twiddles(c, a.H / R, &a.tw_sub)) != GPT_OK) return rc;
    if (!out_cov) { // d/dcov not wanted: a write-only dump of the slab's size
        const size_t need = sizeof(double) * (size_t)H * (size_t)cov_ld;
        if ((rc = ensure(c, c->sink, c->sink_bytes, need)) != GPT_OK) return rc;
        out_cov = c->sink;
    }
    // the kernels index both arrays with lattice column numbers: shift the bases (never dereferenced below cov_col0)
    a.cov = cov - cov_col0;
    a.out_cov = out_cov - cov_col0;
    if (cov_ld != W / 2 + 1) a.cov_ld = (int)cov_ld;
    HIP_TRY(c, gpt_pow2_slab_pass(1, OP_LPDF_VJP, a, reinterpret_cast<cplx*>(spec), (int)ld, 1.0, c->partials, cus,
                                  edge, c->stream));
    return GPT_OK;
}
int gpt_slab_cols_lpdf_vjp(gpt_ctx* c, double* spec, int64_t ld, const double* cov, double* out_cov, int64_t H,
                           int64_t W, int64_t col0, int64_t ncols, double* lpdf_partial, double* min_partial) {
    return slab_cols(c, spec, ld, cov, out_cov, W / 2 + 1, 0, nullptr, H, W, col0, ncols, lpdf_partial, min_partial);
}
int gpt_slab_cols_lpdf_vjp_owned(gpt_ctx* c, double* spec, int64_t ld, const double* cov_cols, double* out_cov_cols,
                                 int64_t cov_ld, double* edge, int64_t H, int64_t W, int64_t col0, int64_t ncols,
                                 double* lpdf_partial, double* min_partial) {
    if (!c) return GPT_ERR_SHAPE;
    if (cov_ld < ncols) return fail(c, GPT_ERR_SHAPE, "cov_ld must be at least ncols");
    if (col0 == 0 && !edge)
        return fail(c, GPT_ERR_SHAPE, "the rank that owns packed column 0 passes the staging of columns 0 and width/2");
    return slab_cols(c, spec, ld, cov_cols, out_cov_cols, cov_ld, col0, col0 == 0 ? edge : nullptr, H, W, col0, ncols,
                     lpdf_partial, min_partial);
}

static int slab_rows_inv(gpt_ctx* c, const double* spec, int64_t ld, int64_t block_cols, int64_t rows, int64_t H,
                         int64_t W, double alpha, double* out) {
    int rc = slab_check(c, H, W);
    if (rc) return rc;
    if (rows < 2 || rows % 2 || rows > H || (!block_cols && ld < W / 2) || !out || !spec)
        return fail(c, GPT_ERR_SHAPE, "rows must be even, 2 <= rows <= height, and ld >= width / 2");
    HIP_TRY(c, hipSetDevice(c->device));
    LatArgs a = base_args(rows, W, 1, 0);
    if ((rc = make_plan(c, a.W, a.pw)) != GPT_OK) return rc;
    a.out = out;
    if ((rc = slab_blocking(c, a, rows, W, block_cols, &ld)) != GPT_OK) return rc;
    if ((rc = slab_long_rows(c, a)) != GPT_OK) return rc;
    HIP_TRY(c, gpt_pow2_slab_pass(2, OP_LPDF_VJP, a, const_cast<cplx*>(reinterpret_cast<const cplx*>(spec)), (int)ld,
                                  alpha, nullptr, slab_cus(c), nullptr, c->stream));
    return GPT_OK;
}
int gpt_slab_rows_inv(gpt_ctx* c, const double* spec, int64_t ld, int64_t rows, int64_t H, int64_t W, double alpha,
                      double* out) {
    return slab_rows_inv(c, spec, ld, 0, rows, H, W, alpha, out);
}
int gpt_slab_rows_inv_blocked(gpt_ctx* c, const double* spec, int64_t block_cols, int64_t rows, int64_t H, int64_t W,
                              double alpha, double* out) {
    if (block_cols < 1) return fail(c, GPT_ERR_SHAPE, "block_cols must be a power of two that divides width / 2");
    return slab_rows_inv(c, spec, 0, block_cols, rows, H, W, alpha, out);
}

// ---- real-FFT layer --------------------------------------------------------------------------------
int gpt_rfft2(gpt_ctx* c, const double* y, int64_t H, int64_t W, int64_t batch, double* out_complex) {
    int rc = check_shape(c, H, W, batch, 0);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, 0);
    a.x = y;
    a.x_bs = H * W;
    a.out_c = reinterpret_cast<cplx*>(out_complex);
    return run_lattice(c, OP_RFFT2, a);
}

int gpt_inv_rfft2(gpt_ctx* c, const double* z_complex, int64_t H, int64_t W, int64_t batch, double* out) {
    int rc = check_shape(c, H, W, batch, 0);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, 0);
    a.cin = reinterpret_cast<const cplx*>(z_complex);
    a.cin_bs = H * (W / 2 + 1);
    a.out = out;
    return run_lattice(c, OP_INV_RFFT2, a);
}

int gpt_expand_rfft(gpt_ctx* c, const double* y_complex, int64_t n, int64_t batch, double* out_complex) {
    int rc = check_shape(c, 1, n, batch, 1);
    if (rc) return rc;
    ElemArgs e = elem_args(1, n, batch, 1);
    e.cin = reinterpret_cast<const cplx*>(y_complex);
    e.cin_bs = n / 2 + 1;
    e.cout = reinterpret_cast<cplx*>(out_complex);
    return run_elem(c, EL_EXPAND, e, false, 0, 0.0, nullptr, nullptr);
}

// ---- GP layer --------------------------------------------------------------------------------------
int gpt_gp_evaluate_rfft2_scale(gpt_ctx* c, const double* cov, int64_t cov_bs, int64_t H, int64_t W,
                                int64_t batch, int dim1, double* out, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.in = cov;
    e.in_bs = cov_bs;
    e.out = out;
    return run_elem(c, EL_SCALE, e, true, 0, 0.0, nullptr, min_eig);
}

int gpt_gp_unpack_rfft2(gpt_ctx* c, const double* z_complex, int64_t H, int64_t W, int64_t batch, int dim1,
                        double* out) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.cin = reinterpret_cast<const cplx*>(z_complex);
    e.cin_bs = H * (W / 2 + 1);
    e.out = out;
    return run_elem(c, EL_UNPACK, e, false, 0, 0.0, nullptr, nullptr);
}

int gpt_gp_pack_rfft2(gpt_ctx* c, const double* z, int64_t H, int64_t W, int64_t batch, int dim1,
                      double* out_complex) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.in = z;
    e.in_bs = H * W;
    e.cout = reinterpret_cast<cplx*>(out_complex);
    return run_elem(c, EL_PACK, e, false, 0, 0.0, nullptr, nullptr);
}

int gpt_gp_rfft2(gpt_ctx* c, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs, const double* cov,
                 int64_t cov_bs, int64_t H, int64_t W, int64_t batch, int dim1, double* z_out, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = y; a.x_bs = y_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out = z_out;
    a.min_eig = min_eig;
    return run_lattice(c, OP_GP_RFFT2, a);
}

int gpt_gp_inv_rfft2(gpt_ctx* c, const double* z, int64_t z_bs, const double* loc, int64_t loc_bs,
                     const double* cov, int64_t cov_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                     double* f_out, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = z; a.x_bs = z_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out = f_out;
    a.min_eig = min_eig;
    return run_lattice(c, OP_GP_INV_RFFT2, a);
}

int gpt_gp_rfft2_log_abs_det_jac(gpt_ctx* c, const double* cov, int64_t cov_bs, int64_t H, int64_t W,
                                 int64_t batch, int dim1, double* out, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.in = cov;
    e.in_bs = cov_bs;
    const int64_t n = H * W;
    int64_t nterms = (n - 1) / 2;
    if (H % 2 == 0 && W % 2 == 0) nterms -= 1;
    // fft2.stan:167-171 / fft1.stan:81
    const double add = -std::log(2.0) * (double)nterms + (double)n * std::log((double)n) / 2.0;
    return run_elem(c, EL_LADJ, e, true, 1, add, out, min_eig);
}

int gpt_gp_rfft2_log_abs_det_jac_vjp(gpt_ctx* c, const double* cov, int64_t cov_bs, int64_t H, int64_t W,
                                     int64_t batch, int dim1, double* grad_cov) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.in = cov;
    e.in_bs = cov_bs;
    e.out = grad_cov;
    return run_elem(c, EL_LADJ_VJP, e, false, 0, 0.0, nullptr, nullptr);
}

int gpt_gp_rfft2_lpdf(gpt_ctx* c, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                      const double* cov, int64_t cov_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                      double* lpdf, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = y; a.x_bs = y_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out_lpdf = lpdf;
    a.min_eig = min_eig;
    return run_lattice(c, OP_LPDF, a);
}

int gpt_gp_rfft2_lpdf_vjp(gpt_ctx* c, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                          const double* cov, int64_t cov_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                          double* lpdf, double* grad_y, double* grad_cov, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!grad_y) return fail(c, GPT_ERR_SHAPE, "grad_y must not be null (use gpt_gp_rfft2_lpdf for the value)");
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = y; a.x_bs = y_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out = grad_y;
    a.out_cov = grad_cov;
    a.out_lpdf = lpdf;
    a.min_eig = min_eig;
    return run_lattice(c, grad_cov ? OP_LPDF_VJP : OP_LPDF_VJP_NOCOV, a);
}

// gpt_gp_rfft2_lpdf_vjp for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid it):
// same arguments, same results, but twiddles, partials and workspace live in the built-in scratch array, so the call
// makes no hipMalloc / hipFree and uses none of the context's cached buffers. Every call on a GPU shares that GPU's
// array: calls that overlap in time on one GPU race (serialise them on one stream). Nothing in the array has to start
// at zero: each call regenerates its tables and every kernel writes what it later reads.
int gpt_gp_rfft2_lpdf_vjp_noalloc(gpt_ctx* c, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                                  const double* cov, int64_t cov_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                                  double* lpdf, double* grad_y, double* grad_cov, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!y || !cov || !lpdf || !grad_y) return fail(c, GPT_ERR_SHAPE, "y, cov, lpdf and grad_y must not be null");
    if (!c->scratch) return fail(c, GPT_ERR_HIP, "the built-in scratch array was not resolved on this device");
    HIP_TRY(c, hipSetDevice(c->device));
    if (batch == 0) return GPT_OK;
    const int op = grad_cov ? OP_LPDF_VJP : OP_LPDF_VJP_NOCOV;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = y; a.x_bs = y_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out = grad_y;
    a.out_cov = grad_cov;
    a.out_lpdf = lpdf;
    a.min_eig = min_eig;
    const int path = path_for_shape(H, W);
    if (path < 0) {
        char buf[160];
        snprintf(buf, sizeof buf, "no kernel for a %d x %d lattice in this build", a.H, a.W);
        return fail(c, GPT_ERR_UNSUPPORTED, buf);
    }
    // carve the scratch array: tables, then partials, then the workspace
    size_t off = 0;
    bool fits = true;
    auto take = [&](size_t bytes) -> unsigned char* {
        unsigned char* p = c->scratch + off;
        off += (bytes + 255) & ~(size_t)255;
        if (off > GPT_NOALLOC_SCRATCH_BYTES) fits = false;
        return p;
    };
    auto table = [&](int n, const cplx** out) -> int {
        cplx* t = reinterpret_cast<cplx*>(take(sizeof(cplx) * (size_t)n));
        if (!fits) return GPT_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(noalloc_twiddle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, t, n);
        *out = t;
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? GPT_OK : hip_fail(c, e, "twiddle table");
    };
    const char* too_big = "this lattice needs more than the 40 MiB of built-in scratch of the allocation-free entry point "
                          "(three-pass lattices up to 2048 x 2048); use gpt_gp_rfft2_lpdf_vjp";
    if (path == 3) // (its plans, tables and Bluestein scratch are allocated per shape)
        return fail(c, GPT_ERR_UNSUPPORTED,
                    "the allocation-free entry point does not take any-length shapes (long, prime or non-power-of-two "
                    "axes beyond 4096 points: their line plans and scratch are allocated); use gpt_gp_rfft2_lpdf_vjp");
    if ((rc = plan_radices(c, a.W, a.pw)) != GPT_OK || (rc = plan_radices(c, a.H, a.ph)) != GPT_OK) return rc;
    if ((rc = table(a.W, &a.pw.tw)) != GPT_OK) return fits ? rc : fail(c, GPT_ERR_UNSUPPORTED, too_big);
    if (a.H == a.W) a.ph.tw = a.pw.tw;
    else if ((rc = table(a.H, &a.ph.tw)) != GPT_OK) return fits ? rc : fail(c, GPT_ERR_UNSUPPORTED, too_big);
    if (path == 0) {
        HIP_TRY(c, gpt_launch_small(op, a, c->stream));
        return GPT_OK;
    }
    if (path == 1) {
        const int R = gpt_pow2_sub_radix(op, a.H, a.W);
        if (R && (rc = table(a.H / R, &a.tw_sub)) != GPT_OK) return fits ? rc : fail(c, GPT_ERR_UNSUPPORTED, too_big);
    }
    const size_t per = path == 1 ? gpt_pow2_workspace_bytes(a.H, a.W) : gpt_generic_workspace_bytes(a.H, a.W);
    const int nblk = path == 1 ? gpt_pow2_col_blocks(a.H, a.W, c->total_cus > 0 ? c->total_cus : 256)
                               : gpt_generic_col_blocks(a.H, a.W);
    if (c->total_cus == 0) c->total_cus = 256;
    const size_t per_lattice = ((per + 255) & ~(size_t)255) + sizeof(double) * GPT_PART * (size_t)nblk;
    const size_t left = off + 512 <= GPT_NOALLOC_SCRATCH_BYTES ? GPT_NOALLOC_SCRATCH_BYTES - off - 512 : 0;
    int64_t chunk = (int64_t)(left / per_lattice);
    if (chunk < 1) return fail(c, GPT_ERR_UNSUPPORTED, too_big);
    if (chunk > batch) chunk = batch;
    double* partials = reinterpret_cast<double*>(take(sizeof(double) * GPT_PART * (size_t)nblk * (size_t)chunk));
    cplx* ws = reinterpret_cast<cplx*>(take(per * (size_t)chunk));
    if (!fits) return fail(c, GPT_ERR_UNSUPPORTED, too_big);
    return run_units(c, op, a, path, chunk, ws, partials, false);
}

// prior_lp != null: the white-noise prior of the non-centred models rides along (zbar -= z, prior_lp[b] =
// std_normal_lpdf(z[b])): the column pass reads z and writes zbar anyway
static int inv_rfft2_vjp(gpt_ctx* c, const double* z, int64_t z_bs, const double* cov, int64_t cov_bs,
                         const double* fbar, int64_t fbar_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                         double* zbar, double* covbar, double* prior_lp) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = z; a.x_bs = z_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.aux = fbar; a.aux_bs = fbar_bs;
    a.out = zbar;
    a.out_cov = covbar;
    a.zprior = prior_lp ? 1 : 0;
    a.out_lpdf = prior_lp;
    if (!covbar) {
        if ((rc = ensure(c, c->sink, c->sink_bytes, sizeof(double) * (size_t)batch * H * (W / 2 + 1)))) return rc;
        a.out_cov = c->sink;
    }
    return run_lattice(c, OP_INV_VJP, a);
}

int gpt_gp_inv_rfft2_vjp(gpt_ctx* c, const double* z, int64_t z_bs, const double* cov, int64_t cov_bs,
                         const double* fbar, int64_t fbar_bs, int64_t H, int64_t W, int64_t batch, int dim1,
                         double* zbar, double* covbar) {
    return inv_rfft2_vjp(c, z, z_bs, cov, cov_bs, fbar, fbar_bs, H, W, batch, dim1, zbar, covbar, nullptr);
}

int gpt_gp_rfft2_vjp(gpt_ctx* c, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                     const double* cov, int64_t cov_bs, const double* zbar, int64_t zbar_bs, int64_t H, int64_t W,
                     int64_t batch, int dim1, double* ybar, double* covbar) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    LatArgs a = base_args(H, W, batch, dim1);
    a.x = y; a.x_bs = y_bs;
    a.loc = loc; a.loc_bs = loc_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.aux = zbar; a.aux_bs = zbar_bs;
    a.out = ybar;
    a.out_cov = covbar;
    if (!covbar) {
        if ((rc = ensure(c, c->sink, c->sink_bytes, sizeof(double) * (size_t)batch * H * (W / 2 + 1)))) return rc;
        a.out_cov = c->sink;
    }
    return run_lattice(c, OP_FWD_VJP, a);
}

// ---- spectral kernels ------------------------------------------------------------------------------
static double gamma_ratio(double nu, int dim1) {
    const double half_ndim = dim1 ? 0.5 : 1.0;
    return std::tgamma(nu + half_ndim) / std::tgamma(nu); // fft1.stan:173, fft2.stan:245
}

// Kernel-fused log density: lpdf(y | loc, kernel(params) + nugget) with the reverse mode down
// to the kernel parameters. Power-of-two lattices of 512..8192 rows evaluate the spectrum inside the column pass
// (OP_LPDF_KVJP: no cov read, no d/dcov write); every other shape composes the existing kernels on library-owned
// scratch (spectrum -> fused value + gradient -> kernel reverse mode), same results.
int gpt_gp_rfft2_lpdf_kernel_vjp(gpt_ctx* c, int matern, double nu, const double* y, int64_t y_bs, const double* loc,
                                 int64_t loc_bs, const double* params, double nugget, int64_t H, int64_t W,
                                 int64_t batch, int dim1, double* lpdf, double* grad_y, double* params_bar,
                                 double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!y || !params || !lpdf || !grad_y) return fail(c, GPT_ERR_SHAPE, "y, params, lpdf and grad_y must not be null");
    if (matern && !(nu > 0)) return fail(c, GPT_ERR_SHAPE, "nu must be positive");
    if (batch == 0) return GPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // the fused evaluation knows exp-quad and half-integer Matern smoothness (nu = 0.5, 1.5, ... 5.5)
    const bool nu_ok = !matern || (nu <= 5.5 && nu - std::floor(nu) == 0.5 /* kern_eval */);
    const int path = path_for_shape(H, W);
    const bool fused = !dim1 && nu_ok &&
                       (path == 0 || (path == 1 && gpt_pow2_sub_radix(OP_LPDF_KVJP, (int)H, (int)W) != 0));
    if (fused) {
        // the one-workgroup path leaves four sums per lattice for a combine launch; the three-pass path finishes
        // inside its last pass (finalize_block) and needs no scratch
        if (path == 0 && (rc = ensure(c, c->kscr, c->kscr_bytes, sizeof(double) * 4 * (size_t)batch)) != GPT_OK) return rc;
        LatArgs a = base_args(H, W, batch, 0);
        a.x = y; a.x_bs = y_bs;
        a.loc = loc; a.loc_bs = loc_bs;
        a.out = grad_y;
        a.min_eig = min_eig;
        a.kpar = params;
        a.kmatern = matern ? 1 : 0;
        a.knu = nu;
        a.kgamma = matern ? gamma_ratio(nu, 0) : 0.0;
        a.knugget = nugget;
        if (path == 0) {
            a.ksums = c->kscr;
        } else {
            a.out_lpdf = lpdf;
            a.kbar = params_bar;
        }
        if ((rc = run_lattice(c, matern ? OP_LPDF_KVJP : OP_LPDF_KVJP_EQ, a)) != GPT_OK) return rc;
        if (path == 0) HIP_TRY(c, gpt_launch_kvjp_combine(c->kscr, params, batch, lpdf, params_bar, c->stream));
        return GPT_OK;
    }
    // composed: cov and d/dcov live in scratch, a chunk of lattices at a time (<= ~256 MiB)
    const size_t hwf = (size_t)H * (size_t)(W / 2 + 1);
    int64_t chunk = (int64_t)((size_t)(128u << 20) / (hwf * sizeof(double)));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if ((rc = ensure(c, c->kscr, c->kscr_bytes, 2 * sizeof(double) * hwf * (size_t)chunk)) != GPT_OK) return rc;
    double* cov = c->kscr;
    double* dcov = c->kscr + hwf * (size_t)chunk;
    const int npar = dim1 ? 3 : 5;
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = batch - b0 < chunk ? batch - b0 : chunk;
        ElemArgs e = elem_args(H, W, nb, dim1);
        e.par = params + b0 * npar;
        e.out = cov;
        e.nu = nu;
        e.gamma_ratio = matern ? gamma_ratio(nu, dim1) : 0.0;
        e.nugget = nugget;
        if ((rc = run_elem(c, matern ? EL_MATERN : EL_EXPQUAD, e, false, 0, 0.0, nullptr, nullptr)) != GPT_OK) return rc;
        if ((rc = gpt_gp_rfft2_lpdf_vjp(c, y + b0 * y_bs, y_bs, loc ? loc + b0 * loc_bs : nullptr, loc_bs, cov, (int64_t)hwf,
                                        H, W, nb, dim1, lpdf + b0, grad_y + b0 * H * W, params_bar ? dcov : nullptr,
                                        min_eig ? min_eig + b0 : nullptr)) != GPT_OK)
            return rc;
        if (params_bar) {
            ElemArgs v = elem_args(H, W, nb, dim1);
            v.par = params + b0 * npar;
            v.in = dcov;
            v.in_bs = (int64_t)hwf;
            v.nu = nu;
            v.gamma_ratio = e.gamma_ratio;
            if ((rc = run_elem(c, matern ? EL_MATERN_VJP : EL_EXPQUAD_VJP, v, true, npar, 0.0, params_bar + b0 * npar,
                               nullptr)) != GPT_OK)
                return rc;
        }
    }
    return GPT_OK;
}

// Non-centred model density: f = gp_inv_rfft2(z, loc, cov) (fft2.stan:126-130), the observation
// model of the docs workloads on the device (gpt_obs.hip), the reverse mode of the transform, and optionally the
// white-noise prior of trees.stan:37. Three transforms' worth of passes + two streaming kernels; f and d/df never
// leave HBM.
int gpt_gp_inv_rfft2_obs_lpdf_vjp(gpt_ctx* c, int family, const double* z, int64_t z_bs, const double* loc,
                                  int64_t loc_bs, const double* cov, int64_t cov_bs, const double* obs, int64_t obs_bs,
                                  int64_t oh, int64_t ow, const double* theta, int prior, int64_t H, int64_t W,
                                  int64_t batch, int dim1, double* lp, double* f_out, double* grad_z, double* grad_loc,
                                  double* grad_loc_sum, double* grad_cov, double* grad_theta, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!z || !cov || !obs || !lp || !grad_z) return fail(c, GPT_ERR_SHAPE, "z, cov, obs, lp and grad_z must not be null");
    if (family != GPT_OBS_NORMAL && family != GPT_OBS_POISSON_LOG && family != GPT_OBS_NEG_BINOMIAL_2_LOG)
        return fail(c, GPT_ERR_SHAPE, "unknown observation family");
    if (family != GPT_OBS_POISSON_LOG && !theta) return fail(c, GPT_ERR_SHAPE, "this observation family needs theta");
    if (oh < 1 || ow < 1 || oh > H || ow > W)
        return fail(c, GPT_ERR_SHAPE, "the observed window must lie inside the lattice");
    if (batch == 0) return GPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t hw = H * W, hwf = H * (W / 2 + 1);
    // GPT_OBS_FUSED=1 (read per call): one row kernel does inverse rows -> observation model -> forward rows in place
    // on the workspace. Same results; measured slower than the three composed steps on MI355X (200
    // VGPRs leave two waves per SIMD to hide the densities' dependent chains), so it is off unless asked for.
    const char* fused_env = getenv("GPT_OBS_FUSED");
    const bool allow_fused = fused_env && fused_env[0] == '1';
    const int nblk = gpt_obs_blocks((int)H, (int)W);
    if (allow_fused && !dim1 && path_for_shape(H, W) == 1 && gpt_pow2_obs_supported((int)H, (int)W)) {
        if ((rc = ensure(c, c->oscr, c->oscr_bytes, sizeof(double) * 5 * (size_t)batch)) != GPT_OK) return rc;
        double* sums = c->oscr;
        double* psum = sums + 4 * batch;
        if ((rc = run_obs_fused(c, family, z, z_bs, loc, loc_bs, cov, cov_bs, obs, obs_bs, (int)oh, (int)ow, theta, (int)H,
                                (int)W, batch, f_out, grad_loc, grad_z, grad_cov, min_eig, sums, prior ? psum : nullptr)) !=
            GPT_OK)
            return rc;
        HIP_TRY(c, gpt_launch_obs_combine(sums, prior ? psum : nullptr, batch, lp, grad_theta, grad_loc_sum, nullptr,
                                          c->stream));
        return GPT_OK;
    }
    // composed: three library calls; lattices per round keep library-owned f / d/df at <= ~1 GiB together
    int64_t chunk = (int64_t)((size_t)(1u << 30) / (16 * (size_t)hw));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if (chunk > 32768) chunk = 32768;
    const size_t nscr = (f_out ? 0 : (size_t)chunk * hw) + (grad_loc ? 0 : (size_t)chunk * hw) + 5 * (size_t)chunk;
    if ((rc = ensure(c, c->oscr, c->oscr_bytes, sizeof(double) * nscr)) != GPT_OK) return rc;
    double* sums = c->oscr;
    double* psum = sums + 4 * chunk;
    double* scr = psum + chunk;
    double* f_scr = f_out ? nullptr : scr;
    double* g_scr = grad_loc ? nullptr : scr + (f_out ? 0 : (size_t)chunk * hw);
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = batch - b0 < chunk ? batch - b0 : chunk;
        double* f = f_out ? f_out + b0 * hw : f_scr;
        double* fbar = grad_loc ? grad_loc + b0 * hw : g_scr;
        const double* zb = z + b0 * z_bs;
        const double* cb = cov + b0 * cov_bs;
        if ((rc = gpt_gp_inv_rfft2(c, zb, z_bs, loc ? loc + b0 * loc_bs : nullptr, loc_bs, cb, cov_bs, H, W, nb, dim1, f,
                                   min_eig ? min_eig + b0 : nullptr)) != GPT_OK)
            return rc;
        if ((rc = ensure(c, c->partials, c->partials_bytes, sizeof(double) * GPT_PART * (size_t)nblk * (size_t)nb)) != GPT_OK)
            return rc;
        ObsArgs o;
        memset(&o, 0, sizeof o);
        o.H = (int)H; o.W = (int)W; o.oh = (int)oh; o.ow = (int)ow; o.family = family;
        o.batch = nb;
        o.f = f;
        o.obs = obs + b0 * obs_bs; o.obs_bs = obs_bs;
        o.theta = theta ? theta + b0 : nullptr;
        o.fbar = fbar;
        HIP_TRY(c, gpt_launch_obs(o, c->partials, c->stream));
        HIP_TRY(c, gpt_launch_finalize(c->partials, nblk, nb, 4, 0.0, sums, nullptr, c->stream));
        if ((rc = inv_rfft2_vjp(c, zb, z_bs, cb, cov_bs, fbar, hw, H, W, nb, dim1, grad_z + b0 * hw,
                                grad_cov ? grad_cov + b0 * hwf : nullptr, prior ? psum : nullptr)) != GPT_OK)
            return rc;
        HIP_TRY(c, gpt_launch_obs_combine(sums, prior ? psum : nullptr, nb, lp + b0, grad_theta ? grad_theta + b0 : nullptr,
                                          grad_loc_sum ? grad_loc_sum + b0 : nullptr, nullptr, c->stream));
    }
    return GPT_OK;
}

int gpt_gp_periodic_exp_quad_cov_rfft2(gpt_ctx* c, int64_t H, int64_t W, const double* params, int64_t batch,
                                       int dim1, double* out) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.par = params;
    e.out = out;
    return run_elem(c, EL_EXPQUAD, e, false, 0, 0.0, nullptr, nullptr);
}

int gpt_gp_periodic_matern_cov_rfft2(gpt_ctx* c, double nu, int64_t H, int64_t W, const double* params,
                                     int64_t batch, int dim1, double* out) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!(nu > 0)) return fail(c, GPT_ERR_SHAPE, "nu must be positive");
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.par = params;
    e.out = out;
    e.nu = nu;
    e.gamma_ratio = gamma_ratio(nu, dim1);
    return run_elem(c, EL_MATERN, e, false, 0, 0.0, nullptr, nullptr);
}

int gpt_gp_periodic_exp_quad_cov_rfft2_vjp(gpt_ctx* c, int64_t H, int64_t W, const double* params,
                                           const double* covbar, int64_t covbar_bs, int64_t batch, int dim1,
                                           double* params_bar) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.par = params;
    e.in = covbar;
    e.in_bs = covbar_bs;
    return run_elem(c, EL_EXPQUAD_VJP, e, true, dim1 ? 3 : 5, 0.0, params_bar, nullptr);
}

int gpt_gp_periodic_matern_cov_rfft2_vjp(gpt_ctx* c, double nu, int64_t H, int64_t W, const double* params,
                                         const double* covbar, int64_t covbar_bs, int64_t batch, int dim1,
                                         double* params_bar) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if (!(nu > 0)) return fail(c, GPT_ERR_SHAPE, "nu must be positive");
    ElemArgs e = elem_args(H, W, batch, dim1);
    e.par = params;
    e.in = covbar;
    e.in_bs = covbar_bs;
    e.nu = nu;
    e.gamma_ratio = gamma_ratio(nu, dim1);
    return run_elem(c, EL_MATERN_VJP, e, true, dim1 ? 3 : 5, 0.0, params_bar, nullptr);
}

} // extern "C"

// ===================================================================================================
// Host-pointer entry points: stage through a device arena, transposing Stan's column-major matrices.
namespace {

struct Stager {
    gpt_ctx* c;
    size_t off = 0;
    int layout;
    int rc = GPT_OK;
    bool drained = false;
    // Every exit of a host entry point, early error returns included, leaves no copy from or to the caller's
    // (pageable) arrays in flight.
    ~Stager() {
        if (!drained) (void)hipStreamSynchronize(c->stream);
    }
    // Small calls (the docs models: 35 x 60, n = 100 ... 200; everything up to TINY_BYTES of arena): what costs there is
    // the number of operations on the stream and above all the copy <-> kernel hand-overs (~12 us each), not bytes. So
    // these calls make NO copies: the caller's arrays are memcpy'd into the context's pinned, device-mapped bounce area
    // and the kernels read them there (the transposes of column-major matrices, or the transform kernels themselves);
    // outputs and scalars are written by the kernels straight into the bounce area and memcpy'd out after the one
    // synchronisation of the call. Only kernels on the stream: 64 x 64 column-major 225 -> 94 (one copy each way) -> ~50 us.
    static constexpr size_t TINY_BYTES = (size_t)64 << 20; // = BOUNCE_BYTES: lattices up to 512 x 512
    bool tiny = false;
    Stager(gpt_ctx* ctx, int lay, size_t need) : c(ctx), layout(lay) {
        if (hipSetDevice(c->device) != hipSuccess) {
            rc = fail(c, GPT_ERR_HIP, "hipSetDevice failed");
            return;
        }
        rc = ensure(c, c->arena, c->arena_bytes, need);
        tiny = rc == GPT_OK && need <= TINY_BYTES && bounce_ready() && c->bounce_dev && need <= c->bounce_bytes;
    }
    // the two views of the bounce area
    void* to_dev(char* host_ptr) { return c->bounce_dev + (host_ptr - c->bounce); }
    char* to_host(const void* dev_ptr) { return c->bounce + (static_cast<const char*>(dev_ptr) - c->bounce_dev); }
    bool is_matrix(int64_t R, int64_t C) const { return layout == GPT_COL_MAJOR && R > 1 && C > 1; }
    // transposes of a small call, gathered: inputs are launched together by go(), outputs by finish()
    XposeBatch xq{};
    int xq_esz = 0;
    void xflush() {
        if (xq.n == 0) return;
        const hipError_t e = gpt_launch_transpose_multi(xq, xq_esz, c->stream);
        if (e != hipSuccess && rc == GPT_OK) rc = hip_fail(c, e, "transpose");
        xq.n = 0;
    }
    void xpose(const void* in, void* out, int R, int C, int esz) {
        if (xq.n == GPT_XPOSE_MAX || (xq.n && xq_esz != esz)) xflush();
        xq_esz = esz;
        xq.in[xq.n] = in;
        xq.out[xq.n] = out;
        xq.R[xq.n] = R;
        xq.C[xq.n] = C;
        xq.n++;
    }
    // after the uploads and before the first kernel of the call: returns rc
    int go() {
        xflush();
        return rc;
    }
    // a scalar input next to device-side results (d from dev())
    void put(double* d, double v) {
        if (rc != GPT_OK) return;
        if (tiny) {
            *reinterpret_cast<double*>(to_host(d)) = v; // (the stream is idle between calls: nothing reads this yet)
            return;
        }
        c->pinned[1] = v; // (slot 0 is the min_eig read-back of finish())
        const hipError_t e = hipMemcpyAsync(d, c->pinned + 1, sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) rc = hip_fail(c, e, "scalar upload");
    }
    // large transfers go through the context's pinned ring (HostPipe), small ones through the runtime's pageable path
    bool piped(size_t bytes) {
        if (bytes < HostPipe::MIN_BYTES) return false;
        if (!c->pipe) c->pipe = new HostPipe();
        if (!c->pipe->ready() && c->pipe->init() != 0) {
            c->pipe->destroy();
            return false; // no pinned memory to be had: the plain path still works
        }
        return true;
    }
    // small transfers: a piece of the context's pinned bounce area (null: none to be had, the pageable path still works)
    static constexpr size_t BOUNCE_BYTES = 32 * HostPipe::MIN_BYTES; // 64 MiB
    size_t boff = 0;
    struct Pending {
        void* host;
        const char* src;
        size_t bytes;
    };
    std::vector<Pending> pending; // downloads that sit in the bounce area until finish() has synchronised
    bool bounce_ready() {
        if (c->bounce) return true;
        void* p = nullptr;
        if (hipHostMalloc(&p, BOUNCE_BYTES, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return false;
        c->bounce = static_cast<char*>(p);
        c->bounce_bytes = BOUNCE_BYTES;
        void* d = nullptr;
        c->bounce_dev = hipHostGetDevicePointer(&d, p, 0) == hipSuccess ? static_cast<char*>(d) : nullptr;
        return true;
    }
    char* bounce(size_t bytes) {
        if (!bounce_ready()) return nullptr;
        const size_t need = (bytes + 255) & ~(size_t)255;
        if (boff + need > c->bounce_bytes) {
            // (cannot happen for a small call: its arrays fit arena_need() <= the bounce area; the caller falls back or fails)
            if (tiny && rc == GPT_OK) rc = fail(c, GPT_ERR_ALLOC, "pinned bounce area exhausted");
            return tiny ? c->bounce : nullptr; // a valid address, never used once rc is set
        }
        char* p = c->bounce + boff;
        boff += need;
        return p;
    }
    hipError_t h2d(void* dev, const void* host, size_t bytes) {
        if (piped(bytes)) return c->pipe->up(dev, host, bytes, c->stream);
        if (char* b = bounce(bytes)) {
            memcpy(b, host, bytes);
            return hipMemcpyAsync(dev, b, bytes, hipMemcpyHostToDevice, c->stream);
        }
        return hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream);
    }
    hipError_t d2h(void* host, const void* dev, size_t bytes) {
        if (piped(bytes)) return c->pipe->down(host, dev, bytes, c->stream);
        if (char* b = bounce(bytes)) {
            pending.push_back({host, b, bytes});
            return hipMemcpyAsync(b, dev, bytes, hipMemcpyDeviceToHost, c->stream);
        }
        return hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream);
    }
    void* bump(size_t bytes) {
        void* p = c->arena + off;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    }
    // upload an R x C matrix of esz-byte elements; returns a row-major device copy
    const void* up(const void* host, int64_t R, int64_t C, int esz) {
        if (rc != GPT_OK || !host) return nullptr;
        const size_t bytes = (size_t)R * C * esz;
        if (tiny) {
            char* b = bounce(bytes);
            memcpy(b, host, bytes);
            if (!is_matrix(R, C)) return to_dev(b); // the kernels read it where it is
            void* d = bump(bytes);
            xpose(to_dev(b), d, (int)C, (int)R, esz);
            return d;
        }
        void* d = bump(bytes);
        hipError_t e;
        if (layout == GPT_COL_MAJOR && R > 1 && C > 1) {
            void* t = bump(bytes);
            e = h2d(t, host, bytes);
            if (e == hipSuccess) e = gpt_launch_transpose(t, d, (int)C, (int)R, 1, esz, c->stream);
        } else {
            e = h2d(d, host, bytes);
        }
        if (e != hipSuccess) rc = hip_fail(c, e, "upload");
        return d;
    }
    // room for an R x C result of the kernels (small calls: what needs no transpose lives in the bounce area already)
    void* dev(int64_t R, int64_t C, int esz) {
        const size_t bytes = (size_t)R * C * esz;
        if (tiny && !is_matrix(R, C)) return to_dev(bounce(bytes));
        return bump(bytes);
    }
    // download a row-major device R x C matrix into the caller's layout
    void down(void* host, void* d, int64_t R, int64_t C, int esz) {
        if (rc != GPT_OK || !host) return;
        const size_t bytes = (size_t)R * C * esz;
        hipError_t e;
        if (tiny) {
            if (is_matrix(R, C)) {
                char* b = bounce(bytes);
                xpose(d, to_dev(b), (int)R, (int)C, esz);
                pending.push_back({host, b, bytes});
            } else {
                pending.push_back({host, to_host(d), bytes}); // (d came from dev(): a piece of the bounce area)
            }
            return;
        }
        if (layout == GPT_COL_MAJOR && R > 1 && C > 1) {
            void* t = bump(bytes);
            e = gpt_launch_transpose(d, t, (int)R, (int)C, 1, esz, c->stream);
            if (e == hipSuccess) e = d2h(host, t, bytes);
        } else {
            e = d2h(host, d, bytes);
        }
        if (e != hipSuccess) rc = hip_fail(c, e, "download");
    }
    int finish(double* d_min_eig) {
        xflush();
        if (rc != GPT_OK) return rc;
        double mn = INFINITY;
        if (!tiny && d_min_eig) {
            hipError_t e = hipMemcpyAsync(c->pinned, d_min_eig, sizeof(double), hipMemcpyDeviceToHost, c->stream);
            if (e != hipSuccess) return hip_fail(c, e, "min_eig read-back");
        }
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return hip_fail(c, e, "hipStreamSynchronize");
        drained = true;
        for (const Pending& p : pending) memcpy(p.host, p.src, p.bytes);
        pending.clear();
        if (d_min_eig) {
            mn = tiny ? *reinterpret_cast<const double*>(to_host(d_min_eig)) : c->pinned[0];
            c->last_min_eig = mn;
            if (mn < 0) {
                char buf[128];
                snprintf(buf, sizeof buf, "covariance matrix is not positive-definite (minimum eigenvalue is %g)", mn);
                return fail(c, GPT_ERR_NOT_PD, buf);
            }
        }
        return GPT_OK;
    }
};

size_t arena_need(int64_t H, int64_t W, int narrays) {
    // every staged array may need a transposed twin; complex arrays count double
    return (size_t)narrays * 2 * ((size_t)H * W * sizeof(double) + 256) + 4096;
}

} // namespace

extern "C" {

int gpt_gp_rfft2_host(gpt_ctx* c, int layout, const double* y, const double* loc, const double* cov, int64_t H,
                      int64_t W, int dim1, double* z_out) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 5));
    const double* dy = (const double*)s.up(y, H, W, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    double* dz = (double*)s.dev(H, W, 8);
    double* dm = (double*)s.dev(1, 1, 8);
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_rfft2(c, dy, 0, dl, 0, dc, 0, H, W, 1, dim1, dz, dm))) return rc;
    s.down(z_out, dz, H, W, 8);
    return s.finish(dm);
}

int gpt_gp_inv_rfft2_host(gpt_ctx* c, int layout, const double* z, const double* loc, const double* cov,
                          int64_t H, int64_t W, int dim1, double* f_out) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 5));
    const double* dz = (const double*)s.up(z, H, W, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    double* df = (double*)s.dev(H, W, 8);
    double* dm = (double*)s.dev(1, 1, 8);
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_inv_rfft2(c, dz, 0, dl, 0, dc, 0, H, W, 1, dim1, df, dm))) return rc;
    s.down(f_out, df, H, W, 8);
    return s.finish(dm);
}

int gpt_gp_inv_rfft2_obs_lpdf_host(gpt_ctx* c, int layout, int family, const double* z, const double* loc,
                                   double loc_scalar, const double* cov, const double* obs, int64_t oh, int64_t ow,
                                   double theta, int prior, int64_t H, int64_t W, int dim1, double* lp, double* grad_z,
                                   double* grad_loc_sum, double* grad_cov, double* grad_theta) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    if (!z || !cov || !obs || !lp || !grad_z) return fail(c, GPT_ERR_SHAPE, "z, cov, obs, lp and grad_z must not be null");
    if (oh < 1 || ow < 1 || oh > H || ow > W)
        return fail(c, GPT_ERR_SHAPE, "the observed window must lie inside the lattice");
    // Stan Math rejects these (normal_lpdf: "Scale parameter", neg_binomial_2_lpmf: "Precision parameter")
    if (family != GPT_OBS_POISSON_LOG && !(theta > 0 && std::isfinite(theta)))
        return fail(c, GPT_ERR_SHAPE, family == GPT_OBS_NORMAL ? "Scale parameter must be positive finite"
                                                               : "Precision parameter must be positive finite");
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 8));
    const double* dz = (const double*)s.up(z, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    const double* dob = (const double*)s.up(obs, oh, ow, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    double* dfill = loc ? nullptr : (double*)s.dev(H, W, 8);
    double* dgz = (double*)s.dev(H, W, 8);
    double* dgc = grad_cov ? (double*)s.dev(H, Wf, 8) : nullptr;
    double* dsc = (double*)s.dev(1, 8, 8); // theta, lp, d/dloc sum, d/dtheta, min_eig
    if (s.go()) return s.rc;
    if (dfill) {
        HIP_TRY(c, gpt_launch_fill(dfill, H * W, loc_scalar, c->stream));
        dl = dfill;
    }
    s.put(dsc, theta);
    if (s.rc) return s.rc;
    if ((rc = gpt_gp_inv_rfft2_obs_lpdf_vjp(c, family, dz, 0, dl, 0, dc, 0, dob, 0, oh, ow,
                                            family == GPT_OBS_POISSON_LOG ? nullptr : dsc, prior, H, W, 1, dim1, dsc + 1,
                                            nullptr, dgz, nullptr, dsc + 2, dgc, dsc + 3, dsc + 4)))
        return rc;
    s.down(grad_z, dgz, H, W, 8);
    s.down(grad_cov, dgc, H, Wf, 8);
    s.down(lp, dsc + 1, 1, 1, 8);
    s.down(grad_loc_sum, dsc + 2, 1, 1, 8);
    if (family != GPT_OBS_POISSON_LOG) s.down(grad_theta, dsc + 3, 1, 1, 8);
    else if (grad_theta) *grad_theta = 0.0;
    return s.finish(dsc + 4);
}

int gpt_gp_rfft2_log_abs_det_jac_host(gpt_ctx* c, int layout, const double* cov, int64_t H, int64_t W, int dim1,
                                      double* out, double* grad_cov) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 4));
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    double* dout = (double*)s.dev(1, 1, 8);
    double* dm = (double*)s.dev(1, 1, 8);
    double* dg = grad_cov ? (double*)s.dev(H, Wf, 8) : nullptr;
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_rfft2_log_abs_det_jac(c, dc, 0, H, W, 1, dim1, dout, dm))) return rc;
    if (grad_cov && (rc = gpt_gp_rfft2_log_abs_det_jac_vjp(c, dc, 0, H, W, 1, dim1, dg))) return rc;
    s.down(out, dout, 1, 1, 8);
    if (grad_cov) s.down(grad_cov, dg, H, Wf, 8);
    return s.finish(dm);
}

int gpt_gp_rfft2_lpdf_host(gpt_ctx* c, int layout, const double* y, const double* loc, const double* cov,
                           int64_t H, int64_t W, int dim1, double* lpdf, double* grad_y, double* grad_cov) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 7));
    const double* dy = (const double*)s.up(y, H, W, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    double* dlp = (double*)s.dev(1, 1, 8);
    double* dm = (double*)s.dev(1, 1, 8);
    double* dgy = grad_y ? (double*)s.dev(H, W, 8) : nullptr;
    double* dgc = (grad_y && grad_cov) ? (double*)s.dev(H, Wf, 8) : nullptr;
    if (s.go()) return s.rc;
    if (grad_y) rc = gpt_gp_rfft2_lpdf_vjp(c, dy, 0, dl, 0, dc, 0, H, W, 1, dim1, dlp, dgy, dgc, dm);
    else rc = gpt_gp_rfft2_lpdf(c, dy, 0, dl, 0, dc, 0, H, W, 1, dim1, dlp, dm);
    if (rc) return rc;
    s.down(lpdf, dlp, 1, 1, 8);
    if (grad_y) s.down(grad_y, dgy, H, W, 8);
    if (dgc) s.down(grad_cov, dgc, H, Wf, 8);
    return s.finish(dm);
}

int gpt_gp_inv_rfft2_vjp_host(gpt_ctx* c, int layout, const double* z, const double* cov, const double* fbar,
                              int64_t H, int64_t W, int dim1, double* zbar, double* covbar) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 7));
    const double* dz = (const double*)s.up(z, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    const double* dfb = (const double*)s.up(fbar, H, W, 8);
    double* dzb = (double*)s.dev(H, W, 8);
    double* dcb = covbar ? (double*)s.dev(H, Wf, 8) : nullptr;
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_inv_rfft2_vjp(c, dz, 0, dc, 0, dfb, 0, H, W, 1, dim1, dzb, dcb))) return rc;
    s.down(zbar, dzb, H, W, 8);
    if (covbar) s.down(covbar, dcb, H, Wf, 8);
    return s.finish(nullptr);
}

int gpt_gp_rfft2_vjp_host(gpt_ctx* c, int layout, const double* y, const double* loc, const double* cov,
                          const double* zbar, int64_t H, int64_t W, int dim1, double* ybar, double* covbar) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 8));
    const double* dy = (const double*)s.up(y, H, W, 8);
    const double* dl = (const double*)s.up(loc, H, W, 8);
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    const double* dzb = (const double*)s.up(zbar, H, W, 8);
    double* dyb = (double*)s.dev(H, W, 8);
    double* dcb = covbar ? (double*)s.dev(H, Wf, 8) : nullptr;
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_rfft2_vjp(c, dy, 0, dl, 0, dc, 0, dzb, 0, H, W, 1, dim1, dyb, dcb))) return rc;
    s.down(ybar, dyb, H, W, 8);
    if (covbar) s.down(covbar, dcb, H, Wf, 8);
    return s.finish(nullptr);
}

// ---- host-pointer versions of the real-FFT layer and of the layout helpers (double only) ---------------
int gpt_rfft2_host(gpt_ctx* c, int layout, const double* y, int64_t H, int64_t W, double* out_complex) {
    int rc = check_shape(c, H, W, 1, 0);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 6));
    const double* dy = (const double*)s.up(y, H, W, 8);
    double* dout = (double*)s.dev(H, Wf, 16);
    if (s.go()) return s.rc;
    if ((rc = gpt_rfft2(c, dy, H, W, 1, dout))) return rc;
    s.down(out_complex, dout, H, Wf, 16);
    return s.finish(nullptr);
}

int gpt_inv_rfft2_host(gpt_ctx* c, int layout, const double* z_complex, int64_t H, int64_t W, double* out) {
    int rc = check_shape(c, H, W, 1, 0);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 8));
    const double* dz = (const double*)s.up(z_complex, H, Wf, 16);
    double* dout = (double*)s.dev(H, W, 8);
    if (s.go()) return s.rc;
    if ((rc = gpt_inv_rfft2(c, dz, H, W, 1, dout))) return rc;
    s.down(out, dout, H, W, 8);
    return s.finish(nullptr);
}

int gpt_expand_rfft_host(gpt_ctx* c, const double* y_complex, int64_t n, double* out_complex) {
    int rc = check_shape(c, 1, n, 1, 1);
    if (rc) return rc;
    Stager s(c, GPT_ROW_MAJOR, arena_need(1, n, 8));
    const double* dy = (const double*)s.up(y_complex, 1, n / 2 + 1, 16);
    double* dout = (double*)s.dev(1, n, 16);
    if (s.go()) return s.rc;
    if ((rc = gpt_expand_rfft(c, dy, n, 1, dout))) return rc;
    s.down(out_complex, dout, 1, n, 16);
    return s.finish(nullptr);
}

int gpt_gp_evaluate_rfft2_scale_host(gpt_ctx* c, int layout, const double* cov, int64_t H, int64_t W, int dim1,
                                     double* out) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 4));
    const double* dc = (const double*)s.up(cov, H, Wf, 8);
    double* dout = (double*)s.dev(H, Wf, 8);
    double* dm = (double*)s.dev(1, 1, 8);
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_evaluate_rfft2_scale(c, dc, 0, H, W, 1, dim1, dout, dm))) return rc;
    s.down(out, dout, H, Wf, 8);
    return s.finish(dm);
}

int gpt_gp_unpack_rfft2_host(gpt_ctx* c, int layout, const double* z_complex, int64_t H, int64_t W, int dim1,
                             double* out) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 8));
    const double* dz = (const double*)s.up(z_complex, H, Wf, 16);
    double* dout = (double*)s.dev(H, W, 8);
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_unpack_rfft2(c, dz, H, W, 1, dim1, dout))) return rc;
    s.down(out, dout, H, W, 8);
    return s.finish(nullptr);
}

int gpt_gp_pack_rfft2_host(gpt_ctx* c, int layout, const double* z, int64_t H, int64_t W, int dim1,
                           double* out_complex) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 8));
    const double* dz = (const double*)s.up(z, H, W, 8);
    double* dout = (double*)s.dev(H, Wf, 16);
    if (s.go()) return s.rc;
    if ((rc = gpt_gp_pack_rfft2(c, dz, H, W, 1, dim1, dout))) return rc;
    s.down(out_complex, dout, H, Wf, 16);
    return s.finish(nullptr);
}

static int spec_host(gpt_ctx* c, int layout, bool matern, double nu, int64_t H, int64_t W, const double* params,
                     int dim1, double* out) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 3));
    const double* dp = (const double*)s.up(params, 1, dim1 ? 3 : 5, 8);
    double* dout = (double*)s.dev(H, Wf, 8);
    if (s.go()) return s.rc;
    rc = matern ? gpt_gp_periodic_matern_cov_rfft2(c, nu, H, W, dp, 1, dim1, dout)
                : gpt_gp_periodic_exp_quad_cov_rfft2(c, H, W, dp, 1, dim1, dout);
    if (rc) return rc;
    s.down(out, dout, H, Wf, 8);
    return s.finish(nullptr);
}

static int spec_vjp_host(gpt_ctx* c, int layout, bool matern, double nu, int64_t H, int64_t W,
                         const double* params, const double* covbar, int dim1, double* params_bar) {
    int rc = check_shape(c, H, W, 1, dim1);
    if (rc) return rc;
    const int64_t Wf = W / 2 + 1;
    Stager s(c, layout, arena_need(H, W, 4));
    const double* dp = (const double*)s.up(params, 1, dim1 ? 3 : 5, 8);
    const double* dcb = (const double*)s.up(covbar, H, Wf, 8);
    double* dpb = (double*)s.dev(1, 8, 8);
    if (s.go()) return s.rc;
    rc = matern ? gpt_gp_periodic_matern_cov_rfft2_vjp(c, nu, H, W, dp, dcb, 0, 1, dim1, dpb)
                : gpt_gp_periodic_exp_quad_cov_rfft2_vjp(c, H, W, dp, dcb, 0, 1, dim1, dpb);
    if (rc) return rc;
    s.down(params_bar, dpb, 1, dim1 ? 3 : 5, 8);
    return s.finish(nullptr);
}

int gpt_gp_periodic_exp_quad_cov_rfft2_host(gpt_ctx* c, int layout, int64_t H, int64_t W, const double* params,
                                            int dim1, double* out) {
    return spec_host(c, layout, false, 0.0, H, W, params, dim1, out);
}
int gpt_gp_periodic_matern_cov_rfft2_host(gpt_ctx* c, int layout, double nu, int64_t H, int64_t W,
                                          const double* params, int dim1, double* out) {
    return spec_host(c, layout, true, nu, H, W, params, dim1, out);
}
int gpt_gp_periodic_exp_quad_cov_rfft2_vjp_host(gpt_ctx* c, int layout, int64_t H, int64_t W,
                                                const double* params, const double* covbar, int dim1,
                                                double* params_bar) {
    return spec_vjp_host(c, layout, false, 0.0, H, W, params, covbar, dim1, params_bar);
}
int gpt_gp_periodic_matern_cov_rfft2_vjp_host(gpt_ctx* c, int layout, double nu, int64_t H, int64_t W,
                                              const double* params, const double* covbar, int dim1,
                                              double* params_bar) {
    return spec_vjp_host(c, layout, true, nu, H, W, params, covbar, dim1, params_bar);
}

} // extern "C"
