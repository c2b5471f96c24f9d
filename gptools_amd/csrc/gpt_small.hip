// Small-lattice path: ONE workgroup owns ONE lattice of the batch and keeps its whole packed spectrum in
// LDS, so every Fourier-GP function is a single launch over the batch (grid = batch). Any H, W (mixed
// radix Stockham, arbitrary prime factors) as long as two copies of the lattice fit the 160 KB LDS of a
// CDNA4 CU. This is the path of the docs workloads (35x60 trees lattice, 1-D n = 100..192, batches of
// chains) and of every odd/even parity case of the reference's tests.
//
// Phases (each a grid-stride loop over LDS, separated by workgroup barriers):
//   load   two real rows -> one complex line  (two-for-one real transform, any W)
//   rows   Stockham line FFTs of length W
//   split  separate the row pairs, pack DC + i*Nyquist into column 0      -> packed spectrum S[H][NC]
//   cols   Stockham line FFTs of length H down the NC packed columns
//   ew     op-specific elementwise stage (gpt_common.h), wave/workgroup reduction for the log density
//   cols^-1, merge row pairs, rows^-1, store
#include <atomic>
#include <type_traits>
#include <cstdlib>

#include "gpt_common.h"
#include "gpt_host.h"
// GPT_SMALL_STAMPS (a variant build, scripts/small_stamps.py): workgroup 0 leaves the cycle counter at every phase boundary
#ifdef GPT_SMALL_STAMPS
__device__ long long gpt_small_stamps[32];
// ... and after every Stockham stage: [0] = number of entries, then (radix, cycle counter) pairs
__device__ long long gpt_small_stage_stamps[64];
// (every stage stamp costs a global round trip on the critical path -- GPT_SMALL_STAMPS=3 only: the totals of such a
// run are not those of the kernel)
#define GPT_STAGE_STAMP(R)                                                                                            \
    do {                                                                                                              \
        if (GPT_SMALL_STAMPS >= 3 && blockIdx.x == 0 && threadIdx.x == 0) {                                                                    \
            const long long i_ = gpt_small_stage_stamps[0];                                                           \
            if (i_ < 31) {                                                                                            \
                gpt_small_stage_stamps[1 + 2 * i_] = (R);                                                             \
                gpt_small_stage_stamps[2 + 2 * i_] = __builtin_readcyclecounter();                                    \
                gpt_small_stage_stamps[0] = i_ + 1;                                                                   \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
// (the same from the workgroup's LAST thread: the waves at the tail run the packed column)
#define STAMP_LAST(k)                                                                                                 \
    do {                                                                                                              \
        if (stamp_on_ && blockIdx.x == 0 && threadIdx.x == blockDim.x - 1) gpt_small_stamps[k] = __builtin_readcyclecounter(); \
    } while (0)
#define STAMP(k)                                                                                                      \
    do {                                                                                                              \
        if (stamp_on_ && blockIdx.x == 0 && threadIdx.x == 0) gpt_small_stamps[k] = (k) >= 16 ? wall_clock64() : __builtin_readcyclecounter(); \
    } while (0)
extern "C" int gpt_debug_small_stamps(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gpt_small_stamps), (size_t)n * sizeof(long long));
}
extern "C" int gpt_debug_small_stage_stamps(long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gpt_small_stage_stamps), (size_t)n * sizeof(long long));
}
#else
#define STAMP(k)
#define STAMP_LAST(k)
#endif

// The line transforms are inlined here (nowhere else): as a called function the stage loop took its plan through a stack
// copy, spilled 13-44 VGPRs around each of the four calls and read scratch back before every phase -- all of it latency on
// a path that is nothing but latency (one 35 x 60 lattice: ~20 barrier-separated phases).
#define GPT_LDS_FFT_INLINE __forceinline__
#include "gpt_lds_fft.h"
#include "gpt_small_phases.h"

namespace {

template <int OP>
__global__ void __launch_bounds__(1024) small_lattice_kernel(LatArgs a, int cap, int sS, SmallStages ss) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef GPT_SMALL_STAMPS
    // GPT_SMALL_STAMPS=2: the whole body twice, stamps from the second time round (instruction cache warm)
    for (int rep_ = 0; rep_ < GPT_SMALL_STAMPS; ++rep_) {
    const bool stamp_on_ = rep_ == GPT_SMALL_STAMPS - 1;
    __syncthreads();
#endif
    double* red = reinterpret_cast<double*>(smem); // 5 x 16 doubles (GPT_SMALL_HDR bytes)
    // (LDS-typed pointers, gpt_lds_fft.h: every access below is a ds_read / ds_write, never a flat one)
    const LdsLine buf0(smem + GPT_SMALL_HDR);
    const LdsLine buf1 = buf0 + cap;
    const int64_t b = blockIdx.x;
    const int H = a.H, W = a.W;
    STAMP(0);
    STAMP(16);
#ifdef GPT_SMALL_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) gpt_small_stage_stamps[0] = 0;
#endif
    const ZLayout L = make_layout(a);
    // (the per-lattice pointers are made where the elementwise stage starts, not here: fewer scalar registers alive across
    // the transforms, which spill otherwise. The kernel-fused ops need theirs now: kern_prepare's loads and divisions
    // then overlap with the forward transform.)
    EwPtrs P_early = make_ptrs(a, b);
    if (op_is_kvjp(OP)) kern_prepare<OP>(a, P_early);
    const int NC = 1 + L.wc;
    const int Hp = (H + 1) / 2;
    const bool weven = (W % 2) == 0;
    LdsLine cur = buf0;
    LdsLine oth = buf1;
    // (index arithmetic of the element loops: multiply-high instead of integer division, gpt_lds_fft.h)
    const unsigned mW = ss.mW, mNC = ss.mNC, mwc = ss.mwc;
    // (Twiddle tables copied into LDS behind the line buffers were measured, profiles/r03_k_small_tw_lds_ab.jsonl: no faster --
    // 16.30 against 16.24 us for one 35 x 60 lattice, 0.4 us slower for the 1-D sizes -- so they stay where the plan has them.)
    auto fft_w = [&](auto inv, LdsLine src, LdsLine dst, int nl, int es, int ls) -> LdsLine {
        return fft_lines_lds_tw<decltype(inv)::value, const cplx*>(src, dst, a.pw, a.pw.tw, nl, es, ls, ss.w, ss.mLw);
    };
    auto fft_h = [&](auto inv, LdsLine src, LdsLine dst, int nl, int es, int ls) -> LdsLine {
        return fft_lines_lds_tw<decltype(inv)::value, const cplx*>(src, dst, a.ph, a.ph.tw, nl, es, ls, ss.h, ss.mLh);
    };

    // cov of this thread's first elementwise entries, asked for now: by the time the transforms are through they are in
    // registers, instead of a memory round trip in the middle of the kernel
    SmallCovPre pre = {1.0, 1.0, 1.0, 1.0};
    if (small_uses_cov<OP>()) pre = small_cov_prefetch(a.cov + b * a.cov_bs, L, H, W, mwc, weven);
    STAMP(1);
    if (op_has_fwd(OP)) {
        // load: line r = rows (2r, 2r+1)
        // (two entries per trip, all their loads asked for before the first is used: one memory round trip for a
        // 35 x 60 lattice on 576 threads instead of two -- a round trip is 0.5 us in HBM and 2.5 us when the host-pointer
        // entry points let the kernel read the caller's arrays in pinned host memory)
        for (int w = threadIdx.x; w < Hp * W; w += 2 * blockDim.x) {
            const int w1 = w + (int)blockDim.x;
            const bool two = w1 < Hp * W;
            const int r = (int)fastdiv((unsigned)w, (unsigned)W, mW), j = w - r * W;
            const int r1 = two ? (int)fastdiv((unsigned)w1, (unsigned)W, mW) : r, j1 = two ? w1 - r1 * W : j;
            const double re = in_transform<OP>(a, b, 2 * r, j);
            const double im = (2 * r + 1 < H) ? in_transform<OP>(a, b, 2 * r + 1, j) : 0.0;
            const double re1 = in_transform<OP>(a, b, 2 * r1, j1);
            const double im1 = (2 * r1 + 1 < H) ? in_transform<OP>(a, b, 2 * r1 + 1, j1) : 0.0;
            cur[w] = make_double2(re, im);
            if (two) cur[w1] = make_double2(re1, im1);
        }
        __syncthreads();
        STAMP(2);
        LdsLine res = fft_w(std::false_type{}, cur, oth, Hp, 1, W);
        if (res != cur) {
            oth = cur;
            cur = res;
        }
        STAMP(3);
        // split row pairs -> packed spectrum S[i * sS + jp] in `oth`
        small_split(cur, oth, H, W, NC, sS, mNC, weven);
        __syncthreads();
        {
            const LdsLine t = cur;
            cur = oth;
            oth = t;
        }
        STAMP(4);
        res = fft_h(std::false_type{}, cur, oth, NC, sS, 1);
        if (res != cur) {
            oth = cur;
            cur = res;
        }
        STAMP(5);
    }

    // elementwise stage on S = cur
    STAMP_LAST(20);
    EwPtrs P = op_is_kvjp(OP) ? P_early : make_ptrs(a, b);
    EwAcc acc;
    acc_init(acc);
    small_interior<OP>(a, L, P, cur, sS, H, mwc, pre, acc);
    STAMP(6);
    STAMP_LAST(21);
    small_packed0<OP>(a, L, P, cur, sS, H, W, weven, pre, acc); // (on the tail threads: beside the interior, not after it)
    STAMP(7);
    STAMP_LAST(22);
    // Reduction of the log density (and the kernel-fused op's three sums): every wave leaves its sums in `red`, the
    // barrier that separates the elementwise stage from the inverse transform anyway makes them visible, and the LAST
    // thread (its wave has no butterflies in the stages that follow) adds them up -- off everybody else's critical path.
    constexpr int NRED = (OP == OP_RFFT2 || OP == OP_INV_RFFT2) ? 0 : op_is_kvjp(OP) ? 5 : 2;
    const int nw = ((int)blockDim.x + 63) >> 6, wv = (int)threadIdx.x >> 6;
    if (NRED) {
        const double s = wave_sum(acc.sum), m = wave_min(acc.mn);
        double kk[3] = {0.0, 0.0, 0.0};
        if (NRED == 5) {
            kk[0] = wave_sum(acc.k0);
            kk[1] = wave_sum(acc.k1);
            kk[2] = wave_sum(acc.k2);
        }
        if ((threadIdx.x & 63) == 0) {
            red[wv] = s;
            red[16 + wv] = m;
            if (NRED == 5) {
                red[32 + wv] = kk[0];
                red[48 + wv] = kk[1];
                red[64 + wv] = kk[2];
            }
        }
    }
    __syncthreads();
    if (NRED && threadIdx.x == blockDim.x - 1) {
        double s = 0.0, m = INFINITY, kk[3] = {0.0, 0.0, 0.0};
        for (int i = 0; i < nw; ++i) {
            s += red[i];
            m = fmin(m, red[16 + i]);
            if (NRED == 5) {
                kk[0] += red[32 + i];
                kk[1] += red[48 + i];
                kk[2] += red[64 + i];
            }
        }
        const double lp = s - 0.5 * (double)H * (double)W * 1.8378770664093454836;
        if ((OP == OP_LPDF || op_is_lpdf_vjp(OP) || OP == OP_INV_VJP) && a.out_lpdf) a.out_lpdf[b] = lp;
        if (op_is_kvjp(OP) && a.kbar) { // the three sums -> d / d(sigma, length_scale, period), as finalize_block (gpt_common.h)
            small_kernel_adjoints(a, b, kk[0], kk[1], kk[2], a.kbar);
        }
        if (a.min_eig) a.min_eig[b] = m;
    }
    STAMP(8);
    STAMP_LAST(23);

    if (op_has_inv(OP)) {
        LdsLine res = fft_h(std::true_type{}, cur, oth, NC, sS, 1);
        if (res != cur) {
            oth = cur;
            cur = res;
        }
        STAMP(9);
        // merge row pairs into complex lines of length W in `oth`
        small_merge(cur, oth, H, W, Hp, sS, mW, weven);
        __syncthreads();
        {
            const LdsLine t = cur;
            cur = oth;
            oth = t;
        }
        STAMP(10);
        res = fft_w(std::true_type{}, cur, oth, Hp, 1, W);
        STAMP(11);
        for (int w = threadIdx.x; w < Hp * W; w += blockDim.x) {
            const int r = (int)fastdiv((unsigned)w, (unsigned)W, mW), j = w - r * W;
            const cplx v = res[w];
            double* o = a.out + b * (int64_t)H * W;
            o[(int64_t)(2 * r) * W + j] = out_transform<OP>(a, b, 2 * r, j, v.x);
            if (2 * r + 1 < H) o[(int64_t)(2 * r + 1) * W + j] = out_transform<OP>(a, b, 2 * r + 1, j, v.y);
        }
    }
    if (ss.done_flag) { // (uniform; one workgroup: batch == 1)
        // every thread's results are out at system scope, then ONE release store of the sequence number the host spins on
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(ss.done_flag, ss.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    STAMP(12);
    STAMP(17);
#ifdef GPT_SMALL_STAMPS
    }
#endif
}

template <int OP>
hipError_t launch_small_t(const LatArgs& a, int cap, int sS, int nthreads, size_t lds, hipStream_t st,
                          unsigned long long* done_flag, unsigned long long done_seq) {
    const SmallStages ss = make_small_stages(a, done_flag, done_seq);
    // The attribute belongs to (kernel, device): one bit per device, set from any thread (raising the cap twice is
    // harmless, so a lost race costs one extra call).
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev < 64 ? (uint64_t)1 << dev : 0;
    if (lds > 64 * 1024 && !(attr_set.load(std::memory_order_relaxed) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&small_lattice_kernel<OP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((small_lattice_kernel<OP>), dim3((unsigned)a.batch), dim3(nthreads), lds, st, a, cap, sS, ss);
    return hipGetLastError();
}

} // namespace

bool gpt_small_fits(int H, int W, int* cap_out, int* stride_out) {
    const int NC = 1 + (W - 1) / 2;
    const int Hp = (H + 1) / 2;
    // odd stride between the rows of the packed spectrum (column walks along a line hit distinct LDS banks) where it
    // fits; the bare NC where only that fits -- 100 x 100: 5100 elements against 5000, the difference between this
    // kernel and three launches of the generic path (the stages number their work items across the columns, so an even
    // stride costs the split / merge / packed-column loops a bank conflict, not the transforms)
    for (int pad = 1; pad >= 0; --pad) {
        const int sS = pad ? (NC | 1) : NC;
        int64_t cap = (int64_t)Hp * W;
        if ((int64_t)H * sS > cap) cap = (int64_t)H * sS;
        if (cap_out) *cap_out = (int)cap;
        if (stride_out) *stride_out = sS;
        if (GPT_SMALL_HDR + 2 * cap * (int64_t)sizeof(cplx) <= 160 * 1024) return true;
    }
    return false;
}

int gpt_small_threads(int H, int W, int64_t batch) {
    int64_t work = (int64_t)H * W / 4;
    int nthreads = (int)((work + 63) / 64 * 64);
    if (nthreads < 64) nthreads = 64;
    // short and wide (the 1-D functions: H = 1): the element loops run over W entries and are most of such a call -- a
    // thread per entry up to 256 (n = 160, 64 chains: 12.25 us at 64 threads, 10.7 at 192; profiles/r03_k_small_threads.jsonl)
    const int wide = W < 256 ? (W + 63) / 64 * 64 : 256;
    if (nthreads < wide) nthreads = wide;
    if (nthreads > 1024) nthreads = 1024;
    // Many lattices per launch: throughput, not latency -- four-wave workgroups let four lattices share a CU
    // (measured at 35 x 60 with 4096 chains: 369 us at 576 threads, 166 us at 256; with 64 chains 24.5 against 26.9)
    if (batch >= 512 && nthreads > 256) nthreads = 256;
    static const int forced = [] { // GPT_SMALL_THREADS=<n>: tuning experiments
        const char* e = getenv("GPT_SMALL_THREADS");
        return e ? atoi(e) : 0;
    }();
    if (forced >= 64 && forced <= 1024) nthreads = forced / 64 * 64;
    return nthreads;
}

hipError_t gpt_launch_small(int op, const LatArgs& a, hipStream_t st, unsigned long long* done_flag,
                            unsigned long long done_seq) {
    int cap, sS;
    if (!gpt_small_fits(a.H, a.W, &cap, &sS)) return hipErrorInvalidValue;
    if (a.batch <= 0) return hipSuccess;
    size_t lds = GPT_SMALL_HDR + 2 * (size_t)cap * sizeof(cplx);
    const int nthreads = gpt_small_threads(a.H, a.W, a.batch);
    switch (op) {
    case OP_RFFT2: return launch_small_t<OP_RFFT2>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    case OP_INV_RFFT2: return launch_small_t<OP_INV_RFFT2>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    case OP_GP_RFFT2: return launch_small_t<OP_GP_RFFT2>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    case OP_GP_INV_RFFT2: return launch_small_t<OP_GP_INV_RFFT2>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    case OP_LPDF: return launch_small_t<OP_LPDF>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    case OP_LPDF_VJP: return launch_small_t<OP_LPDF_VJP>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    case OP_INV_VJP: return launch_small_t<OP_INV_VJP>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    case OP_FWD_VJP: return launch_small_t<OP_FWD_VJP>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    case OP_LPDF_VJP_NOCOV: return launch_small_t<OP_LPDF_VJP_NOCOV>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    case OP_LPDF_KVJP: return launch_small_t<OP_LPDF_KVJP>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    case OP_LPDF_KVJP_EQ: return launch_small_t<OP_LPDF_KVJP_EQ>(a, cap, sS, nthreads, lds, st, done_flag, done_seq);
    default: return hipErrorInvalidValue;
    }
}
