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
#include <cstdlib>

#include "gpt_common.h"
#include "gpt_host.h"
#include "gpt_lds_fft.h"

namespace {

template <int OP>
__global__ void __launch_bounds__(1024) small_lattice_kernel(LatArgs a, int cap, int sS) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* red = reinterpret_cast<double*>(smem); // 64 doubles
    cplx* buf0 = reinterpret_cast<cplx*>(smem + 512);
    cplx* buf1 = buf0 + cap;
    const int64_t b = blockIdx.x;
    const int H = a.H, W = a.W;
    const ZLayout L = make_layout(a);
    EwPtrs P = make_ptrs(a, b);
    kern_prepare<OP>(a, P);
    const int NC = 1 + L.wc;
    const int Hp = (H + 1) / 2;
    const bool weven = (W % 2) == 0;
    cplx* cur = buf0;
    cplx* oth = buf1;
    // (index arithmetic of the element loops: multiply-high instead of integer division, gpt_lds_fft.h)
    const unsigned mW = fastdiv_magic((unsigned)W), mNC = fastdiv_magic((unsigned)NC),
                   mwc = fastdiv_magic((unsigned)(L.wc > 0 ? L.wc : 1));

    if (op_has_fwd(OP)) {
        // load: line r = rows (2r, 2r+1)
        for (int w = threadIdx.x; w < Hp * W; w += blockDim.x) {
            const int r = (int)fastdiv((unsigned)w, (unsigned)W, mW), j = w - r * W;
            const double re = in_transform<OP>(a, b, 2 * r, j);
            const double im = (2 * r + 1 < H) ? in_transform<OP>(a, b, 2 * r + 1, j) : 0.0;
            cur[w] = make_double2(re, im);
        }
        __syncthreads();
        cplx* res = fft_lines_lds<false>(cur, oth, a.pw, Hp, 1, W);
        if (res != cur) {
            oth = cur;
            cur = res;
        }
        // split row pairs -> packed spectrum S[i * sS + jp] in `oth`
        for (int w = threadIdx.x; w < H * NC; w += blockDim.x) {
            const int i = (int)fastdiv((unsigned)w, (unsigned)NC, mNC), jp = w - i * NC;
            const cplx* line = cur + (i >> 1) * W;
            auto rowspec = [&](int j) -> cplx {
                const cplx z = line[j];
                const cplx zc = line[j == 0 ? 0 : W - j];
                // A = (Z + conj Zc)/2 (even row) ; B = (Z - conj Zc)/(2i) (odd row)
                if ((i & 1) == 0) return make_double2(0.5 * (z.x + zc.x), 0.5 * (z.y - zc.y));
                return make_double2(0.5 * (z.y + zc.y), -0.5 * (z.x - zc.x));
            };
            cplx v;
            if (jp == 0) {
                v.x = rowspec(0).x;
                v.y = weven ? rowspec(W / 2).x : 0.0;
            } else {
                v = rowspec(jp);
            }
            oth[i * sS + jp] = v;
        }
        __syncthreads();
        {
            cplx* t = cur;
            cur = oth;
            oth = t;
        }
        res = fft_lines_lds<false>(cur, oth, a.ph, NC, sS, 1);
        if (res != cur) {
            oth = cur;
            cur = res;
        }
    }

    // elementwise stage on S = cur
    EwAcc acc;
    acc_init(acc);
    for (int w = threadIdx.x; w < H * L.wc; w += blockDim.x) {
        const int i = (int)fastdiv((unsigned)w, (unsigned)L.wc, mwc), j = 1 + (w - i * L.wc);
        const cplx F = cur[i * sS + j];
        const cplx G = ew_interior<OP>(a, L, P, i, j, F, acc);
        if (op_has_inv(OP)) cur[i * sS + j] = G;
        acc_renorm(acc);
    }
    acc_fold(acc);
    for (int k = threadIdx.x; k <= H / 2; k += blockDim.x) {
        const int km = (H - k) % H;
        cplx Zk = cur[k * sS], Zm = cur[km * sS];
        ew_packed0<OP>(a, L, P, k, Zk, Zm, acc);
        if (op_has_inv(OP)) {
            cur[k * sS] = Zk;
            if (km != k) cur[km * sS] = Zm;
        }
    }
    if (OP != OP_RFFT2 && OP != OP_INV_RFFT2) {
        double s = acc.sum, m = acc.mn;
        double kk[3] = {acc.k0, acc.k1, acc.k2};
        if (op_is_kvjp(OP)) { // kernel-fused op: the three sums behind d/d(sigma, length_scale, period)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double t = kk[i], dummy = 0.0;
                block_sum_min(t, dummy, red);
                kk[i] = t;
            }
        }
        block_sum_min(s, m, red);
        if (threadIdx.x == 0) {
            const double lp = s - 0.5 * (double)H * (double)W * 1.8378770664093454836;
            if ((OP == OP_LPDF || op_is_lpdf_vjp(OP) || OP == OP_INV_VJP) && a.out_lpdf) a.out_lpdf[b] = lp;
            if (op_is_kvjp(OP) && a.ksums) {
                a.ksums[4 * b] = lp;
                a.ksums[4 * b + 1] = kk[0];
                a.ksums[4 * b + 2] = kk[1];
                a.ksums[4 * b + 3] = kk[2];
            }
            if (a.min_eig) a.min_eig[b] = m;
        }
    }
    __syncthreads();

    if (op_has_inv(OP)) {
        cplx* res = fft_lines_lds<true>(cur, oth, a.ph, NC, sS, 1);
        if (res != cur) {
            oth = cur;
            cur = res;
        }
        // merge row pairs into complex lines of length W in `oth`
        for (int w = threadIdx.x; w < Hp * W; w += blockDim.x) {
            const int r = (int)fastdiv((unsigned)w, (unsigned)W, mW), j = w - r * W;
            auto rowspec = [&](int i, int jj) -> cplx {
                const bool mir = jj > W / 2;
                const int q = mir ? W - jj : jj;
                cplx v;
                if (q == 0) v = make_double2(cur[i * sS].x, 0.0);
                else if (weven && q == W / 2) v = make_double2(cur[i * sS].y, 0.0);
                else v = cur[i * sS + q];
                if (mir) v.y = -v.y;
                return v;
            };
            const cplx A = rowspec(2 * r, j);
            cplx Bv = make_double2(0.0, 0.0);
            if (2 * r + 1 < H) Bv = rowspec(2 * r + 1, j);
            oth[w] = make_double2(A.x - Bv.y, A.y + Bv.x);
        }
        __syncthreads();
        {
            cplx* t = cur;
            cur = oth;
            oth = t;
        }
        res = fft_lines_lds<true>(cur, oth, a.pw, Hp, 1, W);
        for (int w = threadIdx.x; w < Hp * W; w += blockDim.x) {
            const int r = (int)fastdiv((unsigned)w, (unsigned)W, mW), j = w - r * W;
            const cplx v = res[w];
            double* o = a.out + b * (int64_t)H * W;
            o[(int64_t)(2 * r) * W + j] = out_transform<OP>(a, b, 2 * r, j, v.x);
            if (2 * r + 1 < H) o[(int64_t)(2 * r + 1) * W + j] = out_transform<OP>(a, b, 2 * r + 1, j, v.y);
        }
    }
}

template <int OP>
hipError_t launch_small_t(const LatArgs& a, int cap, int sS, int nthreads, size_t lds, hipStream_t st) {
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
    hipLaunchKernelGGL(small_lattice_kernel<OP>, dim3((unsigned)a.batch), dim3(nthreads), lds, st, a, cap, sS);
    return hipGetLastError();
}

} // namespace

bool gpt_small_fits(int H, int W, int* cap_out, int* stride_out) {
    const int NC = 1 + (W - 1) / 2;
    const int sS = NC | 1; // odd stride: column walks hit distinct LDS banks
    const int Hp = (H + 1) / 2;
    int64_t cap = (int64_t)Hp * W;
    if ((int64_t)H * sS > cap) cap = (int64_t)H * sS;
    if (cap_out) *cap_out = (int)cap;
    if (stride_out) *stride_out = sS;
    return 512 + 2 * cap * (int64_t)sizeof(cplx) <= 160 * 1024;
}

hipError_t gpt_launch_small(int op, const LatArgs& a, hipStream_t st) {
    int cap, sS;
    if (!gpt_small_fits(a.H, a.W, &cap, &sS)) return hipErrorInvalidValue;
    if (a.batch <= 0) return hipSuccess;
    const size_t lds = 512 + 2 * (size_t)cap * sizeof(cplx);
    int64_t work = (int64_t)a.H * a.W / 4;
    int nthreads = (int)((work + 63) / 64 * 64);
    if (nthreads < 64) nthreads = 64;
    if (nthreads > 1024) nthreads = 1024;
    // Many lattices per launch: throughput, not latency -- four-wave workgroups let four lattices share a CU
    // (measured at 35 x 60 with 4096 chains: 369 us at 576 threads, 166 us at 256; with 64 chains 24.5 against 26.9)
    if (a.batch >= 512 && nthreads > 256) nthreads = 256;
    static const int forced = [] { // GPT_SMALL_THREADS=<n>: tuning experiments
        const char* e = getenv("GPT_SMALL_THREADS");
        return e ? atoi(e) : 0;
    }();
    if (forced >= 64 && forced <= 1024) nthreads = forced / 64 * 64;
    switch (op) {
    case OP_RFFT2: return launch_small_t<OP_RFFT2>(a, cap, sS, nthreads, lds, st);
    case OP_INV_RFFT2: return launch_small_t<OP_INV_RFFT2>(a, cap, sS, nthreads, lds, st);
    case OP_GP_RFFT2: return launch_small_t<OP_GP_RFFT2>(a, cap, sS, nthreads, lds, st);
    case OP_GP_INV_RFFT2: return launch_small_t<OP_GP_INV_RFFT2>(a, cap, sS, nthreads, lds, st);
    case OP_LPDF: return launch_small_t<OP_LPDF>(a, cap, sS, nthreads, lds, st);
    case OP_LPDF_VJP: return launch_small_t<OP_LPDF_VJP>(a, cap, sS, nthreads, lds, st);
    case OP_INV_VJP: return launch_small_t<OP_INV_VJP>(a, cap, sS, nthreads, lds, st);
    case OP_FWD_VJP: return launch_small_t<OP_FWD_VJP>(a, cap, sS, nthreads, lds, st);
    case OP_LPDF_VJP_NOCOV: return launch_small_t<OP_LPDF_VJP_NOCOV>(a, cap, sS, nthreads, lds, st);
    case OP_LPDF_KVJP: return launch_small_t<OP_LPDF_KVJP>(a, cap, sS, nthreads, lds, st);
    case OP_LPDF_KVJP_EQ: return launch_small_t<OP_LPDF_KVJP_EQ>(a, cap, sS, nthreads, lds, st);
    default: return hipErrorInvalidValue;
    }
}
