// The non-centred model density of the docs workloads on ONE workgroup per lattice (SURVEY 8(f) rank 3 on the path of
// BASELINE config 2, docs/trees: a 35 x 60 padded lattice with negative-binomial counts on its 25 x 50 corner):
//     f = gp_inv_rfft2(z, loc, cov)                      fft2.stan:126-130
//     lp = sum of the observation model at the data      trees.stan:43-49 / padded.stan:34 / poisson_regression_*.stan:26
//     (+ z ~ std_normal(), trees.stan:37)
//     reverse mode: d lp / d f -> d / d(z, cov), d / d theta, sum of d / d loc
// The composed route (gpt_api_obs.hip) is five launches -- inverse transform, observation kernel, its reduction, the
// transform's reverse mode, the combine -- with f and d lp / d f going through HBM in between. Here the lattice never
// leaves LDS: spectrum from z -> inverse transform -> the observation model on the rows where they lie (f replaced by
// d lp / d f in place) -> forward transform of that -> reverse mode against z and cov -> all sums. One launch, and the
// one the host-pointer entry point can let publish its own completion (gpt_ctx::done_*).
// Phases and building blocks are those of gpt_small.hip (gpt_small_phases.h, gpt_lds_fft.h, the elementwise stage of
// gpt_common.h for OP_GP_INV_RFFT2 and OP_INV_VJP, the densities of gpt_obs_dev.h): same arithmetic as the composed route.
#include <atomic>
#include <cstdlib>

#include "../../include/gptools_hip.h"
#include "gpt_common.h"
#include "gpt_host.h"
#define GPT_LDS_FFT_INLINE __forceinline__
#include "gpt_lds_fft.h"
#include "gpt_obs_dev.h"
#include "gpt_small_phases.h"

namespace {

// `a` is set up as for OP_INV_VJP (x = z, cov, out = d/dz, out_cov = d/dcov, zprior); SmallObsArgs carries the rest.
// KMODE 1 / 2 (SURVEY 8(f) rank 4 for the non-centred models: trees.stan rebuilds its spectrum from (sigma, length_scale)
// on every gradient evaluation): no `cov` in memory at all -- the Matern / squared-exponential spectral density is
// evaluated from a.kpar[b] (+ a.knugget) where an entry is needed, in both elementwise passes, and d lp / d c goes into
// the three sums behind d / d(sigma, length_scale, period) instead of into an array (a.out_cov null, o.kbar).
template <int KMODE>
__global__ void __launch_bounds__(1024) small_obs_kernel(LatArgs a, SmallObsArgs o, int cap, int sS, SmallStages ss) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* red = reinterpret_cast<double*>(smem); // 8 x 16 doubles (GPT_SMALL_HDR bytes): step 6
    const LdsLine buf0(smem + GPT_SMALL_HDR);
    const LdsLine buf1 = buf0 + cap;
    const int64_t b = blockIdx.x;
    const int H = a.H, W = a.W;
    const ZLayout L = make_layout(a);
    EwPtrs P = make_ptrs(a, b);
    if (KMODE) kern_prepare<KMODE == 1 ? OP_LPDF_KVJP : OP_LPDF_KVJP_EQ>(a, P);
    const int NC = 1 + L.wc, Hp = (H + 1) / 2;
    const bool weven = (W % 2) == 0;
    const unsigned mW = ss.mW, mNC = ss.mNC, mwc = ss.mwc;
    LdsLine cur = buf0, oth = buf1;
    auto fft_w = [&](auto inv, LdsLine src, LdsLine dst) -> LdsLine {
        return fft_lines_lds_tw<decltype(inv)::value, const cplx*>(src, dst, a.pw, a.pw.tw, Hp, 1, W, ss.w, ss.mLw);
    };
    auto fft_h = [&](auto inv, LdsLine src, LdsLine dst) -> LdsLine {
        return fft_lines_lds_tw<decltype(inv)::value, const cplx*>(src, dst, a.ph, a.ph.tw, NC, sS, 1, ss.h, ss.mLh);
    };
    auto settle = [&](LdsLine res) { // `res` is where a transform left its result: make it `cur`
        if (res != cur) {
            oth = cur;
            cur = res;
        }
    };
    auto swap_bufs = [&] {
        const LdsLine t = cur;
        cur = oth;
        oth = t;
    };
    // cov of this thread's first entries: used by BOTH elementwise passes
    SmallCovPre pre = {1.0, 1.0, 1.0, 1.0};
    if (!KMODE) pre = small_cov_prefetch(a.cov + b * a.cov_bs, L, H, W, mwc, weven);
    // per-lattice constants of the observation family (a logarithm, lgamma and digamma of theta): under the transforms
    const double th = o.theta ? o.theta[b] : 0.0;
    double kf[3];
    obs_consts(o.family, th, kf);

    // 1. packed spectrum of f - loc: pack(z) * scale (fft2.stan:127-128), built in `cur`
    {
        EwAcc acc1; // (its minimum is found again in step 5)
        acc_init(acc1);
        small_interior<OP_GP_INV_RFFT2, KMODE>(a, L, P, cur, sS, H, mwc, pre, acc1);
        small_packed0<OP_GP_INV_RFFT2, KMODE>(a, L, P, cur, sS, H, W, weven, pre, acc1);
    }
    __syncthreads();
    // 2. inverse transform: columns, merge of the row pairs, rows
    settle(fft_h(std::true_type{}, cur, oth));
    small_merge(cur, oth, H, W, Hp, sS, mW, weven);
    __syncthreads();
    swap_bufs();
    settle(fft_w(std::true_type{}, cur, oth));
    // 3. the observation model where the rows lie: entry w of `cur` = (n (f - loc) at row 2r, the same at row 2r + 1),
    //    replaced by (d lp / d f at row 2r, at row 2r + 1) -- the input layout of the forward row transform
    const double n = (double)H * (double)W;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0; // log density, d / d theta, sum of d / d f
    {
        const double* loc = o.loc ? o.loc + b * o.loc_bs : nullptr;
        const double* y = o.obs + b * o.obs_bs;
        double* fo = o.f_out ? o.f_out + b * (int64_t)H * W : nullptr;
        double* go = o.grad_loc ? o.grad_loc + b * (int64_t)H * W : nullptr;
        for (int w = threadIdx.x; w < Hp * W; w += blockDim.x) {
            const int r = (int)fastdiv((unsigned)w, (unsigned)W, mW), j = w - r * W;
            const cplx v = cur[w];
            double g[2] = {0.0, 0.0};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * r + h;
                if (i >= H) break;
                const unsigned at = (unsigned)i * (unsigned)W + (unsigned)j;
                const double f = (h ? v.y : v.x) / n + (loc ? loc[at] : 0.0); // (out_transform<OP_GP_INV_RFFT2>)
                if (fo) fo[at] = f;
                if (i < o.oh && j < o.ow) {
                    const double yv = y[(unsigned)i * (unsigned)o.ow + (unsigned)j];
                    // masked: NaN for every family; negative counts (trees.stan:45)
                    if (yv == yv && (o.family == GPT_OBS_NORMAL || yv >= 0.0)) {
                        const ObsTerm t = obs_term_rt(o.family, yv, f, th, kf);
                        g[h] = t.df;
                        s0 += t.lp;
                        s1 += t.dth;
                        s2 += t.df;
                    }
                }
                if (go) go[at] = g[h];
            }
            cur[w] = make_double2(g[0], g[1]);
        }
    }
    __syncthreads();
    // 4. forward transform of d lp / d f: rows, separation of the row pairs, columns
    settle(fft_w(std::false_type{}, cur, oth));
    small_split(cur, oth, H, W, NC, sS, mNC, weven);
    __syncthreads();
    swap_bufs();
    settle(fft_h(std::false_type{}, cur, oth));
    // 5. reverse mode of the inverse transform against z and cov (writes d/dz, d/dcov; the prior's share if asked for)
    EwAcc acc;
    acc_init(acc);
    small_interior<OP_INV_VJP, KMODE>(a, L, P, cur, sS, H, mwc, pre, acc);
    small_packed0<OP_INV_VJP, KMODE>(a, L, P, cur, sS, H, W, weven, pre, acc);
    // 6. sums: wave by wave into LDS (8 x 16 slots = GPT_SMALL_HDR), the last thread adds them up
    {
        const int nw = ((int)blockDim.x + 63) >> 6, wv = (int)threadIdx.x >> 6;
        const double r0 = wave_sum(s0), r1 = wave_sum(s1), r2 = wave_sum(s2), r3 = wave_sum(acc.sum), r4 = wave_min(acc.mn);
        double rk[3] = {0.0, 0.0, 0.0};
        if (KMODE) {
            rk[0] = wave_sum(acc.k0);
            rk[1] = wave_sum(acc.k1);
            rk[2] = wave_sum(acc.k2);
        }
        if ((threadIdx.x & 63) == 0) {
            red[wv] = r0;
            red[16 + wv] = r1;
            red[32 + wv] = r2;
            red[48 + wv] = r3;
            red[64 + wv] = r4;
            if (KMODE) {
                red[80 + wv] = rk[0];
                red[96 + wv] = rk[1];
                red[112 + wv] = rk[2];
            }
        }
        __syncthreads();
        if (threadIdx.x == blockDim.x - 1) {
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0, t4 = INFINITY, tk[3] = {0.0, 0.0, 0.0};
            for (int i = 0; i < nw; ++i) {
                t0 += red[i];
                t1 += red[16 + i];
                t2 += red[32 + i];
                t3 += red[48 + i];
                t4 = fmin(t4, red[64 + i]);
                if (KMODE) {
                    tk[0] += red[80 + i];
                    tk[1] += red[96 + i];
                    tk[2] += red[112 + i];
                }
            }
            // z ~ std_normal(): the elementwise stage summed -z^2 / 2 (a.zprior), the constant goes on here
            const double prior = a.zprior ? t3 - 0.5 * n * 1.8378770664093454836 : 0.0;
            o.lp[b] = t0 + prior;
            if (o.grad_theta) o.grad_theta[b] = t1;
            if (o.grad_loc_sum) o.grad_loc_sum[b] = t2;
            if (a.min_eig) a.min_eig[b] = t4;
            if (KMODE && o.kbar) { // (as finalize_block, gpt_common.h)
                small_kernel_adjoints(a, b, tk[0], tk[1], tk[2], o.kbar);
            }
        }
    }
    if (ss.done_flag) { // (uniform; one workgroup: batch == 1 -- see gpt_small.hip)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(ss.done_flag, ss.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

} // namespace

hipError_t gpt_launch_small_obs(const LatArgs& a, const SmallObsArgs& o, hipStream_t st, unsigned long long* done_flag,
                                unsigned long long done_seq) {
    int cap, sS;
    if (!gpt_small_fits(a.H, a.W, &cap, &sS)) return hipErrorInvalidValue;
    if (a.batch <= 0) return hipSuccess;
    const size_t lds = GPT_SMALL_HDR + 2 * (size_t)cap * sizeof(cplx);
    const int nthreads = gpt_small_threads(a.H, a.W, a.batch);
    const SmallStages ss = make_small_stages(a, done_flag, done_seq);
    static std::atomic<uint64_t> attr_set{0}; // (kernel, device) attribute: as in gpt_small.hip
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev < 64 ? (uint64_t)1 << dev : 0;
    if (lds > 64 * 1024 && !(attr_set.load(std::memory_order_relaxed) & bit)) {
        for (const void* k : {reinterpret_cast<const void*>(&small_obs_kernel<0>), reinterpret_cast<const void*>(&small_obs_kernel<1>),
                              reinterpret_cast<const void*>(&small_obs_kernel<2>)}) {
            e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        attr_set.fetch_or(bit, std::memory_order_relaxed);
    }
    const dim3 grid((unsigned)a.batch), block(nthreads);
    if (!a.kpar) hipLaunchKernelGGL(small_obs_kernel<0>, grid, block, lds, st, a, o, cap, sS, ss);
    else if (a.kmatern) hipLaunchKernelGGL(small_obs_kernel<1>, grid, block, lds, st, a, o, cap, sS, ss);
    else hipLaunchKernelGGL(small_obs_kernel<2>, grid, block, lds, st, a, o, cap, sS, ss);
    return hipGetLastError();
}
