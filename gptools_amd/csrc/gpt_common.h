// Shared device-side definitions for the Fourier-GP kernels (gfx950 / MI355X only).
//
// Layout vocabulary (all fp64):
//   lattice        H x W real field, row-major (torch / NumPy order); Stan's column-major matrices are
//                  transposed by the host-pointer entry points before they get here.
//   half spectrum  H x Wf complex, Wf = W/2 + 1: what the reference's rfft2 returns (util.stan:302-304).
//   packed spectrum (internal workspace) H x NC complex, NC = 1 + (W-1)/2:
//                  column 0 holds  A + i*B  where A is the (real-row-transform) DC column j = 0 and B the
//                  Nyquist column j = W/2 (B = 0 when W is odd); columns 1..wc hold j = 1..wc as they are.
//                  Both A and B are column transforms of real sequences, so they separate by Hermitian
//                  symmetry: A[k] = (Z[k] + conj Z[H-k])/2, B[k] = (Z[k] - conj Z[H-k])/(2i).
//                  This is exactly the set of entries fft2.stan:55-74 reads (rows 0..H/2 of the two edge
//                  columns, all rows of the interior columns).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double2 cplx;

#define GPT_MAX_STAGES 16

enum GptOp : int {
    OP_RFFT2 = 0,        // raw rfft2: real lattice -> half spectrum                       util.stan:302
    OP_INV_RFFT2 = 1,    // raw inv_rfft2: half spectrum -> real lattice                   util.stan:315
    OP_GP_RFFT2 = 2,     // z = unpack(rfft2(y-loc)/scale)                                fft2.stan:112
    OP_GP_INV_RFFT2 = 3, // f = inv_rfft2(pack(z)*scale) + loc                            fft2.stan:126
    OP_LPDF = 4,         // gp_rfft2_lpdf value only                                      fft2.stan:189
    OP_LPDF_VJP = 5,     // value + d/dy + d/dcov (fused forward/backward)
    OP_INV_VJP = 6,      // VJP of OP_GP_INV_RFFT2: fbar -> zbar, covbar
    OP_FWD_VJP = 7,      // VJP of OP_GP_RFFT2: zbar -> ybar, covbar
    OP_LPDF_VJP_NOCOV = 8, // OP_LPDF_VJP without d/dcov (caller passed grad_cov == NULL)
    OP_LPDF_KVJP = 9,    // OP_LPDF_VJP with the spectrum evaluated on the fly from kernel parameters
                         // (fft2.stan:206-247 + nugget) and d/dcov contracted into d/d(sigma, length_scale, period)
    OP_LPDF_KVJP_EQ = 10, // ... periodic squared exponential kernel (OP_LPDF_KVJP: Matern)
    OP_COUNT = 11
};
__host__ __device__ constexpr bool op_is_kvjp(int op) { return op == OP_LPDF_KVJP || op == OP_LPDF_KVJP_EQ; }

__host__ __device__ constexpr bool op_is_lpdf_vjp(int op) {
    return op == OP_LPDF_VJP || op == OP_LPDF_VJP_NOCOV || op == OP_LPDF_KVJP || op == OP_LPDF_KVJP_EQ;
}

__host__ __device__ constexpr bool op_has_fwd(int op) {
    return op == OP_RFFT2 || op == OP_GP_RFFT2 || op == OP_LPDF || op == OP_LPDF_VJP ||
           op == OP_INV_VJP || op == OP_FWD_VJP || op == OP_LPDF_VJP_NOCOV || op_is_kvjp(op);
}
__host__ __device__ constexpr bool op_has_inv(int op) {
    return op == OP_INV_RFFT2 || op == OP_GP_INV_RFFT2 || op == OP_LPDF_VJP || op == OP_FWD_VJP ||
           op == OP_LPDF_VJP_NOCOV || op_is_kvjp(op);
}

// One line-FFT plan (any length): Stockham stages with radices r[0..ns).
struct LinePlan {
    int n;
    int ns;
    int r[GPT_MAX_STAGES];
    const cplx* tw; // device table w_n^m = exp(-2 pi i m / n), m = 0..n-1
};

struct LatArgs {
    int H, W;        // lattice shape (H = 1 for the 1-D functions)
    int dim1;        // 1: 1-D packing layout of fft1.stan:44-51, 0: 2-D layout of fft2.stan:55-74
    int64_t batch;
    // real inputs (device pointers); *_bs = batch stride in elements (0 broadcasts)
    const double* x;   // y (forward ops), z (OP_GP_INV_RFFT2 / OP_INV_VJP), y (OP_FWD_VJP)
    const double* loc; // may be null (treated as 0)
    const double* cov; // H x Wf
    const double* aux; // fbar (OP_INV_VJP: transformed), zbar (OP_FWD_VJP)
    const cplx* cin;   // complex half-spectrum input (OP_INV_RFFT2)
    int64_t x_bs, loc_bs, cov_bs, aux_bs, cin_bs;
    // outputs
    double* out;      // H x W real per lattice
    double* out_cov;  // H x Wf real per lattice (may be null)
    cplx* out_c;      // H x Wf complex per lattice (OP_RFFT2)
    double* out_lpdf; // [batch]; OP_INV_VJP with zprior: std_normal_lpdf(z)
    int zprior;       // OP_INV_VJP: add the white-noise prior z ~ std_normal() (trees.stan:37): zbar -= z, its log density
    double* min_eig;  // [batch]: min over the whole H x Wf matrix of n*c/2 (may be null)
    LinePlan pw, ph;  // line plans for length W and length H
    // Column slab (one lattice split over several GPUs, gpt_slab_*): the column pass works on packed columns
    // col0 .. col0 + ncols - 1 of the lattice; the spectrum buffer holds just those, cov / out_cov keep the
    // lattice's own H x Wf layout. ncols == 0: the whole lattice.
    int col0, ncols;
    // OP_LPDF_KVJP: the spectrum is c(i, j) = kernel(i, j; kpar[b]) + knugget (fft2.stan:206-247), never stored.
    // kpar: [batch][5] = (sigma, length_scale[0], length_scale[1], period[0], period[1]); kmatern: 0 = periodic
    // squared exponential, 1 = Matern with smoothness knu (kgamma = tgamma(knu + 1) / tgamma(knu)).
    const double* kpar;
    int kmatern;
    double knu, kgamma, knugget;
    double* kbar;  // [batch][5] d/d(sigma, length_scale, period), or null: written (with out_lpdf) by the last thread of the
                   // one-workgroup kernel / the finalize step folded into pass C -- no separate reduction or combine launch
    // Twiddle table w_{H/4}^m of the sub-line column kernels (gpt_pow2_sub.h); null: whole-column kernels only.
    const cplx* tw_sub;
    // Slab decomposition, layouts that need no copy on either side of the exchange (0 = not used):
    //  * spec_cshift, spec_blk: the row kernels address packed column j of a spectrum row at
    //    (j >> spec_cshift) * spec_blk + (j & mask) instead of j -- the [rank][rows][columns per rank] layout an
    //    all-to-all sends and receives;
    //  * cov_ld: row stride of cov / d/dcov when a rank holds only its own columns ([H][cov_ld], the pointers shifted so
    //    that lattice column numbers still index them) instead of the lattice's H x (W/2+1) arrays.
    int spec_cshift, spec_blk;
    int cov_ld;
};
// offset of packed column j inside a spectrum row (see LatArgs::spec_cshift)
__device__ __forceinline__ unsigned spec_col(const LatArgs& a, int j) {
    return a.spec_cshift ? ((unsigned)j >> a.spec_cshift) * (unsigned)a.spec_blk + ((unsigned)j & ((1u << a.spec_cshift) - 1u))
                         : (unsigned)j;
}

// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
    return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cplx cmulc(cplx a, cplx b) { // a * conj(b)
    return make_double2(fma(a.x, b.x, a.y * b.y), fma(a.y, b.x, -a.x * b.y));
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cconj(cplx a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return make_double2(a.x * s, a.y * s); }

// 64-lane wavefront reductions on the DPP data path (no LDS traffic): butterflies inside a row of 16 lanes
// (quad_perm, row_half_mirror, row_mirror), then row_bcast15 / row_bcast31 carry the row totals up to lane 63, which
// is read back into every lane. Then one LDS hop across the waves of a workgroup (block_sum_min).
// `old` is what a lane keeps when the DPP source is masked out: the reduction's neutral element.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_f64(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63_to_all(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1, 0xf>(0.0, v);  // quad_perm [1, 0, 3, 2]
    v += dpp_f64<0x4E, 0xf>(0.0, v);  // quad_perm [2, 3, 0, 1]
    v += dpp_f64<0x141, 0xf>(0.0, v); // row_half_mirror
    v += dpp_f64<0x140, 0xf>(0.0, v); // row_mirror: every lane of a row holds the row's sum
    v += dpp_f64<0x142, 0xa>(0.0, v); // row_bcast15 into rows 1 and 3
    v += dpp_f64<0x143, 0xc>(0.0, v); // row_bcast31 into rows 2 and 3: lane 63 holds the total
    return lane63_to_all(v);
}
__device__ __forceinline__ double wave_min(double v) {
    v = fmin(v, dpp_f64<0xB1, 0xf>(INFINITY, v));
    v = fmin(v, dpp_f64<0x4E, 0xf>(INFINITY, v));
    v = fmin(v, dpp_f64<0x141, 0xf>(INFINITY, v));
    v = fmin(v, dpp_f64<0x140, 0xf>(INFINITY, v));
    v = fmin(v, dpp_f64<0x142, 0xa>(INFINITY, v));
    v = fmin(v, dpp_f64<0x143, 0xc>(INFINITY, v));
    return lane63_to_all(v);
}
// red must hold >= 2 * (blockDim.x / 64) doubles. Result valid in thread 0.
__device__ __forceinline__ void block_sum_min(double& s, double& m, double* red) {
    s = wave_sum(s);
    m = wave_min(m);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) {
        red[w] = s;
        red[nw + w] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ss = 0.0, mm = INFINITY;
        for (int i = 0; i < nw; ++i) {
            ss += red[i];
            mm = fmin(mm, red[nw + i]);
        }
        s = ss;
        m = mm;
    }
}

#ifndef GPT_PART
#define GPT_PART 8 // doubles per workgroup partial (gpt_host.h)
#endif
// block-level reduction of K sums + one min into partials[(b * nblk + blk) * GPT_PART + ...]
template <int K>
__device__ __forceinline__ void block_reduce_store(double (&s)[K], double mn, double* partials, int64_t b) {
    __shared__ double red[(K + 1) * 16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = wave_sum(s[k]);
    mn = wave_min(mn);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[k * 16 + w] = s[k];
        red[K * 16 + w] = mn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* p = partials + (b * gridDim.x + blockIdx.x) * GPT_PART;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double t = 0.0;
            for (int i = 0; i < nw; ++i) t += red[k * 16 + i];
            p[k] = t;
        }
        double m = INFINITY;
        for (int i = 0; i < nw; ++i) m = fmin(m, red[K * 16 + i]);
        p[5] = m;
    }
}

// Second reduction stage folded into a later kernel of the same stream: one workgroup sums the per-workgroup
// partials of lattice b (slot 0 = log-density terms, slot 5 = min of the scaled spectrum) in a fixed order.
// Every thread of the workgroup must call it; `red` is LDS scratch of 2 * (blockDim.x / 64) doubles.
#ifndef GPT_PART
#define GPT_PART 8 // doubles per workgroup partial (gpt_host.h)
#endif
// Observation model fused between the inverse and the forward row transforms (row_obs_kernel, gpt_pow2_impl.h)
struct ObsRowArgs {
    int family, oh, ow;  // GPT_OBS_*; observed corner of the lattice
    const double* obs;   // [batch][oh][ow], obs_bs elements apart (0 = shared)
    int64_t obs_bs;
    const double* theta; // [batch] or null
    double* f_out;       // [batch][H][W] or null
    double* fbar_out;    // [batch][H][W] or null
    double* partials;    // [batch][gridDim.x][GPT_PART]: log density, d/dtheta, sum of d/df, observed points
};
struct FinArgs {
    const double* partials; // null: nothing to do
    int nblk;
    double add;
    double* out_sum; // may be null
    double* out_min; // may be null
    // kernel-fused log density (OP_LPDF_KVJP): slots 1..3 of the partials hold the three sums of acc_kern; kpar =
    // [batch][5] kernel parameters, kbar = [batch][5] their gradient (null: not asked for)
    const double* kpar;
    double* kbar;
};
__device__ __forceinline__ void finalize_block(const FinArgs& f, int64_t b, double* red) {
    const double* p = f.partials + b * (int64_t)f.nblk * GPT_PART;
    double s = 0.0, mn = INFINITY;
    for (int i = threadIdx.x; i < f.nblk; i += blockDim.x) {
        s += p[(int64_t)i * GPT_PART];
        mn = fmin(mn, p[(int64_t)i * GPT_PART + 5]);
    }
    block_sum_min(s, mn, red);
    if (threadIdx.x == 0) {
        if (f.out_sum) f.out_sum[b] = s + f.add;
        if (f.out_min) f.out_min[b] = mn;
    }
    if (f.kpar && f.kbar) { // (workgroup-uniform)
        double k[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            double t = 0.0, dummy = INFINITY;
            for (int i = threadIdx.x; i < f.nblk; i += blockDim.x) t += p[(int64_t)i * GPT_PART + 1 + q];
            __syncthreads(); // thread 0 is done reading the previous round's red[]
            block_sum_min(t, dummy, red);
            k[q] = t;
        }
        if (threadIdx.x == 0) {
            const double* par = f.kpar + 5 * b;
            double* o = f.kbar + 5 * b;
            o[0] = 2.0 * k[0] / par[0];    // sigma
            o[1] = (k[0] - k[1]) / par[1]; // length_scale[0]
            o[2] = (k[0] - k[2]) / par[2]; // length_scale[1]
            o[3] = (k[1] - k[0]) / par[3]; // period[0]
            o[4] = (k[2] - k[0]) / par[4]; // period[1]
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Packing layout of the whitened coefficients z (the bit-exact contract of the reference).
// 2-D (fft2.stan:55-74): column 0 <- edge column j=0; columns 1..wc <- Re of j=1..wc; columns
//   1+wc..2wc <- Im of j=1..wc; (W even) column W-1 <- edge column j=W/2. Inside an edge column
//   (fft1.stan:44-51 applied along the height): rows 0..H/2 <- Re of k, rows Hf..H-1 <- Im of k=1..hc.
// 1-D (fft1.stan:44-51): [Re_0..Re_{n/2}, Im_1..Im_nc].
struct ZLayout {
    int H, W, Wf, Hf, wc, hc, dim1;
    __device__ __forceinline__ int re_col(int j) const { // interior j in 1..wc
        return j;
    }
    __device__ __forceinline__ int im_col(int j) const { return dim1 ? (Wf + j - 1) : (j + wc); }
    // column of z that holds edge column j (j == 0 or j == W/2)
    __device__ __forceinline__ int edge_col(int j) const { return j == 0 ? 0 : (dim1 ? W / 2 : W - 1); }
};

__device__ __forceinline__ ZLayout make_layout(const LatArgs& a) {
    ZLayout L;
    L.H = a.H;
    L.W = a.W;
    L.Wf = a.cov_ld ? a.cov_ld : a.W / 2 + 1; // (row stride of the H x Wf arrays: cov, d/dcov, complex in / out)
    L.Hf = a.H / 2 + 1;
    L.wc = (a.W - 1) / 2;
    L.hc = (a.H - 1) / 2;
    L.dim1 = a.dim1;
    return L;
}

// 1 / c for the elementwise stage. GPT_FAST_RCP: v_rcp_f64 + two Newton steps (<= 1 ulp for normal c) instead of the
// IEEE division sequence (11 dependent instructions); c = 0, inf or denormal then give NaN where IEEE gives inf / 0.
#ifndef GPT_FAST_RCP
#define GPT_FAST_RCP 0
#endif
__device__ __forceinline__ double gpt_rcp(double c) {
#if GPT_FAST_RCP
    double r = __builtin_amdgcn_rcp(c);
    r = fma(fma(-c, r, 1.0), r, r);
    r = fma(fma(-c, r, 1.0), r, r);
    return r;
#else
    return 1.0 / c;
#endif
}

// Per-thread accumulators of the elementwise stage.
struct EwAcc {
    double sum;  // log-density partial (everything except the pair entries' log c)
    double mn;   // min of n*c/2 seen
    double mant; // sum over pair entries of log(c) is kept as log(mant) + ex * ln 2: one frexp and one
    int ex;      // multiplication per entry instead of a 60-instruction fp64 log (and more accurate)
    // OP_LPDF_KVJP: sums of cbar * ck, cbar * ck * f1, cbar * ck * f2 (ck = kernel part of c, f = its logarithmic
    // length-scale sensitivities, kern_eval): everything d/d(sigma, length_scale, period) needs
    double k0, k1, k2;
    // OP_INV_VJP: d lp / d c of the entries the last elementwise call handled (put_cbar). The kernel-fused one-workgroup
    // kernel (gpt_small_obs.hip) feeds it to acc_kern instead of reading d/dcov back; everywhere else it is dead code.
    double cb;
};

__device__ __forceinline__ void acc_init(EwAcc& acc) {
    acc.sum = 0.0;
    acc.mn = INFINITY;
    acc.mant = 1.0;
    acc.ex = 0;
    acc.k0 = acc.k1 = acc.k2 = 0.0;
    acc.cb = 0.0;
}
// acc += log(c) for a pair entry, deferred.
__device__ __forceinline__ void acc_log(EwAcc& acc, double c) {
    acc.mant *= __builtin_amdgcn_frexp_mant(c);
    acc.ex += __builtin_amdgcn_frexp_exp(c);
}
// Renormalise (call at least every ~900 entries: each factor is in [0.5, 1)).
__device__ __forceinline__ void acc_renorm(EwAcc& acc) {
    const int e = __builtin_amdgcn_frexp_exp(acc.mant);
    acc.mant = __builtin_amdgcn_frexp_mant(acc.mant);
    acc.ex += e;
}
// Fold the deferred logs into the running sum (sum -= sum of log c).
__device__ __forceinline__ void acc_fold(EwAcc& acc) {
    acc.sum -= log(acc.mant) + (double)acc.ex * 0.69314718055994530942;
    acc.mant = 1.0;
    acc.ex = 0;
}

// Per-lattice constants of the kernel-fused ops (kern_prepare): everything of the spectral density that does not
// depend on the entry, computed ONCE per workgroup and made wave-uniform (scalar registers), so that an entry costs no
// parameter load and no division by a period (they used to be redone per entry: the pins that keep the register tiles
// from spilling also kept the compiler from hoisting them).
struct KernPre {
    double r1, r2; // length_scale / period per axis
    double camp;   // ck = camp * arg^-(nu + 1)  (Matern)   or   camp * exp(-2 pi^2 (s1 + s2))  (squared exponential)
    double tn;     // Matern: arg = 1 + tn (s1 + s2)
    double gco;    // f = gco * s / arg (Matern), gco * s (squared exponential)
    int m;         // Matern: nu + 1/2
    // Matern: ck = camp * (ysel * y + yoff) * arg^-m with y = arg^-1/2 -- (1, 0) for the 2-D density (exponent nu + 1,
    // fft2.stan:228-247), (0, 1) for the 1-D one (exponent nu + 1/2, fft1.stan:170-176; one-workgroup kernels only)
    double ysel, yoff;
};
__device__ __forceinline__ double gpt_uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// Per-lattice base pointers (workgroup-uniform: they live in SGPRs); elements are then addressed with
// 32-bit offsets (H * W < 2^31 as in the reference, fft2.stan:15), one VGPR per address.
struct EwPtrs {
    const double* x;
    const double* cov;
    const double* aux;
    const cplx* cin;
    double* out;
    double* out_cov;
    cplx* out_c;
    // Optional staging of the two edge columns (j = 0 and W/2), contiguous per lattice: edge_cov[0..H) =
    // cov[:, 0], [H..2H) = cov[:, W/2] (filled by an earlier pass); edge_dcov the same for d/dcov (scattered
    // into out_cov by a later pass). Null: the elementwise stage reads cov / writes out_cov itself.
    const double* edge_cov;
    double* edge_dcov;
    const double* kpar; // OP_LPDF_KVJP: this lattice's kernel parameters
    KernPre kp;         // ... and what kern_prepare made of them
};

__device__ __forceinline__ EwPtrs make_ptrs(const LatArgs& a, int64_t b) {
    const int64_t hw = (int64_t)a.H * a.W, hwf = (int64_t)a.H * (a.W / 2 + 1);
    EwPtrs P;
    P.x = a.x ? a.x + b * a.x_bs : nullptr;
    P.cov = a.cov ? a.cov + b * a.cov_bs : nullptr;
    P.aux = a.aux ? a.aux + b * a.aux_bs : nullptr;
    P.cin = a.cin ? a.cin + b * a.cin_bs : nullptr;
    P.out = a.out ? a.out + b * hw : nullptr;
    P.out_cov = a.out_cov ? a.out_cov + b * hwf : nullptr;
    P.out_c = a.out_c ? a.out_c + b * hwf : nullptr;
    P.edge_cov = nullptr;
    P.edge_dcov = nullptr;
    P.kpar = a.kpar ? a.kpar + b * (a.dim1 ? 3 : 5) : nullptr;
    P.kp.r1 = P.kp.r2 = P.kp.camp = P.kp.tn = P.kp.gco = 0.0;
    P.kp.m = 0;
    P.kp.ysel = 1.0;
    P.kp.yoff = 0.0;
    return P;
}
// d lp / d c of one spectrum entry out of the reverse mode of the inverse transform: stored (a null out_cov: the caller
// has no use for the array -- the kernel-fused route, whose spectrum was never in memory) and left in acc.cb.
__device__ __forceinline__ void put_cbar(const EwPtrs& P, EwAcc& acc, unsigned at, double v) {
    if (P.out_cov) P.out_cov[at] = v;
    acc.cb += v;
}
// Kernels that run the kernel-fused ops call this once after make_ptrs (workgroup-uniform control flow).
template <int OP> __device__ __forceinline__ void kern_prepare(const LatArgs& a, EwPtrs& P) {
    if (!op_is_kvjp(OP)) return;
    constexpr double PI_ = 3.14159265358979323846;
    const double* par = P.kpar;
    // 2-D: (sigma, l1, l2, p1, p2), fft2.stan:206-247. 1-D (a.dim1, H = 1): (sigma, l, p), fft1.stan:154-176 -- the one
    // axis is the kernel's second (column) axis, r1 = 0 switches the first one off.
    const double sigma = par[0];
    const double r1 = a.dim1 ? 0.0 : par[1] / par[3], r2 = a.dim1 ? par[1] / par[2] : par[2] / par[4];
    const double amp = sigma * sigma * (double)a.H * (double)a.W * (a.dim1 ? r2 : r1 * r2);
    P.kp.r1 = gpt_uniform(r1);
    P.kp.r2 = gpt_uniform(r2);
    if (OP == OP_LPDF_KVJP) {
        // a.kgamma = tgamma(nu + ndim / 2) / tgamma(nu)
        P.kp.camp = gpt_uniform(a.dim1 ? amp * sqrt(2.0 * PI_ / a.knu) * a.kgamma : amp * 4.0 * (PI_ / (2.0 * a.knu)) * a.kgamma);
        P.kp.tn = gpt_uniform(2.0 / a.knu * PI_ * PI_);
        P.kp.gco = gpt_uniform((a.knu + (a.dim1 ? 0.5 : 1.0)) * (4.0 * PI_ * PI_ / a.knu));
        P.kp.m = __builtin_amdgcn_readfirstlane((int)(a.knu + 0.5));
        P.kp.ysel = a.dim1 ? 0.0 : 1.0;
        P.kp.yoff = a.dim1 ? 1.0 : 0.0;
    } else {
        P.kp.camp = gpt_uniform(a.dim1 ? amp * sqrt(2.0 * PI_) : amp * (2.0 * PI_));
        P.kp.tn = 0.0;
        P.kp.gco = 4.0 * PI_ * PI_;
        P.kp.m = 0;
    }
}

// ---------------------------------------------------------------------------------------------------
// Spectral density at (i, j) of the half spectrum WITHOUT the nugget (fft2.stan:206-213 exp-quad, :228-247
// Matern; rows mirrored, k_i = min(i, H - i)) and its sensitivities: d ck / d l1 = ck (1 - f1) / l1,
// d ck / d p1 = ck (f1 - 1) / p1 (same with f2 for the second axis), d ck / d sigma = 2 ck / sigma.
struct KernVal {
    double ck, f1, f2;
};
template <bool MATERN>
__device__ __forceinline__ KernVal kern_eval(const LatArgs& a, const KernPre& kp, int i, int j) {
    constexpr double PI_ = 3.14159265358979323846;
    const int ki = (i <= a.H / 2) ? i : a.H - i;
    const double a1 = (double)ki * kp.r1, a2 = (double)j * kp.r2;
    const double s1 = a1 * a1, s2 = a2 * a2;
    KernVal v;
    if (!MATERN) {
        v.ck = kp.camp * exp(-2.0 * PI_ * PI_ * (s1 + s2));
        v.f1 = kp.gco * s1;
        v.f2 = kp.gco * s2;
        return v;
    }
    const double arg = fma(kp.tn, s1 + s2, 1.0); // >= 1
    // arg^-(nu + 1) for half-integer smoothness nu = m - 1/2, m = 1..6 (the reference's Python mirror allows 1.5 and
    // 2.5 only, kernels.py:306) is an odd power of y = arg^-1/2: one v_rsq_f64 with a third-order correction and a few
    // products -- no division and no square root (the column pass of this op is bound by its arithmetic). Selects with
    // trivial arms only (a nested choice of products became 32 branchy copies and 400 spilled registers). Other
    // smoothness values never get here (gpt_api.hip composes them).
    double y = __builtin_amdgcn_rsq(arg);
    {
        const double e = fma(-(arg * y), y, 1.0);
        y = fma(y * e, fma(0.375, e, 0.5), y);
    }
    const double inv = y * y;
    const double i2 = inv * inv, i4 = i2 * i2;
    const double t1 = (kp.m & 1) ? inv : 1.0, t2 = (kp.m & 2) ? i2 : 1.0, t4 = (kp.m & 4) ? i4 : 1.0; // plain selects
    v.ck = kp.camp * (fma(kp.ysel, y, kp.yoff) * (t1 * t2 * t4)); // (2-D: exactly y)
    const double g = kp.gco * inv;
    v.f1 = g * s1;
    v.f2 = g * s2;
    return v;
}
// 1 / c for the kernel-fused ops (their column pass is bound by arithmetic): v_rcp_f64 + two Newton steps (<= 1 ulp)
// for normal c, the bare hardware reciprocal for zero, denormal or non-finite c (the reference's c = 0 quirk, SURVEY
// 8(a) quirk 1, keeps its inf / NaN downstream). 5 + 3 instructions instead of the 12 of the IEEE division.
__device__ __forceinline__ double gpt_rcp_guarded(double c) {
    const double r0 = __builtin_amdgcn_rcp(c); // (0 -> inf, inf -> 0, NaN -> NaN as they are)
    double r = fma(fma(-c, r0, 1.0), r0, r0);
    r = fma(fma(-c, r, 1.0), r, r);
    return __builtin_amdgcn_class(c, 0x108 /* +-normal */) ? r : r0;
}
// cbar = d lpdf / d c at one entry -> the three sums
__device__ __forceinline__ void acc_kern(EwAcc& acc, const KernVal& kv, double cbar) {
    const double w = cbar * kv.ck;
    acc.k0 += w;
    acc.k1 = fma(w, kv.f1, acc.k1);
    acc.k2 = fma(w, kv.f2, acc.k2);
}

// ---------------------------------------------------------------------------------------------------
// Elementwise stage, INTERIOR entry (i, j), 1 <= j <= wc: a "complex pair" whose Re and Im are both
// whitened with s^2 = n c / 2 (fft2.stan:13-40, 67-68). F is the spectrum value on entry (ops with a
// forward transform); the return value replaces it (ops with an inverse transform).
// `ic` = i * Wf + j, the offset into H x Wf arrays (callers that walk rows at a fixed stride have it for an addition)
template <int OP>
__device__ __forceinline__ cplx ew_interior_ic(const LatArgs& a, const ZLayout& L, const EwPtrs& P, int i, int j,
                                              unsigned ic, cplx F, double c, EwAcc& acc) {
    const double n = (double)a.H * (double)a.W;
    const unsigned iz = (unsigned)i * (unsigned)a.W;                // row offset into H x W arrays
    cplx ret = F;
#ifdef GPT_TRAFFIC_ONLY
#if GPT_TRAFFIC_ONLY
    if (OP == OP_LPDF_VJP) P.out_cov[ic] = c;
    return make_double2(F.x + c, F.y);
#endif
#endif
    if (OP == OP_RFFT2) {
        P.out_c[ic] = F;
        return ret;
    }
    if (OP == OP_INV_RFFT2) {
        return P.cin[ic];
    }
    KernVal kv = {0.0, 0.0, 0.0};
    if (op_is_kvjp(OP)) { // `c` is not an input: the kernel is evaluated here
        kv = kern_eval<OP == OP_LPDF_KVJP>(a, P.kp, i, j);
        c = kv.ck + a.knugget;
    }
    acc.mn = fmin(acc.mn, 0.5 * n * c);
    if (OP == OP_GP_RFFT2) {
        const double inv_s = 1.0 / sqrt(0.5 * n * c);
        P.out[iz + L.re_col(j)] = F.x * inv_s;
        P.out[iz + L.im_col(j)] = F.y * inv_s;
    } else if (OP == OP_GP_INV_RFFT2) {
        const double s = sqrt(0.5 * n * c);
        ret = make_double2(P.x[iz + L.re_col(j)] * s, P.x[iz + L.im_col(j)] * s);
    } else if (OP == OP_LPDF || op_is_lpdf_vjp(OP)) {
        const double f2 = fma(F.x, F.x, F.y * F.y);
        const double inv_c = op_is_kvjp(OP) ? gpt_rcp_guarded(c) : gpt_rcp(c);
        const double q = f2 * inv_c * (1.0 / n); // |F|^2 / (n c); 1/n is uniform and hoisted: one division per
                                                 // element instead of two
        acc.sum -= q;
        acc_log(acc, c);
        if (op_is_lpdf_vjp(OP)) {
            if (OP == OP_LPDF_VJP) P.out_cov[ic] = (q - 1.0) * inv_c;
            if (op_is_kvjp(OP)) acc_kern(acc, kv, (q - 1.0) * inv_c);
            ret = cscale(F, inv_c);
        }
    } else if (OP == OP_INV_VJP) {
        // F = R = rfft2(fbar); P = pack(z)
        const double s = sqrt(0.5 * n * c);
        const double pr = P.x[iz + L.re_col(j)], pi = P.x[iz + L.im_col(j)];
        const double k2 = 2.0 / n;
        const double pz = a.zprior ? 1.0 : 0.0;
        P.out[iz + L.re_col(j)] = fma(-pz, pr, k2 * s * F.x);
        P.out[iz + L.im_col(j)] = fma(-pz, pi, k2 * s * F.y);
        acc.sum -= pz * 0.5 * fma(pr, pr, pi * pi);
        const double sbar = k2 * fma(pr, F.x, pi * F.y);
        put_cbar(P, acc, ic, sbar * s / (2.0 * c));
    } else if (OP == OP_FWD_VJP) {
        // F = rfft2(y - loc); Q = pack(zbar); returns Dh = Q / (2 s)
        const double s = sqrt(0.5 * n * c);
        const double qr = P.aux[iz + L.re_col(j)], qi = P.aux[iz + L.im_col(j)];
        P.out_cov[ic] = -fma(qr, F.x, qi * F.y) / (2.0 * c * s);
        ret = make_double2(0.5 * qr / s, 0.5 * qi / s);
    }
    return ret;
}

template <int OP>
__device__ __forceinline__ cplx ew_interior_c(const LatArgs& a, const ZLayout& L, const EwPtrs& P, int i, int j,
                                             cplx F, double c, EwAcc& acc) {
    return ew_interior_ic<OP>(a, L, P, i, j, (unsigned)i * (unsigned)L.Wf + (unsigned)j, F, c, acc);
}

// Same, loading c itself (ops without cov never touch the pointer).
template <int OP>
__device__ __forceinline__ cplx ew_interior(const LatArgs& a, const ZLayout& L, const EwPtrs& P, int i, int j,
                                           cplx F, EwAcc& acc) {
    double c = 1.0;
    if (OP != OP_RFFT2 && OP != OP_INV_RFFT2 && !op_is_kvjp(OP)) c = P.cov[(unsigned)i * (unsigned)L.Wf + (unsigned)j];
    return ew_interior_c<OP>(a, L, P, i, j, F, c, acc);
}

// Elementwise stage, EDGE column j (j == 0 or, W even, j == W/2), row k in 0..H/2. FA is the value of
// that column's own transform at row k (already separated from the packed column). Rows k = 0 and
// (H even) k = H/2 are "real points" (only Re is whitened, s^2 = n c); rows 1..hc are complex pairs whose
// mirror row H-k is never read by the reference in the forward direction (fft2.stan:65,71), but whose
// scale IS used, row by row, in the inverse direction (fft2.stan:127-128).
// Returns the replacement value at row k; the caller mirrors it (conj) to row H-k.
// `c`, `cm`: cov at (k, j) and at the mirror row (ignored by the ops without cov).
template <int OP>
__device__ __forceinline__ cplx ew_edge_c(const LatArgs& a, const ZLayout& L, const EwPtrs& P, int k, int j, cplx FA,
                                         double c, double cm, EwAcc& acc) {
    const double n = (double)a.H * (double)a.W;
    const bool real_pt = (k == 0) || (2 * k == a.H);
    const int km = real_pt ? k : a.H - k; // mirror row
    const unsigned ck = (unsigned)k * (unsigned)L.Wf + (unsigned)j, cm_ = (unsigned)km * (unsigned)L.Wf + (unsigned)j;
    cplx ret = FA;
    if (OP == OP_RFFT2) {
        if (real_pt) {
            P.out_c[ck] = make_double2(FA.x, 0.0);
        } else {
            P.out_c[ck] = FA;
            P.out_c[cm_] = cconj(FA);
        }
        return ret;
    }
    if (OP == OP_INV_RFFT2) {
        // the real part of inv_fft2 keeps the Hermitian part of the column only (util.stan:329)
        const cplx u = P.cin[ck];
        if (real_pt) return make_double2(u.x, 0.0);
        const cplx v = P.cin[cm_];
        return make_double2(0.5 * (u.x + v.x), 0.5 * (u.y - v.y));
    }
    KernVal kv = {0.0, 0.0, 0.0};
    if (op_is_kvjp(OP)) { // rows are mirrored in the kernel: the mirror row has the same value
        kv = kern_eval<OP == OP_LPDF_KVJP>(a, P.kp, k, j);
        c = cm = kv.ck + a.knugget;
    }
    if (real_pt) cm = c;
    acc.mn = fmin(acc.mn, 0.5 * n * fmin(c, cm));
    const int jz = L.edge_col(j);
    // position of Re/Im of row k inside the packed z column (fft1.stan:44-51 along the height);
    // for the 1-D functions H == 1 and only k == 0 exists.
    const unsigned zr = (unsigned)k * (unsigned)a.W + (unsigned)jz;
    const unsigned zi = (unsigned)(L.Hf + k - 1) * (unsigned)a.W + (unsigned)jz;
    if (OP == OP_GP_RFFT2) {
        if (real_pt) {
            P.out[zr] = FA.x / sqrt(n * c);
        } else {
            const double inv_s = 1.0 / sqrt(0.5 * n * c);
            P.out[zr] = FA.x * inv_s;
            P.out[zi] = FA.y * inv_s;
        }
    } else if (OP == OP_GP_INV_RFFT2) {
        if (real_pt) {
            ret = make_double2(P.x[zr] * sqrt(n * c), 0.0);
        } else {
            const double se = 0.5 * (sqrt(0.5 * n * c) + sqrt(0.5 * n * cm));
            ret = make_double2(P.x[zr] * se, P.x[zi] * se);
        }
    } else if (OP == OP_LPDF || op_is_lpdf_vjp(OP)) {
        // One straight line for real points (weight 1/2, only Re counts) and complex pairs (weight 1): a wave that holds
        // both kinds -- every wave of the one-workgroup path does -- used to run the division and the logarithm twice,
        // once per arm. Same operations in the same order as the two arms had, so the results are bit for bit the same.
        const double inv_c = 1.0 / c;
        const double f2 = real_pt ? FA.x * FA.x : fma(FA.x, FA.x, FA.y * FA.y);
        const double q = f2 * inv_c / n;
        const double wgt = real_pt ? 0.5 : 1.0;
        acc.sum -= wgt * (q + log(c));
        if (op_is_lpdf_vjp(OP)) {
            const double cbar = wgt * (q - 1.0) * inv_c;
            if (OP == OP_LPDF_VJP) {
                if (P.edge_dcov) {
                    P.edge_dcov[(j ? a.H : 0) + k] = cbar;
                    if (!real_pt) P.edge_dcov[(j ? a.H : 0) + km] = 0.0;
                } else {
                    P.out_cov[ck] = cbar;
                    if (!real_pt) P.out_cov[cm_] = 0.0;
                }
            }
            if (op_is_kvjp(OP)) acc_kern(acc, kv, cbar);
            ret = make_double2(FA.x * inv_c, real_pt ? 0.0 : FA.y * inv_c);
        }
    } else if (OP == OP_INV_VJP) {
        const double k1 = 1.0 / n;
        const double pz = a.zprior ? 1.0 : 0.0;
        if (real_pt) {
            const double s = sqrt(n * c);
            const double pr = P.x[zr];
            P.out[zr] = fma(-pz, pr, k1 * s * FA.x);
            acc.sum -= pz * 0.5 * pr * pr;
            const double sbar = k1 * pr * FA.x;
            put_cbar(P, acc, ck, sbar * s / (2.0 * c));
        } else {
            const double s = sqrt(0.5 * n * c), sm = sqrt(0.5 * n * cm);
            const double pr = P.x[zr], pi = P.x[zi];
            P.out[zr] = fma(-pz, pr, k1 * (s + sm) * FA.x);
            P.out[zi] = fma(-pz, pi, k1 * (s + sm) * FA.y);
            acc.sum -= pz * 0.5 * fma(pr, pr, pi * pi);
            const double sbar = k1 * fma(pr, FA.x, pi * FA.y);
            put_cbar(P, acc, ck, sbar * s / (2.0 * c));
            put_cbar(P, acc, cm_, sbar * sm / (2.0 * cm));
        }
    } else if (OP == OP_FWD_VJP) {
        if (real_pt) {
            const double s = sqrt(n * c);
            const double qr = P.aux[zr];
            P.out_cov[ck] = -(qr * FA.x) / (2.0 * c * s);
            ret = make_double2(qr / s, 0.0);
        } else {
            const double s = sqrt(0.5 * n * c);
            const double qr = P.aux[zr], qi = P.aux[zi];
            P.out_cov[ck] = -fma(qr, FA.x, qi * FA.y) / (2.0 * c * s);
            P.out_cov[cm_] = 0.0;
            ret = make_double2(0.5 * qr / s, 0.5 * qi / s);
        }
    }
    return ret;
}

// Same, loading cov itself.
template <int OP>
__device__ __forceinline__ cplx ew_edge(const LatArgs& a, const ZLayout& L, const EwPtrs& P, int k, int j, cplx FA,
                                       EwAcc& acc) {
    double c = 1.0, cm = 1.0;
    if (OP != OP_RFFT2 && OP != OP_INV_RFFT2 && !op_is_kvjp(OP)) {
        const bool real_pt = (k == 0) || (2 * k == a.H);
        c = P.cov[(unsigned)k * (unsigned)L.Wf + (unsigned)j];
        cm = real_pt ? c : P.cov[(unsigned)(a.H - k) * (unsigned)L.Wf + (unsigned)j];
    }
    return ew_edge_c<OP>(a, L, P, k, j, FA, c, cm, acc);
}

// cov values the packed column 0 needs at row k: edge column 0 (c0, c0m = mirror row) and W/2 (ch, chm).
struct EdgeCov {
    double c0, c0m, ch, chm;
};
template <int OP>
__device__ __forceinline__ EdgeCov load_edge_cov(const LatArgs& a, const ZLayout& L, const EwPtrs& P, int k) {
    EdgeCov e = {1.0, 1.0, 1.0, 1.0};
    if (OP != OP_RFFT2 && OP != OP_INV_RFFT2 && !op_is_kvjp(OP)) {
        const bool real_pt = (k == 0) || (2 * k == a.H);
        if (P.edge_cov) {
            const int km = real_pt ? k : a.H - k;
            e.c0 = P.edge_cov[k];
            e.c0m = P.edge_cov[km];
            if ((a.W % 2) == 0) {
                e.ch = P.edge_cov[a.H + k];
                e.chm = P.edge_cov[a.H + km];
            }
            return e;
        }
        const unsigned rk = (unsigned)k * (unsigned)L.Wf, rm = (unsigned)(real_pt ? k : a.H - k) * (unsigned)L.Wf;
        e.c0 = P.cov[rk];
        e.c0m = P.cov[rm];
        if ((a.W % 2) == 0) {
            e.ch = P.cov[rk + (unsigned)(a.W / 2)];
            e.chm = P.cov[rm + (unsigned)(a.W / 2)];
        }
    }
    return e;
}

// Elementwise stage on the packed column 0 at row k in 0..H/2: Zk = Z[k], Zm = Z[(H-k)%H] on entry
// (forward ops) and on exit (inverse ops). Splits into the two edge columns, applies ew_edge, recombines.
template <int OP>
__device__ __forceinline__ void ew_packed0_c(const LatArgs& a, const ZLayout& L, const EwPtrs& P, int k, cplx& Zk,
                                             cplx& Zm, const EdgeCov& ec, EwAcc& acc) {
    const bool weven = (a.W % 2) == 0;
    // A = (Zk + conj Zm)/2 ; B = (Zk - conj Zm)/(2i)
    cplx A = make_double2(0.5 * (Zk.x + Zm.x), 0.5 * (Zk.y - Zm.y));
    cplx B = make_double2(0.5 * (Zk.y + Zm.y), -0.5 * (Zk.x - Zm.x));
    A = ew_edge_c<OP>(a, L, P, k, 0, A, ec.c0, ec.c0m, acc);
    if (weven) B = ew_edge_c<OP>(a, L, P, k, a.W / 2, B, ec.ch, ec.chm, acc);
    else B = make_double2(0.0, 0.0);
    if (op_has_inv(OP)) {
        // Z[k] = A + iB ; Z[H-k] = conj(A) + i conj(B)
        Zk = make_double2(A.x - B.y, A.y + B.x);
        Zm = make_double2(A.x + B.y, -A.y + B.x);
    }
}

template <int OP>
__device__ __forceinline__ void ew_packed0(const LatArgs& a, const ZLayout& L, const EwPtrs& P, int k, cplx& Zk,
                                           cplx& Zm, EwAcc& acc) {
    ew_packed0_c<OP>(a, L, P, k, Zk, Zm, load_edge_cov<OP>(a, L, P, k), acc);
}

// Output transform of the final (real) stage.
//   OP_INV_RFFT2:    out = v / n
//   OP_GP_INV_RFFT2: out = v / n + loc
//   OP_LPDF_VJP:     out = -v / n            (d lpdf / d y)
//   OP_FWD_VJP:      out = v                 (the 1/2 weights already carry the n)
template <int OP>
__device__ __forceinline__ double out_transform(const LatArgs& a, int64_t b, int i, int j, double v) {
    const double n = (double)a.H * (double)a.W;
    if (OP == OP_INV_RFFT2) return v / n;
    if (OP == OP_GP_INV_RFFT2) {
        const double l = a.loc ? a.loc[b * a.loc_bs + (int64_t)i * a.W + j] : 0.0;
        return v / n + l;
    }
    if (op_is_lpdf_vjp(OP)) return -v / n;
    return v;
}

// Input transform of the first (real) stage.
template <int OP>
__device__ __forceinline__ double in_transform(const LatArgs& a, int64_t b, int i, int j) {
    const int64_t off = (int64_t)i * a.W + j;
    if (OP == OP_INV_VJP) return a.aux[b * a.aux_bs + off]; // fbar
    const double v = a.x[b * a.x_bs + off];
    if (OP == OP_RFFT2) return v;
    return a.loc ? v - a.loc[b * a.loc_bs + off] : v;
}
