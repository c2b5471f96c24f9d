// Phases of the one-workgroup kernels (gpt_small.hip, gpt_small_obs.hip) that both share: loops over the lattice in LDS,
// each ended by the caller's barrier. Everything is inlined into the calling kernel.
#pragma once
#include "gpt_common.h"
#include "gpt_lds_fft.h"

#define GPT_SMALL_HDR 1024 // bytes of LDS in front of the two line buffers (the reduction's per-wave slots)

// Stage constants of the two line plans, made on the host once per launch (StageK, gpt_lds_fft.h), and the division magics
// of the two line counts: (H + 1) / 2 row pairs, 1 + (W - 1) / 2 packed columns.
struct SmallStages {
    StageK w[GPT_MAX_STAGES], h[GPT_MAX_STAGES];
    unsigned mLw, mLh;
    unsigned mW, mNC, mwc; // magics of the element loops: W, NC = 1 + (W - 1) / 2, max((W - 1) / 2, 1)
    // one-lattice launches of the host-pointer entry points: where to publish completion (gpt_ctx::done_*), or null
    unsigned long long* done_flag;
    unsigned long long done_seq;
};
inline SmallStages make_small_stages(const LatArgs& a, unsigned long long* done_flag, unsigned long long done_seq) {
    SmallStages ss;
    ss.done_flag = a.batch == 1 ? done_flag : nullptr;
    ss.done_seq = done_seq;
    auto fill = [](const LinePlan& p, StageK* k) {
        int Ns = 1;
        for (int s = 0; s < p.ns && s < GPT_MAX_STAGES; ++s) {
            k[s] = stage_consts(p.n, p.r[s], Ns);
            Ns *= p.r[s];
        }
    };
    fill(a.pw, ss.w);
    fill(a.ph, ss.h);
    ss.mLw = fastdiv_magic_hd((unsigned)((a.H + 1) / 2));
    ss.mLh = ss.mNC = fastdiv_magic_hd((unsigned)(1 + (a.W - 1) / 2));
    ss.mW = fastdiv_magic_hd((unsigned)a.W);
    ss.mwc = fastdiv_magic_hd((unsigned)((a.W - 1) / 2 > 0 ? (a.W - 1) / 2 : 1));
    return ss;
}
// Workgroup size of a one-workgroup kernel for `batch` lattices of H x W (GPT_SMALL_THREADS=<n>: tuning experiments)
int gpt_small_threads(int H, int W, int64_t batch);

// Separate the transformed row pairs (lines of W points at `rows`: line r = rows 2r and 2r+1 as one complex line) into
// the packed spectrum S[i * sS + jp] at `S`: column 0 = DC + i * Nyquist, columns 1..(W-1)/2 the interior frequencies.
__device__ __forceinline__ void small_split(LdsLine rows, LdsLine S, int H, int W, int NC, int sS, unsigned mNC, bool weven) {
    for (int w = threadIdx.x; w < H * NC; w += blockDim.x) {
        const int i = (int)fastdiv((unsigned)w, (unsigned)NC, mNC), jp = w - i * NC;
        const LdsLine line = rows + (i >> 1) * W;
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
        S[i * sS + jp] = v;
    }
}

// The reverse: merge the rows of the packed spectrum at `S` pairwise into complex lines of W points at `rows`
// (line r = row 2r + i * row 2r+1 of the Hermitian-completed spectrum), ready for the inverse row transforms.
__device__ __forceinline__ void small_merge(LdsLine S, LdsLine rows, int H, int W, int Hp, int sS, unsigned mW, bool weven) {
    for (int w = threadIdx.x; w < Hp * W; w += blockDim.x) {
        const int r = (int)fastdiv((unsigned)w, (unsigned)W, mW), j = w - r * W;
        auto rowspec = [&](int i, int jj) -> cplx {
            const bool mir = jj > W / 2;
            const int q = mir ? W - jj : jj;
            cplx v;
            if (q == 0) v = make_double2(cplx(S[i * sS]).x, 0.0);
            else if (weven && q == W / 2) v = make_double2(cplx(S[i * sS]).y, 0.0);
            else v = S[i * sS + q];
            if (mir) v.y = -v.y;
            return v;
        };
        const cplx A = rowspec(2 * r, j);
        cplx Bv = make_double2(0.0, 0.0);
        if (2 * r + 1 < H) Bv = rowspec(2 * r + 1, j);
        rows[w] = make_double2(A.x - Bv.y, A.y + Bv.x);
    }
}

// cov of a thread's first two interior entries and of its packed-column row, requested ahead of the transforms
// (small_cov_prefetch) and handed to the two loops below.
struct SmallCovPre {
    double ci, ci1, c, cm;
};
template <int OP> __device__ __forceinline__ constexpr bool small_uses_cov() {
    return OP != OP_RFFT2 && OP != OP_INV_RFFT2 && !op_is_kvjp(OP);
}
__device__ __forceinline__ SmallCovPre small_cov_prefetch(const double* cov_b, const ZLayout& L, int H, int W, unsigned mwc,
                                                         bool weven) {
    SmallCovPre p = {1.0, 1.0, 1.0, 1.0};
    const int tail = (int)blockDim.x - 1 - (int)threadIdx.x;
    if ((int)threadIdx.x < H * L.wc) {
        const int i = (int)fastdiv(threadIdx.x, (unsigned)L.wc, mwc), j = 1 + ((int)threadIdx.x - i * L.wc);
        p.ci = cov_b[(unsigned)i * (unsigned)L.Wf + (unsigned)j];
    }
    if ((int)(threadIdx.x + blockDim.x) < H * L.wc) { // ... and of its second
        const unsigned w1 = threadIdx.x + blockDim.x;
        const int i = (int)fastdiv(w1, (unsigned)L.wc, mwc), j = 1 + ((int)w1 - i * L.wc);
        p.ci1 = cov_b[(unsigned)i * (unsigned)L.Wf + (unsigned)j];
    }
    const int k = tail >> 1, e = tail & 1;
    if (k <= H / 2 && (e == 0 || weven)) {
        const int j = e ? W / 2 : 0;
        p.c = cov_b[(unsigned)k * (unsigned)L.Wf + (unsigned)j];
        p.cm = cov_b[(unsigned)(k ? H - k : 0) * (unsigned)L.Wf + (unsigned)j];
    }
    return p;
}

// Elementwise stage on the interior columns 1..wc of the packed spectrum at `S` (in place for the ops with an inverse).
// KMODE: where the spectrum comes from -- 0: `cov` in memory; 1 / 2: the Matern / squared-exponential kernel evaluated
// here from the lattice's parameters (kern_eval on P.kp, + a.knugget; gpt_small_obs.hip), and for OP_INV_VJP the three
// sums behind d / d(sigma, length_scale, period) accumulated from each entry's d lp / d c (acc.cb, put_cbar).
template <int OP, int KMODE = 0>
__device__ __forceinline__ void small_interior(const LatArgs& a, const ZLayout& L, const EwPtrs& P, LdsLine S, int sS, int H,
                                               unsigned mwc, const SmallCovPre& pre, EwAcc& acc) {
    for (int w = threadIdx.x; w < H * L.wc; w += blockDim.x) {
        const int i = (int)fastdiv((unsigned)w, (unsigned)L.wc, mwc), j = 1 + (w - i * L.wc);
        const cplx F = S[i * sS + j];
        double c = 1.0;
        KernVal kv = {0.0, 0.0, 0.0};
        if (KMODE) {
            kv = kern_eval<KMODE == 1>(a, P.kp, i, j);
            c = kv.ck + a.knugget;
            acc.cb = 0.0;
        } else if (small_uses_cov<OP>()) {
            c = (w == (int)threadIdx.x) ? pre.ci
                                        : (w == (int)(threadIdx.x + blockDim.x)) ? pre.ci1
                                                                                 : P.cov[(unsigned)i * (unsigned)L.Wf + (unsigned)j];
        }
        const cplx G = ew_interior_c<OP>(a, L, P, i, j, F, c, acc);
        if (KMODE && OP == OP_INV_VJP) acc_kern(acc, kv, acc.cb);
        if (op_has_inv(OP)) S[i * sS + j] = G;
        acc_renorm(acc);
    }
    acc_fold(acc);
}

// Packed column 0 = edge column 0 + i * edge column W/2: TWO lanes per row pair (k, H - k), one per edge column, so
// that the two divisions and logarithms of a row run side by side, and on the LAST threads of the workgroup, whose
// waves have the fewest interior entries (the two loops then overlap instead of queueing up on the first wave).
// Lane pairs (t, t ^ 1) swap their results through DPP.
template <int OP, int KMODE = 0>
__device__ __forceinline__ void small_packed0(const LatArgs& a, const ZLayout& L, const EwPtrs& P, LdsLine S, int sS, int H,
                                              int W, bool weven, const SmallCovPre& pre, EwAcc& acc) {
    const int tail = (int)blockDim.x - 1 - (int)threadIdx.x; // work item: (row k, edge column)
    for (int t = tail; (t >> 1) <= H / 2; t += blockDim.x) {
        const int k = t >> 1, e = t & 1;
        const int km = k ? H - k : 0; // (H - k) % H
        const cplx Zk = S[k * sS], Zm = S[km * sS];
        // A = (Zk + conj Zm) / 2 (column 0) ; B = (Zk - conj Zm) / (2i) (column W/2)
        cplx V = e ? make_double2(0.5 * (Zk.y + Zm.y), -0.5 * (Zk.x - Zm.x)) : make_double2(0.5 * (Zk.x + Zm.x), 0.5 * (Zk.y - Zm.y));
        if (e == 0 || weven) {
            const int j = e ? W / 2 : 0;
            double c = 1.0, cm = 1.0;
            KernVal kv = {0.0, 0.0, 0.0};
            if (KMODE) { // (rows are mirrored in the kernel: row H - k has the value of row k)
                kv = kern_eval<KMODE == 1>(a, P.kp, k, j);
                c = cm = kv.ck + a.knugget;
                acc.cb = 0.0;
            } else if (small_uses_cov<OP>()) {
                c = (t == tail) ? pre.c : P.cov[(unsigned)k * (unsigned)L.Wf + (unsigned)j];
                cm = (t == tail) ? pre.cm : P.cov[(unsigned)km * (unsigned)L.Wf + (unsigned)j];
            }
            V = ew_edge_c<OP>(a, L, P, k, j, V, c, cm, acc);
            if (KMODE && OP == OP_INV_VJP) acc_kern(acc, kv, acc.cb);
        } else {
            V = make_double2(0.0, 0.0);
        }
        if (op_has_inv(OP)) {
            const cplx O = make_double2(dpp_f64<0xB1, 0xf>(0.0, V.x), dpp_f64<0xB1, 0xf>(0.0, V.y)); // the other lane's
            const cplx A = e ? O : V, B = e ? V : O;
            // Z[k] = A + iB ; Z[H-k] = conj(A) + i conj(B)
            if (e == 0) S[k * sS] = make_double2(A.x - B.y, A.y + B.x);
            else if (km != k) S[km * sS] = make_double2(A.x + B.y, -A.y + B.x);
        }
    }
}

// The three sums of acc_kern over a lattice (k0 = sum cbar ck, k1 / k2 = the same weighted with the first / second axis'
// logarithmic sensitivity) -> d lp / d(kernel parameters) of lattice b, in the layout of a.kpar: (sigma, l1, l2, p1, p2),
// or (sigma, l, p) for the 1-D functions (whose one axis is the kernel's second). As finalize_block, gpt_common.h.
__device__ __forceinline__ void small_kernel_adjoints(const LatArgs& a, int64_t b, double k0, double k1, double k2, double* kbar) {
    if (a.dim1) {
        const double* par = a.kpar + 3 * b;
        double* kb = kbar + 3 * b;
        kb[0] = 2.0 * k0 / par[0];   // sigma
        kb[1] = (k0 - k2) / par[1];  // length_scale
        kb[2] = (k2 - k0) / par[2];  // period
        return;
    }
    const double* par = a.kpar + 5 * b;
    double* kb = kbar + 5 * b;
    kb[0] = 2.0 * k0 / par[0];   // sigma
    kb[1] = (k0 - k1) / par[1];  // length_scale[0]
    kb[2] = (k0 - k2) / par[2];  // length_scale[1]
    kb[3] = (k1 - k0) / par[3];  // period[0]
    kb[4] = (k2 - k0) / par[4];  // period[1]
}
