// Dispatcher of the power-of-two three-pass path (kernels in gpt_pow2_impl.h / gpt_pow2_n.hip).
#include "gpt_pow2_sub.h"
#include <cstdlib>

namespace {
bool is_pow2_in_range(int n) { return n >= 256 && n <= 16384 && (n & (n - 1)) == 0; }

template <typename F> hipError_t by_size(int n, F f) {
    switch (n) {
    case 256: return f(pow2::Launch<256>());
    case 512: return f(pow2::Launch<512>());
    case 1024: return f(pow2::Launch<1024>());
    case 2048: return f(pow2::Launch<2048>());
    case 4096: return f(pow2::Launch<4096>());
    case 8192: return f(pow2::Launch<8192>());
    case 16384: return f(pow2::Launch<16384>());
    default: return hipErrorInvalidValue;
    }
}
template <typename F> hipError_t sub_by_size(int n, F f) {
    switch (n) {
    case 256: return f(pow2::LaunchSub<256>());
    case 512: return f(pow2::LaunchSub<512>());
    case 1024: return f(pow2::LaunchSub<1024>());
    case 2048: return f(pow2::LaunchSub<2048>());
    case 4096: return f(pow2::LaunchSub<4096>());
    default: return hipErrorInvalidValue;
    }
}
#ifndef GPT_SUB_DEFAULT
#define GPT_SUB_DEFAULT 2
#endif
// Radix of the step across rows: GPT_SUBLINE=0 keeps the whole-column kernels (A/B timing; both are parity-tested).
int sub_choice() {
    static const int r = [] {
        const char* e = getenv("GPT_SUBLINE");
        if (!e) return GPT_SUB_DEFAULT;
        return e[0] == '0' ? 0 : 2;
    }();
    return r;
}
} // namespace

// Sub-line variant (gpt_pow2_sub.h): whole lattices whose columns split into two sub-lines of 256..4096 points.
int gpt_pow2_sub_radix(int op, int H, int W) {
    const int r = sub_choice();
    if (r == 0 || op < 0 || op >= OP_COUNT || !gpt_pow2_supported(H, W)) return 0;
    return (H >= 512 && H <= 8192) ? 2 : 0; // sub-lines of 256..4096 points, any row length, every op
}
static int sub_col_blocks(int R, int H, int W, int cus) {
    int nb = 1;
    (void)sub_by_size(H / R, [&](auto l) {
        nb = decltype(l)::col_blocks(R, W / 2, cus);
        return hipSuccess;
    });
    return nb;
}

// Rows up to 16384 points (one line per 1024-thread workgroup). Columns up to 8192: the packed column 0 of
// the column pass needs a whole complex column in LDS (H * 16 bytes).
bool gpt_pow2_supported(int H, int W) { return is_pow2_in_range(H) && H <= 8192 && is_pow2_in_range(W); }

// per lattice: the packed spectrum, then 4 H doubles of edge-column staging (cov[:,0], cov[:,W/2] and the
// same two columns of d/dcov)
static size_t ws_spec_elems(int H, int W) { return (size_t)H * (size_t)(W / 2 + GPT_WS_PAD); }
size_t gpt_pow2_workspace_bytes(int H, int W) {
    return ws_spec_elems(H, W) * sizeof(cplx) + 4 * (size_t)H * sizeof(double);
}

static int whole_col_blocks(int H, int W, int cus) {
    int nb = 1;
    (void)by_size(H, [&](auto l) {
        nb = decltype(l)::col_blocks(W, cus);
        return hipSuccess;
    });
    return nb;
}
// upper bound over the kernels a lattice of this shape may run (sizes the partials buffer)
int gpt_pow2_col_blocks(int H, int W, int cus) {
    int nb = whole_col_blocks(H, W, cus);
    if (const int R = gpt_pow2_sub_radix(OP_LPDF_VJP, H, W)) { // (the widest grid of the variants)
        const int ns = sub_col_blocks(R, H, W, cus);
        if (ns > nb) nb = ns;
    }
    return nb;
}

hipError_t gpt_pow2_pass(int pass, int op, const LatArgs& a, cplx* ws, double* partials, int col_cus,
                         hipStream_t st) {
    if (!gpt_pow2_supported(a.H, a.W)) return hipErrorInvalidValue;
    const int wstride = a.W / 2 + GPT_WS_PAD;
    const double n = (double)a.H * (double)a.W;
    const int R = (a.ncols == 0 && a.tw_sub) ? gpt_pow2_sub_radix(op, a.H, a.W) : 0;
    const bool sub = R != 0;
    if (op_is_kvjp(op) && !sub) return hipErrorInvalidValue; // only the sub-line kernel evaluates the spectrum
    const int nblk = sub ? sub_col_blocks(R, a.H, a.W, col_cus) : whole_col_blocks(a.H, a.W, col_cus);
    const bool reduces = (op != OP_RFFT2 && op != OP_INV_RFFT2);
    const bool lp = (op == OP_LPDF || op_is_lpdf_vjp(op) || op == OP_INV_VJP); // (OP_INV_VJP: the z prior, if asked)
    const double add = -0.5 * n * 1.8378770664093454836;
    // edge-column staging (speed only): cov by pass A when the op has one and reads cov; d/dcov through pass C
    // for the fused log-density gradient
    double* edge = reinterpret_cast<double*>(ws + (size_t)a.batch * ws_spec_elems(a.H, a.W));
    const int edge_flags = ((op_has_fwd(op) && a.cov) ? 1 : 0) | ((op == OP_LPDF_VJP && a.out_cov) ? 2 : 0);
    if (pass == 0) {
        if (!op_has_fwd(op)) return hipSuccess;
        const int in_mode = op == OP_RFFT2 ? 0 : (op == OP_INV_VJP ? 2 : 1);
        return by_size(a.W, [&](auto l) { return decltype(l)::row_fwd(a, in_mode, ws, wstride, (edge_flags & 1) ? edge : nullptr, R == 2, st); });
    }
    if (pass == 1) {
        hipError_t e;
        if (sub)
            e = sub_by_size(a.H / R, [&](auto l) {
                return decltype(l)::colsub(op, a, R, ws, wstride, a.tw_sub, partials, nblk,
                                           edge_flags ? edge : nullptr, edge_flags, st);
            });
        else
            e = by_size(a.H, [&](auto l) { return decltype(l)::col(op, a, ws, wstride, partials, nblk, edge_flags ? edge : nullptr, edge_flags, st); });
        if (e != hipSuccess) return e;
        // second reduction stage: folded into pass C when there is one
        // (kernel-fused ops: the four sums per lattice are reduced and combined by pass C's workgroup 0 as well)
        if (reduces && !op_has_inv(op))
            e = gpt_launch_finalize(partials, nblk, a.batch, 1, add, lp ? a.out_lpdf : nullptr, a.min_eig, st);
        return e;
    }
    if (pass == 2) {
        if (!op_has_inv(op)) return hipSuccess;
        FinArgs fin = {nullptr, 0, 0.0, nullptr, nullptr, nullptr, nullptr};
        if (reduces) {
            fin.partials = partials;
            fin.nblk = nblk;
            fin.add = add;
            fin.out_sum = lp ? a.out_lpdf : nullptr;
            fin.out_min = a.min_eig;
            if (op_is_kvjp(op)) {
                fin.kpar = a.kpar;
                fin.kbar = a.kbar;
            }
        }
        double alpha = 1.0;
        int add_loc = 0;
        if (op == OP_INV_RFFT2) alpha = 1.0 / n;
        else if (op == OP_GP_INV_RFFT2) { alpha = 1.0 / n; add_loc = 1; }
        else if (op_is_lpdf_vjp(op)) alpha = -1.0 / n;
        return by_size(a.W, [&](auto l) { return decltype(l)::row_inv(a, alpha, add_loc, ws, wstride, fin, (edge_flags & 2) ? edge : nullptr, R == 2, st); });
    }
    return hipErrorInvalidValue;
}

// Fused rows of the non-centred model density (row_obs_kernel): between pass 1 of OP_GP_INV_RFFT2 and pass 1 of
// OP_INV_VJP on the same workspace. `a` carries the arguments of the inverse transform (loc, cov, min_eig, plans);
// `partials` are those of its column pass (summed here into min_eig), o.partials the observation model's.
bool gpt_pow2_obs_supported(int H, int W) { return gpt_pow2_supported(H, W) && W <= 8192; }
int gpt_pow2_obs_blocks(int H, int W) {
    int nb = 0;
    (void)by_size(W, [&](auto l) {
        nb = decltype(l)::row_obs_blocks(H);
        return hipSuccess;
    });
    return nb;
}
hipError_t gpt_pow2_obs_rows(const LatArgs& a, cplx* ws, double* partials, int col_cus, const ObsRowArgs& o,
                             hipStream_t st) {
    if (!gpt_pow2_obs_supported(a.H, a.W)) return hipErrorInvalidValue;
    const int wstride = a.W / 2 + GPT_WS_PAD;
    const int R = a.tw_sub ? gpt_pow2_sub_radix(OP_GP_INV_RFFT2, a.H, a.W) : 0;
    if (R != 0 && R != 2) return hipErrorInvalidValue;
    if (R != (a.tw_sub ? gpt_pow2_sub_radix(OP_INV_VJP, a.H, a.W) : 0)) return hipErrorInvalidValue;
    const int nblk = R ? sub_col_blocks(R, a.H, a.W, col_cus) : whole_col_blocks(a.H, a.W, col_cus);
    FinArgs fin = {partials, nblk, 0.0, nullptr, a.min_eig, nullptr, nullptr};
    double* edge = reinterpret_cast<double*>(ws + (size_t)a.batch * ws_spec_elems(a.H, a.W));
    const double alpha = 1.0 / ((double)a.H * (double)a.W);
    return by_size(a.W, [&](auto l) { return decltype(l)::row_obs(a, alpha, ws, wstride, fin, a.cov ? edge : nullptr, o, R == 2, st); });
}

// Columns a column-pass workgroup takes at once for lines of length H: slab widths must be multiples of it.
int gpt_pow2_col_tile(int H) {
    int tc = 0;
    (void)by_size(H, [&](auto l) {
        tc = decltype(l)::col_tile();
        return hipSuccess;
    });
    return tc;
}

// Long lines (gpt_long.hip): rows of more than 16384 points and columns of more than 8192 take the four-step
// route; GPT_LONG_MIN=<n> sends every line of at least n points (n >= 2048) that way (tests).
hipError_t gpt_long_rowstep(bool inverse, int R, cplx* spec, int64_t ld, int HS, int ncols, const cplx* twH,
                            hipStream_t st);
static int long_min() {
    static const int v = [] {
        const char* e = getenv("GPT_LONG_MIN");
        const int n = e ? atoi(e) : 0;
        return n >= 2048 ? n : 1 << 30;
    }();
    return v;
}
static bool pow2_between(int n, int lo, int hi) { return n >= lo && n <= hi && (n & (n - 1)) == 0; }
// Rows of W points: 0 = two real rows share one complex line of W points (one workgroup holds it), 2 = long rows, one
// real row per complex line of W / 2 points (rowreal_*_kernel), -1 = unsupported
int gpt_long_row_split_radix(int W) {
    if (!pow2_between(W, 256, 32768)) return -1;
    return (W > 16384 || W >= long_min()) ? 2 : 0;
}
// ... and for a column of H points: sub-lines of GPT_LONG_HS points (4096, or 2048: two workgroups per CU in the
// sub-line column kernel but a radix-16 step; measured 47.1 ms against 42.5 ms per 32768^2 evaluation on one GPU)
int gpt_long_col_split_radix(int H) {
    if (!pow2_between(H, 256, 32768)) return -1;
    if (H >= long_min()) return 8;
    if (H <= 8192) return 0;
    static const int hs = [] {
        const char* e = getenv("GPT_LONG_HS");
        const int v = e ? atoi(e) : 4096;
        return (v == 2048 || v == 4096) ? v : 4096;
    }();
    return H / hs;
}
bool gpt_slab_supported(int H, int W) { return gpt_long_row_split_radix(W) >= 0 && gpt_long_col_split_radix(H) >= 0; }
int gpt_slab_col_tile(int H) {
    const int R = gpt_long_col_split_radix(H);
    if (R <= 0) return gpt_pow2_col_tile(H);
    int tc = 0;
    (void)sub_by_size(H / R, [&](auto l) {
        tc = decltype(l)::col_tile();
        return hipSuccess;
    });
    return tc;
}
int gpt_slab_col_blocks(int H, int ncols, int cus) {
    const int R = gpt_long_col_split_radix(H);
    if (R <= 0) return whole_col_blocks(H, 2 * ncols, cus);
    int nb = 1;
    (void)sub_by_size(H / R, [&](auto l) {
        nb = decltype(l)::col_blocks(R, ncols, cus);
        return hipSuccess;
    });
    return nb;
}

// The three passes on caller-owned spectrum buffers (slab decomposition of one lattice over several GPUs).
//   pass 0: rows -> packed half spectrum of `a.H` (= local rows, even) rows of width a.W; spec[rows][ld]
//   pass 1: column pass of `op` on packed columns a.col0 .. a.col0 + a.ncols - 1, all a.H rows; spec[H][ld];
//           the per-workgroup partials are summed into a.out_lpdf[0] (no constant added) and a.min_eig[0]
//   pass 2: spec[rows][ld] -> a.out[rows][W] = alpha * (inverse row transform)
// Long lines: a.tw_sub = twiddles of the line the kernels transform (W / 2 for the row passes of long rows, H / R for
// the column pass of long columns).
hipError_t gpt_pow2_slab_pass(int pass, int op, const LatArgs& a, cplx* spec, int ld, double alpha, double* partials,
                              int cus, double* edge, hipStream_t st) {
    const int edge_flags = edge ? 3 : 0; // pass 1: cov of the two self-mirrored columns staged by the caller, their d/dcov returned
    if (pass == 0 || pass == 2) {
        const int RWd = gpt_long_row_split_radix(a.W);
        if (RWd < 0) return hipErrorInvalidValue;
        if (RWd > 0) { // one real row per complex line of W / 2 points: one sweep each way
            if (!a.tw_sub) return hipErrorInvalidValue;
            if (pass == 0) return by_size(a.W / 2, [&](auto l) { return decltype(l)::rowreal_fwd(a, spec, ld, a.tw_sub, st); });
            return by_size(a.W / 2, [&](auto l) { return decltype(l)::rowreal_inv(a, alpha, spec, ld, a.tw_sub, st); });
        }
    }
    if (pass == 0)
        return by_size(a.W, [&](auto l) { return decltype(l)::row_fwd(a, a.loc ? 1 : 0, spec, ld, nullptr, false, st); });
    if (pass == 1) {
        const int R = gpt_long_col_split_radix(a.H);
        if (R < 0) return hipErrorInvalidValue;
        const int nblk = gpt_slab_col_blocks(a.H, a.ncols, cus);
        hipError_t e;
        if (R > 0) {
            if (!a.tw_sub || !op_is_lpdf_vjp(op)) return hipErrorInvalidValue;
            const int HS = a.H / R;
            e = gpt_long_rowstep(false, R, spec, ld, HS, a.ncols, a.ph.tw, st);
            if (e != hipSuccess) return e;
            e = sub_by_size(HS, [&](auto l) { return decltype(l)::colsub(op, a, R, spec, ld, a.tw_sub, partials, nblk, edge, edge_flags, st); });
            if (e != hipSuccess) return e;
            e = gpt_long_rowstep(true, R, spec, ld, HS, a.ncols, a.ph.tw, st);
        } else {
            e = by_size(a.H, [&](auto l) { return decltype(l)::col(op, a, spec, ld, partials, nblk, edge, edge_flags, st); });
        }
        if (e != hipSuccess) return e;
        return gpt_launch_finalize(partials, nblk, 1, 1, 0.0, a.out_lpdf, a.min_eig, st);
    }
    if (pass == 2) {
        const FinArgs fin = {nullptr, 0, 0.0, nullptr, nullptr, nullptr, nullptr};
        return by_size(a.W, [&](auto l) { return decltype(l)::row_inv(a, alpha, 0, spec, ld, fin, nullptr, false, st); });
    }
    return hipErrorInvalidValue;
}

// Register-tile transform of `nlines` contiguous complex lines of n = 256 ... 4096 points (a power of two), in place,
// unnormalised; tw = w_n^m table. For the paths that compose long lines from short ones (gpt_anyn.hip).
bool gpt_pow2_line_fft_supported(int n) { return n >= 256 && n <= 4096 && (n & (n - 1)) == 0; }
hipError_t gpt_pow2_line_fft(int n, bool inverse, cplx* data, int64_t nlines, const cplx* tw, hipStream_t st) {
    if (!gpt_pow2_line_fft_supported(n)) return hipErrorInvalidValue;
    return sub_by_size(n, [&](auto l) { return decltype(l)::line_fft(inverse, data, nlines, tw, st); });
}
