// Dispatcher of the power-of-two three-pass path (kernels in gpt_pow2_impl.h / gpt_pow2_n.hip).
#include "gpt_pow2_impl.h"

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
} // namespace

// Rows up to 16384 points (one line per 1024-thread workgroup). Columns up to 8192: the packed column 0 of
// the column pass needs a whole complex column in LDS (H * 16 bytes).
bool gpt_pow2_supported(int H, int W) { return is_pow2_in_range(H) && H <= 8192 && is_pow2_in_range(W); }

// per lattice: the packed spectrum, then 4 H doubles of edge-column staging (cov[:,0], cov[:,W/2] and the
// same two columns of d/dcov)
static size_t ws_spec_elems(int H, int W) { return (size_t)H * (size_t)(W / 2 + GPT_WS_PAD); }
size_t gpt_pow2_workspace_bytes(int H, int W) {
    return ws_spec_elems(H, W) * sizeof(cplx) + 4 * (size_t)H * sizeof(double);
}

int gpt_pow2_col_blocks(int H, int W, int cus) {
    int nb = 1;
    (void)by_size(H, [&](auto l) {
        nb = decltype(l)::col_blocks(W, cus);
        return hipSuccess;
    });
    return nb;
}

hipError_t gpt_pow2_pass(int pass, int op, const LatArgs& a, cplx* ws, double* partials, int col_cus,
                         hipStream_t st) {
    if (!gpt_pow2_supported(a.H, a.W)) return hipErrorInvalidValue;
    const int wstride = a.W / 2 + GPT_WS_PAD;
    const double n = (double)a.H * (double)a.W;
    const int nblk = gpt_pow2_col_blocks(a.H, a.W, col_cus);
    const bool reduces = (op != OP_RFFT2 && op != OP_INV_RFFT2);
    const bool lp = (op == OP_LPDF || op_is_lpdf_vjp(op));
    const double add = -0.5 * n * 1.8378770664093454836;
    // edge-column staging (speed only): cov by pass A when the op has one and reads cov; d/dcov through pass C
    // for the fused log-density gradient
    double* edge = reinterpret_cast<double*>(ws + (size_t)a.batch * ws_spec_elems(a.H, a.W));
    const int edge_flags = ((op_has_fwd(op) && a.cov) ? 1 : 0) | ((op == OP_LPDF_VJP && a.out_cov) ? 2 : 0);
    if (pass == 0) {
        if (!op_has_fwd(op)) return hipSuccess;
        const int in_mode = op == OP_RFFT2 ? 0 : (op == OP_INV_VJP ? 2 : 1);
        return by_size(a.W, [&](auto l) { return decltype(l)::row_fwd(a, in_mode, ws, wstride, (edge_flags & 1) ? edge : nullptr, st); });
    }
    if (pass == 1) {
        hipError_t e = by_size(a.H, [&](auto l) { return decltype(l)::col(op, a, ws, wstride, partials, nblk, edge_flags ? edge : nullptr, edge_flags, st); });
        if (e != hipSuccess) return e;
        // second reduction stage: folded into pass C when there is one
        if (reduces && !op_has_inv(op))
            e = gpt_launch_finalize(partials, nblk, a.batch, 1, add, lp ? a.out_lpdf : nullptr, a.min_eig, st);
        return e;
    }
    if (pass == 2) {
        if (!op_has_inv(op)) return hipSuccess;
        FinArgs fin = {nullptr, 0, 0.0, nullptr, nullptr};
        if (reduces) {
            fin.partials = partials;
            fin.nblk = nblk;
            fin.add = add;
            fin.out_sum = lp ? a.out_lpdf : nullptr;
            fin.out_min = a.min_eig;
        }
        double alpha = 1.0;
        int add_loc = 0;
        if (op == OP_INV_RFFT2) alpha = 1.0 / n;
        else if (op == OP_GP_INV_RFFT2) { alpha = 1.0 / n; add_loc = 1; }
        else if (op_is_lpdf_vjp(op)) alpha = -1.0 / n;
        return by_size(a.W, [&](auto l) { return decltype(l)::row_inv(a, alpha, add_loc, ws, wstride, fin, (edge_flags & 2) ? edge : nullptr, st); });
    }
    return hipErrorInvalidValue;
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

// The three passes on caller-owned spectrum buffers (slab decomposition of one lattice over several GPUs).
//   pass 0: rows -> packed half spectrum of `a.H` (= local rows, even) rows of width a.W; spec[rows][ld]
//   pass 1: column pass of `op` on packed columns a.col0 .. a.col0 + a.ncols - 1, all a.H rows; spec[H][ld];
//           the per-workgroup partials are summed into a.out_lpdf[0] (no constant added) and a.min_eig[0]
//   pass 2: spec[rows][ld] -> a.out[rows][W] = alpha * (inverse row transform)
hipError_t gpt_pow2_slab_pass(int pass, int op, const LatArgs& a, cplx* spec, int ld, double alpha, double* partials,
                              int cus, hipStream_t st) {
    if (pass == 0)
        return by_size(a.W, [&](auto l) { return decltype(l)::row_fwd(a, a.loc ? 1 : 0, spec, ld, nullptr, st); });
    if (pass == 1) {
        const int nblk = gpt_pow2_col_blocks(a.H, 2 * a.ncols, cus); // tiles = ncols / tile width
        hipError_t e = by_size(a.H, [&](auto l) { return decltype(l)::col(op, a, spec, ld, partials, nblk, nullptr, 0, st); });
        if (e != hipSuccess) return e;
        return gpt_launch_finalize(partials, nblk, 1, 1, 0.0, a.out_lpdf, a.min_eig, st);
    }
    if (pass == 2) {
        const FinArgs fin = {nullptr, 0, 0.0, nullptr, nullptr};
        return by_size(a.W, [&](auto l) { return decltype(l)::row_inv(a, alpha, 0, spec, ld, fin, nullptr, st); });
    }
    return hipErrorInvalidValue;
}
