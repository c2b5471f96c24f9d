// Slab decomposition of one lattice over several GPUs: the three local passes behind the C ABI (gpt_slab_*).
#include "gpt_ctx.h"
using namespace gptd;

extern "C" {

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
static int slab_cus(gpt_ctx* c) { return device_cus(c); }

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

} // extern "C"
