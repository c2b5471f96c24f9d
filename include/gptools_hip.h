/*
 * gptools_hip.h -- C ABI of libgptools_hip.so: the Fourier-method Gaussian-process path of
 * onnela-lab/gptools as hand-written HIP kernels for AMD MI355X (gfx950).
 *
 * Every entry point replaces one function of the reference's Stan library (cited as file:line relative
 * to the reference checkout) or is the reverse-mode derivative Stan's autodiff would have produced for it.
 *
 * Conventions
 *   - all data are IEEE fp64; complex values are interleaved (re, im) pairs (std::complex<double>);
 *   - sizes are int64_t (`height`, `width`, `n`), `batch` >= 0 lattices are evaluated per call;
 *   - the 1-D functions are the 2-D ones with height == 1 and the 1-D packing layout of fft1.stan:44-51;
 *   - gpt_*        : every pointer is a DEVICE pointer, arrays are row-major (NumPy / torch order), the
 *                    call is asynchronous on the context's stream and never allocates in the steady
 *                    state (workspaces, plans and tables are cached in the context; the first call on a
 *                    new shape may allocate and synchronise). `*_bs` arguments are batch strides in
 *                    elements; 0 broadcasts one array over the batch. Output arrays are dense
 *                    ([batch][height][width] etc.).
 *   - gpt_*_host   : every pointer is a HOST pointer; the library stages H2D/D2H, synchronises, and
 *                    applies the reference's `reject` checks. `layout` selects GPT_COL_MAJOR (Stan /
 *                    Eigen matrices) or GPT_ROW_MAJOR.
 *   - return value : GPT_OK, or
 *         GPT_ERR_NOT_PD      (host entry points only) the reference would `reject` with "covariance matrix
 *                             is not positive-definite (minimum eigenvalue is <v>)" (fft1.stan:19-21,
 *                             fft2.stan:21-23); <v> is returned by gpt_last_min_eig(). Device entry
 *                             points report it through their `min_eig` output instead (min over the whole
 *                             matrix of n*c/2, per lattice; reject iff < 0) so that they stay asynchronous.
 *         GPT_ERR_SHAPE       sizes that the reference rejects or cannot represent (util.stan:274-276);
 *         GPT_ERR_UNSUPPORTED a lattice shape this build has no kernel for (message says which);
 *         GPT_ERR_HIP / GPT_ERR_ALLOC  a HIP failure; gpt_last_error() has the text.
 *     The library never aborts or exits.
 *   - threading    : a context is owned by one thread at a time; use one context per calling thread
 *                    (cmdstan chains are processes or STAN_THREADS threads). Several processes may share
 *                    a GPU.
 */
#ifndef GPTOOLS_HIP_H
#define GPTOOLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpt_ctx gpt_ctx;

enum {
    GPT_OK = 0,
    GPT_ERR_NOT_PD = 1,
    GPT_ERR_SHAPE = 2,
    GPT_ERR_UNSUPPORTED = 3,
    GPT_ERR_TIMEOUT = 4, /* gpt_batcher_lpdf_host only: the other chains did not arrive within gpt_batcher_set_timeout */
    GPT_ERR_HIP = -1,
    GPT_ERR_ALLOC = -2
};
enum { GPT_ROW_MAJOR = 0, GPT_COL_MAJOR = 1 };

/* ---- context ------------------------------------------------------------------------------------- */
/* `stream` is a hipStream_t the caller owns (NULL: HIP's default stream). */
int gpt_create(int device, void* stream, gpt_ctx** out);
int gpt_destroy(gpt_ctx* ctx);
int gpt_set_stream(gpt_ctx* ctx, void* stream);
int gpt_synchronize(gpt_ctx* ctx);
const char* gpt_last_error(const gpt_ctx* ctx);
double gpt_last_min_eig(const gpt_ctx* ctx);
const char* gpt_version(void);
/* Measurement aid (no counterpart in the reference): between gpt_profile_begin and gpt_profile_end the
 * power-of-two three-pass path brackets each pass of every launch with HIP events on the stream the pass
 * runs on. gpt_profile_end synchronises and returns the summed durations in milliseconds: pass_ms[0] row
 * transforms (pass A), [1] column pass (B), [2] unused (0), [3] inverse row transforms (C; includes the
 * second reduction stage), and the number of column-pass launches they were summed over. */
int gpt_profile_begin(gpt_ctx* ctx);
int gpt_profile_end(gpt_ctx* ctx, double* pass_ms, int64_t* launches);
/* Test aid (no counterpart in the reference; host arithmetic only): how the pinned staging ring of the *_host entry
 * points splits an n-byte copy over nt threads -- slice `id` is [*lo, *lo + *len). The slices tile [0, n) exactly. */
void gpt_host_copy_slice(size_t n, int nt, int id, size_t* lo, size_t* len);
/* Which kernel family a lattice shape runs on: 0 small (one workgroup per lattice, any H/W),
 * 1 power-of-two three-pass, 2 generic three-pass, 3 any-length (axes of up to 2^23 points by four-step and
 * Bluestein line transforms: long 1-D series, util.stan:263-265 takes any n), -1 unsupported. */
int gpt_path_for_shape(int64_t height, int64_t width);

/* ---- slab decomposition: ONE lattice over several GPUs (no counterpart in the reference, which
 * is single-process) ---------------------------------------------------------------------------------
 * Rank g owns `rows` consecutive rows of y, loc and d/dy (row slab) and `ncols` consecutive PACKED columns of
 * the half spectrum (column slab; packed column j holds frequency j, column 0 holds DC + i * Nyquist). The
 * exchange between the two layouts (an all-to-all over RCCL) belongs to the caller: gptools_amd/slab.py.
 * Device pointers, asynchronous on the context's stream, power-of-two height and width in [256, 32768] (lines of more
 * than 8192 / 16384 points take a four-step route inside the same calls and use library-owned scratch).
 *   gpt_slab_rows_fwd:      spec[rows][ld] complex <- packed real FFT along the width of (y - loc)[rows][W]
 *   gpt_slab_cols_lpdf_vjp: on spec[H][ld] (all rows of columns col0 .. col0+ncols-1), in place: column FFT,
 *                           log-density terms of fft2.stan:189-192 and F/c, inverse column FFT. cov and
 *                           out_cov (may be NULL) are whole H x (W/2+1) arrays of which only the slab's
 *                           columns (and column W/2 on the rank that owns packed column 0) are touched.
 *                           lpdf_partial[0] <- this slab's sum WITHOUT the -(n/2) log(2 pi) constant,
 *                           min_partial[0] (may be NULL) <- min of n*c/2 over the slab. col0 and ncols must be
 *                           multiples of the column tile (4 at H = 4096).
 *   gpt_slab_rows_inv:      out[rows][W] <- alpha * inverse packed real FFT of spec[rows][ld]
 *                           (alpha = -1/(H*W) gives d lpdf / d y). */
int gpt_slab_rows_fwd(gpt_ctx*, const double* y, const double* loc, int64_t rows, int64_t height, int64_t width,
                      double* spec, int64_t ld);
int gpt_slab_cols_lpdf_vjp(gpt_ctx*, double* spec, int64_t ld, const double* cov, double* out_cov, int64_t height,
                           int64_t width, int64_t col0, int64_t ncols, double* lpdf_partial, double* min_partial);
int gpt_slab_rows_inv(gpt_ctx*, const double* spec, int64_t ld, int64_t rows, int64_t height, int64_t width,
                      double alpha, double* out);
/* The same three passes on layouts that need no copy on either side of the all-to-all and no replicated cov:
 *   gpt_slab_rows_fwd_blocked / gpt_slab_rows_inv_blocked: spec is [width / 2 / block_cols][rows][block_cols] -- block p
 *       holds this rank's rows of the packed columns rank p owns (block_cols = width / 2 / ranks, a power of two): the
 *       send buffer of the first exchange, the receive buffer of the second;
 *   gpt_slab_cols_lpdf_vjp_owned: cov_cols / out_cov_cols (may be NULL) are [height][cov_ld] arrays holding ONLY this
 *       rank's lattice columns col0 .. col0 + ncols - 1. The two self-mirrored columns 0 and width / 2 belong to the rank
 *       with col0 == 0, which passes `edge` = 4 * height doubles: cov[:, 0], cov[:, width / 2] on entry (first 2 * height),
 *       d/dcov of the same two columns on return (last 2 * height); column 0 of its cov_cols / out_cov_cols is then
 *       neither read nor written. Other ranks pass NULL. */
int gpt_slab_rows_fwd_blocked(gpt_ctx*, const double* y, const double* loc, int64_t rows, int64_t height, int64_t width,
                              double* spec, int64_t block_cols);
int gpt_slab_cols_lpdf_vjp_owned(gpt_ctx*, double* spec, int64_t ld, const double* cov_cols, double* out_cov_cols,
                                 int64_t cov_ld, double* edge, int64_t height, int64_t width, int64_t col0,
                                 int64_t ncols, double* lpdf_partial, double* min_partial);
int gpt_slab_rows_inv_blocked(gpt_ctx*, const double* spec, int64_t block_cols, int64_t rows, int64_t height,
                              int64_t width, double alpha, double* out);

/* ---- cross-chain batching (no counterpart in the reference, where every chain evaluates
 * gp_rfft(2)_lpdf on its own CPU core) ---------------------------------------------------------------------
 * The chains of ONE process (cmdstan threads under STAN_THREADS, or any threads) share a batcher and own one
 * slot each. gpt_batcher_lpdf_host is gpt_gp_rfft2_lpdf_host for the calling chain's lattice (host pointers,
 * the batcher's `layout`; loc may be NULL; grad_y NULL: value only; grad_cov may be NULL), but it BLOCKS until
 * every active slot has called, and the last arrival evaluates all slots in ONE fused launch. Return value as
 * for the host entry points, per chain: GPT_ERR_NOT_PD concerns the calling chain's spectrum only
 * (gpt_batcher_min_eig has the value). A chain that stops calling (sampling finished, exception) must
 * gpt_batcher_leave its slot, or the others wait for it forever. Thread-safe by construction; one batcher per
 * lattice shape. */
typedef struct gpt_batcher gpt_batcher;
int gpt_batcher_create(int device, int nslots, int layout, int64_t height, int64_t width, int dim1,
                       gpt_batcher** out);
int gpt_batcher_destroy(gpt_batcher* b);
int gpt_batcher_lpdf_host(gpt_batcher* b, int slot, const double* y, const double* loc, const double* cov,
                          double* lpdf, double* grad_y, double* grad_cov);
int gpt_batcher_leave(gpt_batcher* b, int slot);
/* Longest time a call waits for the other active chains (seconds; 0, the default: for ever). When it expires the call
 * returns GPT_ERR_TIMEOUT, nothing has been evaluated for it, and the slot may call again -- the way out when a chain
 * died without gpt_batcher_leave. No counterpart in the reference. */
int gpt_batcher_set_timeout(gpt_batcher* b, double seconds);
double gpt_batcher_min_eig(gpt_batcher* b, int slot);
long long gpt_batcher_launches(gpt_batcher* b); /* fused launches so far */
const char* gpt_batcher_last_error(gpt_batcher* b);

/* ---- real-FFT layer: stan/gptools/util.stan:255-330 ------------------------------------------------ */
/* rfft2 (util.stan:302-304): y[batch][H][W] -> out[batch][H][W/2+1] complex.  rfft (util.stan:263-265) is
 * the height == 1 case. */
int gpt_rfft2(gpt_ctx*, const double* y, int64_t height, int64_t width, int64_t batch, double* out_complex);
/* inv_rfft2 (util.stan:315-330): z[batch][H][W/2+1] complex -> out[batch][H][W]; inv_rfft
 * (util.stan:291-293) is the height == 1 case. */
int gpt_inv_rfft2(gpt_ctx*, const double* z_complex, int64_t height, int64_t width, int64_t batch, double* out);
/* expand_rfft (util.stan:271-281): y[batch][n/2+1] complex -> out[batch][n] complex. */
int gpt_expand_rfft(gpt_ctx*, const double* y_complex, int64_t n, int64_t batch, double* out_complex);

/* ---- GP layer: stan/gptools/fft1.stan, fft2.stan ---------------------------------------------------
 * `dim1` != 0 selects the 1-D functions (height must be 1; layout of fft1.stan:44-51). */

/* gp_evaluate_rfft_scale (fft1.stan:14-30) / gp_evaluate_rfft2_scale (fft2.stan:13-40) */
int gpt_gp_evaluate_rfft2_scale(gpt_ctx*, const double* cov, int64_t cov_bs, int64_t height, int64_t width,
                                int64_t batch, int dim1, double* out, double* min_eig);
/* gp_unpack_rfft (fft1.stan:44-51) / gp_unpack_rfft2 (fft2.stan:55-74) */
int gpt_gp_unpack_rfft2(gpt_ctx*, const double* z_complex, int64_t height, int64_t width, int64_t batch,
                        int dim1, double* out);
/* gp_pack_rfft (fft1.stan:108-118) / gp_pack_rfft2 (fft2.stan:85-100) */
int gpt_gp_pack_rfft2(gpt_ctx*, const double* z, int64_t height, int64_t width, int64_t batch, int dim1,
                      double* out_complex);
/* gp_rfft (fft1.stan:64-66) / gp_rfft2 (fft2.stan:112-114): z = unpack(rfft2(y - loc) ./ scale) */
int gpt_gp_rfft2(gpt_ctx*, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs, const double* cov,
                 int64_t cov_bs, int64_t height, int64_t width, int64_t batch, int dim1, double* z_out,
                 double* min_eig);
/* gp_inv_rfft (fft1.stan:139-143) / gp_inv_rfft2 (fft2.stan:126-130) */
int gpt_gp_inv_rfft2(gpt_ctx*, const double* z, int64_t z_bs, const double* loc, int64_t loc_bs,
                     const double* cov, int64_t cov_bs, int64_t height, int64_t width, int64_t batch, int dim1,
                     double* f_out, double* min_eig);
/* gp_rfft_log_abs_det_jac (fft1.stan:78-82) / gp_rfft2_log_abs_det_jac (fft2.stan:143-173) */
int gpt_gp_rfft2_log_abs_det_jac(gpt_ctx*, const double* cov, int64_t cov_bs, int64_t height, int64_t width,
                                 int64_t batch, int dim1, double* out, double* min_eig);
/* its gradient w.r.t. cov (reverse mode of the function above; no reference code: Stan autodiff) */
int gpt_gp_rfft2_log_abs_det_jac_vjp(gpt_ctx*, const double* cov, int64_t cov_bs, int64_t height,
                                     int64_t width, int64_t batch, int dim1, double* grad_cov);
/* gp_rfft_lpdf (fft1.stan:96-101) / gp_rfft2_lpdf (fft2.stan:189-192): value only */
int gpt_gp_rfft2_lpdf(gpt_ctx*, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                      const double* cov, int64_t cov_bs, int64_t height, int64_t width, int64_t batch, int dim1,
                      double* lpdf, double* min_eig);
/* Fused value + gradients of the function above (the hot path): grad_y[batch][H][W] = d lpdf / d y
 * (d lpdf / d loc = -grad_y), grad_cov[batch][H][W/2+1] (may be NULL). */
int gpt_gp_rfft2_lpdf_vjp(gpt_ctx*, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                          const double* cov, int64_t cov_bs, int64_t height, int64_t width, int64_t batch,
                          int dim1, double* lpdf, double* grad_y, double* grad_cov, double* min_eig);
/* The function above for callers that cannot allocate. No counterpart in the reference: it exists for code under HIP
 * graph capture, tight serving loops and harnesses that forbid hipMalloc. Same arguments and results as
 * gpt_gp_rfft2_lpdf_vjp, but twiddle tables, partial sums and the packed-spectrum workspace live in a 40 MiB
 * `static __device__` array built into the library (one copy per GPU, held until the process ends), so the call makes
 * no hipMalloc / hipFree and touches none of the context's cached buffers. Limit: lattices of the one-workgroup path
 * of any batch size, three-pass lattices whose workspace fits the array (up to 2048 x 2048; larger batches are
 * processed in units that fit); anything beyond returns GPT_ERR_UNSUPPORTED with a message. Every call on a GPU
 * shares that GPU's array: calls that overlap in time on one GPU (two streams, two contexts, two processes are
 * fine -- each process has its own copy -- but two threads of one process are not) race. */
int gpt_gp_rfft2_lpdf_vjp_noalloc(gpt_ctx*, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                                  const double* cov, int64_t cov_bs, int64_t height, int64_t width, int64_t batch,
                                  int dim1, double* lpdf, double* grad_y, double* grad_cov, double* min_eig);
/* Reverse mode of gp_inv_rfft(2): fbar[batch][H][W] -> zbar[batch][H][W], covbar[batch][H][W/2+1] (may be
 * NULL); d/dloc = fbar. */
int gpt_gp_inv_rfft2_vjp(gpt_ctx*, const double* z, int64_t z_bs, const double* cov, int64_t cov_bs,
                         const double* fbar, int64_t fbar_bs, int64_t height, int64_t width, int64_t batch,
                         int dim1, double* zbar, double* covbar);
/* Reverse mode of gp_rfft(2): zbar -> ybar (d/dloc = -ybar), covbar (may be NULL). */
int gpt_gp_rfft2_vjp(gpt_ctx*, const double* y, int64_t y_bs, const double* loc, int64_t loc_bs,
                     const double* cov, int64_t cov_bs, const double* zbar, int64_t zbar_bs, int64_t height,
                     int64_t width, int64_t batch, int dim1, double* ybar, double* covbar);

/* Spectral kernels. `params` is [batch][5] = (sigma, length_scale[0], length_scale[1], period[0],
 * period[1]) for the 2-D functions and [batch][3] = (sigma, length_scale, period) for the 1-D ones.
 * gp_periodic_exp_quad_cov_rfft (fft1.stan:154-158) / _rfft2 (fft2.stan:206-213) */
int gpt_gp_periodic_exp_quad_cov_rfft2(gpt_ctx*, int64_t height, int64_t width, const double* params,
                                       int64_t batch, int dim1, double* out);
/* gp_periodic_matern_cov_rfft (fft1.stan:170-176) / _rfft2 (fft2.stan:228-247) */
int gpt_gp_periodic_matern_cov_rfft2(gpt_ctx*, double nu, int64_t height, int64_t width, const double* params,
                                     int64_t batch, int dim1, double* out);
/* Reverse mode: covbar[batch][H][W/2+1] -> params_bar[batch][5] (or [3]); nu is data (no d/dnu). */
int gpt_gp_periodic_exp_quad_cov_rfft2_vjp(gpt_ctx*, int64_t height, int64_t width, const double* params,
                                           const double* covbar, int64_t covbar_bs, int64_t batch, int dim1,
                                           double* params_bar);
int gpt_gp_periodic_matern_cov_rfft2_vjp(gpt_ctx*, double nu, int64_t height, int64_t width,
                                         const double* params, const double* covbar, int64_t covbar_bs,
                                         int64_t batch, int dim1, double* params_bar);

/* Kernel-fused log density (in the reference: gp_periodic_*_cov_rfft2 (fft2.stan:206-247) + the
 * model's nugget (docs/padding/padded.stan:21) feeding gp_rfft2_lpdf (fft2.stan:189-192), three functions and two
 * H x (W/2+1) temporaries): lpdf[b] = gp_rfft2_lpdf(y[b] | loc, kernel(params[b]) + nugget) with the reverse mode
 * grad_y[batch][H][W] and params_bar[batch][5] (or [3] for dim1; may be NULL) = d lpdf / d (sigma, length_scale,
 * period); nu and the nugget are data. `matern` 0: periodic squared exponential, else Matern with smoothness nu.
 * Lattices that fit a workgroup, power-of-two lattices (those of 256 rows on the generic path's kernels) and the generic
 * three-pass path (gpt_path_for_shape 0, 1, 2) evaluate the spectrum inside their elementwise stage and never store it
 * (exp-quad, Matern with nu = 0.5 ... 5.5);
 * the any-length path and other smoothness values compose the existing kernels on library-owned scratch with identical
 * results. */
int gpt_gp_rfft2_lpdf_kernel_vjp(gpt_ctx*, int matern, double nu, const double* y, int64_t y_bs, const double* loc,
                                 int64_t loc_bs, const double* params, double nugget, int64_t height, int64_t width,
                                 int64_t batch, int dim1, double* lpdf, double* grad_y, double* params_bar,
                                 double* min_eig);

/* Observation models on the device. In the reference the docs models score data against the
 * latent field inside Stan, after f = gp_inv_rfft2(z, loc, cov) (fft2.stan:126-130):
 *   docs/trees/trees.stan:37,43-49 (neg_binomial_2(exp(f), 1/kappa) on the unpadded corner, negative counts masked),
 *   docs/padding/padded.stan:33-34 (normal(f[:observe_first], kappa)),
 *   docs/poisson_regression/poisson_regression_fourier_non_centered.stan:25-26 (poisson_log(eta)).
 * One call evaluates, per lattice b,
 *   lp[b] = sum over the obs_h x obs_w corner, unmasked points, of family_lpdf(obs[i,j] | f[i,j], theta[b])
 *           (+ std_normal_lpdf(z[b]) when `prior` is non-zero),
 * all normalising constants included, and its reverse mode: grad_z, grad_cov (may be NULL), grad_loc (= d lp / d f
 * over the whole lattice, zero where nothing is observed; may be NULL), grad_loc_sum[b] (the derivative with respect to
 * a scalar loc as in trees.stan:32; may be NULL), grad_theta[b] (may be NULL). f_out (may be NULL) receives f.
 * `obs` is [batch][obs_h][obs_w] doubles (obs_bs elements apart, 0 = shared), counts stored as doubles; a NaN, or a
 * negative count, is a masked point. theta: GPT_OBS_NORMAL standard deviation, GPT_OBS_NEG_BINOMIAL_2_LOG precision phi
 * (trees.stan passes 1 / kappa), GPT_OBS_POISSON_LOG none (NULL). f and d/df stay in HBM between the three steps. */
enum { GPT_OBS_NORMAL = 0, GPT_OBS_POISSON_LOG = 1, GPT_OBS_NEG_BINOMIAL_2_LOG = 2 };
int gpt_gp_inv_rfft2_obs_lpdf_vjp(gpt_ctx*, int family, const double* z, int64_t z_bs, const double* loc, int64_t loc_bs,
                                  const double* cov, int64_t cov_bs, const double* obs, int64_t obs_bs, int64_t obs_h,
                                  int64_t obs_w, const double* theta, int prior, int64_t height, int64_t width,
                                  int64_t batch, int dim1, double* lp, double* f_out, double* grad_z, double* grad_loc,
                                  double* grad_loc_sum, double* grad_cov, double* grad_theta, double* min_eig);

/* The same density with the spectrum given by its kernel parameters instead of an array: SURVEY 8(f) rank 4 for the
 * non-centred models (fft2.stan:206-213 / :228-247 feeding fft2.stan:126-130; docs/trees/trees.stan:26-29 rebuilds
 * gp_periodic_matern_cov_rfft2(1.5, ...) + a nugget on every gradient evaluation). params: [batch][5] = (sigma,
 * length_scale[0..1], period[0..1]) ([batch][3] = (sigma, length_scale, period) when dim1); matern != 0: Matern with
 * smoothness nu, else squared exponential; nugget is added to every entry. params_bar: [batch][5] (or [3]) d lp / d params,
 * may be NULL. Lattices that fit one workgroup's LDS never hold the spectrum anywhere; other shapes compose the spectrum,
 * gpt_gp_inv_rfft2_obs_lpdf_vjp and the spectrum's reverse mode on library-owned scratch. */
int gpt_gp_inv_rfft2_obs_lpdf_kernel_vjp(gpt_ctx*, int family, int matern, double nu, const double* z, int64_t z_bs,
                                         const double* loc, int64_t loc_bs, const double* params, double nugget,
                                         const double* obs, int64_t obs_bs, int64_t obs_h, int64_t obs_w,
                                         const double* theta, int prior, int64_t height, int64_t width, int64_t batch,
                                         int dim1, double* lp, double* grad_z, double* grad_loc_sum, double* params_bar,
                                         double* grad_theta, double* min_eig);

/* ---- host-pointer entry points (what the Stan user header binds) ----------------
 * One lattice per call (batch == 1), `layout` as above. Matrices are height x width (cov: height x
 * (width/2+1)). */
int gpt_gp_rfft2_host(gpt_ctx*, int layout, const double* y, const double* loc, const double* cov,
                      int64_t height, int64_t width, int dim1, double* z_out);
int gpt_gp_inv_rfft2_host(gpt_ctx*, int layout, const double* z, const double* loc, const double* cov,
                          int64_t height, int64_t width, int dim1, double* f_out);
int gpt_gp_rfft2_log_abs_det_jac_host(gpt_ctx*, int layout, const double* cov, int64_t height, int64_t width,
                                      int dim1, double* out, double* grad_cov /* may be NULL */);
int gpt_gp_rfft2_lpdf_host(gpt_ctx*, int layout, const double* y, const double* loc, const double* cov,
                           int64_t height, int64_t width, int dim1, double* lpdf,
                           double* grad_y /* may be NULL: value only */, double* grad_cov /* may be NULL */);
/* gpt_gp_rfft2_lpdf_kernel_vjp on host pointers: gp_rfft2_lpdf (fft2.stan:189-192) over gp_periodic_*_cov_rfft2
 * (fft2.stan:206-247) + nugget in one call; params / params_bar: (sigma, length_scale[0..1], period[0..1]), or (sigma,
 * length_scale, period) when dim1. loc, grad_y and params_bar may be NULL. */
int gpt_gp_rfft2_lpdf_kernel_host(gpt_ctx*, int layout, int matern, double nu, const double* y, const double* loc,
                                  const double* params, double nugget, int64_t height, int64_t width, int dim1,
                                  double* lpdf, double* grad_y, double* params_bar);
int gpt_gp_inv_rfft2_vjp_host(gpt_ctx*, int layout, const double* z, const double* cov, const double* fbar,
                              int64_t height, int64_t width, int dim1, double* zbar, double* covbar);
int gpt_gp_rfft2_vjp_host(gpt_ctx*, int layout, const double* y, const double* loc, const double* cov,
                          const double* zbar, int64_t height, int64_t width, int dim1, double* ybar,
                          double* covbar);
/* Host-pointer form of gpt_gp_inv_rfft2_obs_lpdf_vjp for one lattice: what a `_lpmf`-style function of the Stan user
 * header binds for docs/trees/trees.stan:31-49. `loc` NULL: the scalar `loc_scalar` everywhere (rep_matrix, trees.stan:32).
 * `obs` is obs_h x obs_w in `layout`. Only z, cov and the data go up and only grad_z, grad_cov (may be NULL) and four
 * scalars come down: the field f and d lp / d f stay on the device. GPT_ERR_NOT_PD as for the other host entry points;
 * GPT_ERR_SHAPE for a scale / precision that is not positive and finite (Stan Math rejects those). The device entry
 * point cannot look at theta without a synchronisation: there such a value yields NaN. */
int gpt_gp_inv_rfft2_obs_lpdf_host(gpt_ctx*, int layout, int family, const double* z, const double* loc,
                                   double loc_scalar, const double* cov, const double* obs, int64_t obs_h,
                                   int64_t obs_w, double theta, int prior, int64_t height, int64_t width, int dim1,
                                   double* lp, double* grad_z, double* grad_loc_sum /* may be NULL */,
                                   double* grad_cov /* may be NULL */, double* grad_theta /* may be NULL */);
/* Host-pointer form of gpt_gp_inv_rfft2_obs_lpdf_kernel_vjp for one lattice: the whole right-hand side of
 * docs/trees/trees.stan:26-49 (spectrum from the kernel parameters, field, masked counts, white-noise prior) in one call.
 * params: (sigma, length_scale[0..1], period[0..1]), or (sigma, length_scale, period) when dim1; params_bar likewise, may
 * be NULL. Errors as gpt_gp_inv_rfft2_obs_lpdf_host. */
int gpt_gp_inv_rfft2_obs_lpdf_kernel_host(gpt_ctx*, int layout, int family, int matern, double nu, const double* z,
                                          const double* loc, double loc_scalar, const double* params, double nugget,
                                          const double* obs, int64_t obs_h, int64_t obs_w, double theta, int prior,
                                          int64_t height, int64_t width, int dim1, double* lp, double* grad_z,
                                          double* grad_loc_sum /* may be NULL */, double* params_bar /* may be NULL */,
                                          double* grad_theta /* may be NULL */);
/* Real-FFT layer and layout helpers on host pointers (values only; the reference uses them in tests and
 * generated quantities): rfft/rfft2 (util.stan:263,302), inv_rfft/inv_rfft2 (util.stan:291,315),
 * expand_rfft (util.stan:271), gp_evaluate_rfft(2)_scale, gp_unpack_rfft(2), gp_pack_rfft(2). */
int gpt_rfft2_host(gpt_ctx*, int layout, const double* y, int64_t height, int64_t width, double* out_complex);
int gpt_inv_rfft2_host(gpt_ctx*, int layout, const double* z_complex, int64_t height, int64_t width, double* out);
int gpt_expand_rfft_host(gpt_ctx*, const double* y_complex, int64_t n, double* out_complex);
int gpt_gp_evaluate_rfft2_scale_host(gpt_ctx*, int layout, const double* cov, int64_t height, int64_t width,
                                     int dim1, double* out);
int gpt_gp_unpack_rfft2_host(gpt_ctx*, int layout, const double* z_complex, int64_t height, int64_t width,
                             int dim1, double* out);
int gpt_gp_pack_rfft2_host(gpt_ctx*, int layout, const double* z, int64_t height, int64_t width, int dim1,
                           double* out_complex);
int gpt_gp_periodic_exp_quad_cov_rfft2_host(gpt_ctx*, int layout, int64_t height, int64_t width,
                                            const double* params, int dim1, double* out);
int gpt_gp_periodic_matern_cov_rfft2_host(gpt_ctx*, int layout, double nu, int64_t height, int64_t width,
                                          const double* params, int dim1, double* out);
int gpt_gp_periodic_exp_quad_cov_rfft2_vjp_host(gpt_ctx*, int layout, int64_t height, int64_t width,
                                                const double* params, const double* covbar, int dim1,
                                                double* params_bar);
int gpt_gp_periodic_matern_cov_rfft2_vjp_host(gpt_ctx*, int layout, double nu, int64_t height, int64_t width,
                                              const double* params, const double* covbar, int dim1,
                                              double* params_bar);

#ifdef __cplusplus
}
#endif
#endif /* GPTOOLS_HIP_H */
