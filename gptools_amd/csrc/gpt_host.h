// Host-side declarations shared by the translation units of libgptools_hip.so (internal; the public C
// ABI is include/gptools_hip.h).
#pragma once
#include "gpt_common.h"

// --- small-lattice path (gpt_small.hip)
bool gpt_small_fits(int H, int W, int* cap_out, int* stride_out);
// done_flag (device view of host-mapped memory) != null and a.batch == 1: the kernel stores done_seq there after its last result
hipError_t gpt_launch_small(int op, const LatArgs& a, hipStream_t st, unsigned long long* done_flag = nullptr,
                            unsigned long long done_seq = 0);

// --- large power-of-two path (gpt_pow2.hip)
// Workspace: packed spectrum, (H * (W/2 + GPT_WS_PAD)) complex per lattice processed concurrently.
#define GPT_WS_PAD 8
bool gpt_pow2_supported(int H, int W);
size_t gpt_pow2_workspace_bytes(int H, int W);
// Sub-line variant of the three passes (gpt_pow2_sub.h): radix R of the step across rows (2 or 4), 0 = the
// whole-column kernels. Needs LatArgs::tw_sub = twiddles of length H / R.
int gpt_pow2_sub_radix(int op, int H, int W);
// Workgroups of the persistent column kernel (= partials per lattice) when it may use `cus` CUs.
int gpt_pow2_col_blocks(int H, int W, int cus);
// One pass of the three: 0 = row forward (A), 1 = column (B; plus the second reduction stage when the op has
// no pass C), 2 = row inverse (C; sums B's partials). A pass the op does not have is a no-op. `col_cus`
// must be the same for the three passes of one lattice.
hipError_t gpt_pow2_pass(int pass, int op, const LatArgs& a, cplx* ws, double* partials, int col_cus,
                         hipStream_t st);
// Slab decomposition (gpt_pow2.hip): column-tile width for lines of length H and the three passes on
// caller-owned spectrum buffers.
int gpt_pow2_col_tile(int H);
// Slab passes also take lines of up to 32768 points (four-step route, gpt_long.hip): shapes, radices (0 = the
// line fits one workgroup, < 0 = unsupported), scratch size of the row passes, column tile / grid of pass 1.
bool gpt_slab_supported(int H, int W);
int gpt_long_row_split_radix(int W);
int gpt_long_col_split_radix(int H);
int gpt_slab_col_tile(int H);
int gpt_slab_col_blocks(int H, int ncols, int cus);
// `edge` (pass 1, may be null): 4 H doubles -- cov[:, 0], cov[:, W/2] staged by the caller, then d/dcov of the same two
// columns written by the pass (a rank that owns only its columns of cov keeps these two apart).
hipError_t gpt_pow2_slab_pass(int pass, int op, const LatArgs& a, cplx* spec, int ld, double alpha, double* partials,
                              int cus, double* edge, hipStream_t st);

// Non-centred model density: inverse rows -> observation model -> forward rows in place on the workspace, between
// pass 1 of OP_GP_INV_RFFT2 and pass 1 of OP_INV_VJP (rows of up to 8192 points). o.partials holds
// batch * gpt_pow2_obs_blocks(H, W) * GPT_PART doubles.
bool gpt_pow2_obs_supported(int H, int W);
int gpt_pow2_obs_blocks(int H, int W);
hipError_t gpt_pow2_obs_rows(const LatArgs& a, cplx* ws, double* partials, int col_cus, const ObsRowArgs& o,
                             hipStream_t st);

// Contiguous complex lines of n = 256 ... 4096 points (power of two), in place, with the register-tile kernels.
bool gpt_pow2_line_fft_supported(int n);
hipError_t gpt_pow2_line_fft(int n, bool inverse, cplx* data, int64_t nlines, const cplx* tw, hipStream_t st);

// --- generic large path (gpt_generic.hip): any H, W whose lines fit LDS
bool gpt_generic_supported(int H, int W);
size_t gpt_generic_workspace_bytes(int H, int W);
int gpt_generic_col_blocks(int H, int W);
hipError_t gpt_launch_generic(int op, const LatArgs& a, cplx* ws, double* partials, hipStream_t st);

// --- any-length path (gpt_anyn.hip): axes no other path takes (long 1-D series, long non-power-of-two axes), lines of up
// to 2^23 points by four-step and Bluestein transforms; owns its plans, tables and scratch
struct gpt_anyn;
bool gpt_anyn_supported(int H, int W);
bool gpt_anyn_preferred(int H, int W); // ... and an axis has a prime factor > 127 (Bluestein beats the O(n p) Stockham stage)
int64_t gpt_anyn_chunk(int H, int W); // most lattices one gpt_launch_anyn call takes
gpt_anyn* gpt_anyn_create();
void gpt_anyn_destroy(gpt_anyn* s);
hipError_t gpt_launch_anyn(gpt_anyn* s, int op, const LatArgs& a, hipStream_t st);

// --- elementwise / reduction kernels (gpt_elem.hip)
struct ElemArgs {
    int H, W, dim1;
    int64_t batch;
    const double* in;  // real input
    const cplx* cin;   // complex input
    const double* par; // kernel parameters [batch][5] (2-D: sigma, l1, l2, p1, p2) or [batch][3] (1-D)
    int64_t in_bs, cin_bs;
    double* out;
    cplx* cout;
    double* min_eig;
    double nu, gamma_ratio; // Matern smoothness and tgamma(nu + ndim/2) / tgamma(nu)
    double nugget;          // added to the spectrum by EL_EXPQUAD / EL_MATERN (0 for the plain Stan functions)
};
enum ElemOp : int {
    EL_SCALE = 0,      // gp_evaluate_rfft(2)_scale
    EL_UNPACK = 1,     // gp_unpack_rfft(2)
    EL_PACK = 2,       // gp_pack_rfft(2)
    EL_LADJ = 3,       // gp_rfft(2)_log_abs_det_jac
    EL_LADJ_VJP = 4,   // its gradient w.r.t. cov
    EL_EXPQUAD = 5,    // gp_periodic_exp_quad_cov_rfft(2)
    EL_MATERN = 6,     // gp_periodic_matern_cov_rfft(2)
    EL_EXPQUAD_VJP = 7,
    EL_MATERN_VJP = 8,
    EL_EXPAND = 9      // expand_rfft
};
// `partials` must hold batch * gpt_elem_blocks(H, W) * 8 doubles for the reducing ops.
int gpt_elem_blocks(int H, int W);
hipError_t gpt_launch_elem(int op, const ElemArgs& e, double* partials, hipStream_t st);
// Second (atomic-free) reduction stage. `partials` is [batch][nblk][GPT_PART] doubles: slots 0..4 are
// sums, slot 5 a minimum. Emits out_sum[b * K + k] = sum_k (+ add for k == 0) for k < K (if out_sum) and
// out_min[b] (if out_min).
#ifndef GPT_PART
#define GPT_PART 8
#endif
hipError_t gpt_launch_finalize(const double* partials, int nblk, int64_t batch, int K, double add,
                               double* out_sum, double* out_min, hipStream_t st);
// Row-major <-> column-major (Stan/Eigen) transposes of a batch of R x C matrices of `esz`-byte elements.
hipError_t gpt_launch_transpose(const void* in, void* out, int R, int C, int64_t batch, int esz, hipStream_t st);
// several R[k] x C[k] matrices of esz-byte elements in one launch
#define GPT_XPOSE_MAX 4
struct XposeBatch {
    const void* in[GPT_XPOSE_MAX];
    void* out[GPT_XPOSE_MAX];
    int R[GPT_XPOSE_MAX], C[GPT_XPOSE_MAX];
    int first_tile[GPT_XPOSE_MAX]; // filled by the launcher
    int n;
};
hipError_t gpt_launch_transpose_multi(XposeBatch b, int esz, hipStream_t st);

// --- observation models on the latent field (gpt_obs.hip); families as in include/gptools_hip.h
struct ObsArgs {
    int H, W, oh, ow, family;
    int64_t batch;
    const double* f;     // [batch][H][W]
    const double* obs;   // [batch][oh][ow], obs_bs elements apart (0 = shared)
    int64_t obs_bs;
    const double* theta; // [batch] or null
    double* fbar;        // [batch][H][W]
};
// partials: batch * gpt_obs_blocks(H, W) * GPT_PART doubles; slots 0..3 = log density, d/dtheta, sum of d/df, points
int gpt_obs_blocks(int H, int W);
hipError_t gpt_launch_obs(const ObsArgs& a, double* partials, hipStream_t st);
hipError_t gpt_launch_obs_combine(const double* sums, const double* psum, int64_t batch, double* lp, double* grad_theta,
                                  double* grad_loc_sum, double* nobs, hipStream_t st);
hipError_t gpt_launch_fill(double* p, int64_t n, double v, hipStream_t st);
// The same model density in ONE launch for lattices that fit a workgroup's LDS (gpt_small_obs.hip). `a` as for OP_INV_VJP
// (x = z, cov, out = d/dz, out_cov = d/dcov, zprior, min_eig), plans made; the rest here. a.kpar set instead of a.cov:
// the kernel-fused form (a.kmatern, a.knu, a.kgamma, a.knugget as for OP_LPDF_KVJP).
struct SmallObsArgs {
    const double* loc; // [batch][H][W], loc_bs apart (0 = shared), or null (0 everywhere)
    int64_t loc_bs;
    const double* obs; // [batch][oh][ow], obs_bs apart (0 = shared)
    int64_t obs_bs;
    int oh, ow, family;
    const double* theta; // [batch] or null
    double* f_out;       // [batch][H][W] or null
    double* grad_loc;    // [batch][H][W] or null: d lp / d f
    double* lp;          // [batch]
    double* grad_theta;  // [batch] or null
    double* grad_loc_sum; // [batch] or null
    double* kbar;         // a.kpar set (kernel-fused: no cov, a.out_cov null): [batch][5] d / d(sigma, length_scale, period), or null
};
hipError_t gpt_launch_small_obs(const LatArgs& a, const SmallObsArgs& o, hipStream_t st,
                                unsigned long long* done_flag = nullptr, unsigned long long done_seq = 0);
