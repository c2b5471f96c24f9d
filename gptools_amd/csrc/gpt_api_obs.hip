// Non-centred model density on the device: gpt_gp_inv_rfft2_obs_lpdf_vjp.
#include "gpt_ctx.h"
using namespace gptd;

namespace {

// Fused route of gpt_gp_inv_rfft2_obs_lpdf_vjp on the power-of-two path: column pass of the inverse transform, one
// row kernel (inverse rows -> observation model -> forward rows, in place on the workspace), column pass of the
// reverse mode. sums: [batch][4] on the device.
int run_obs_fused(gpt_ctx* c, int family, const double* z, int64_t z_bs, const double* loc, int64_t loc_bs,
                  const double* cov, int64_t cov_bs, const double* obs, int64_t obs_bs, int oh, int ow,
                  const double* theta, int H, int W, int64_t batch, double* f_out, double* fbar_out, double* grad_z,
                  double* grad_cov, double* min_eig, double* sums, double* psum) {
    int rc;
    LatArgs a = base_args(H, W, batch, 0);
    if ((rc = make_plan(c, W, a.pw)) != GPT_OK) return rc;
    if ((rc = make_plan(c, H, a.ph)) != GPT_OK) return rc;
    const int R = gpt_pow2_sub_radix(OP_GP_INV_RFFT2, H, W);
    if (R && (rc = get_twiddles(c, H / R, &a.tw_sub)) != GPT_OK) return rc;
    if (c->total_cus == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
        c->total_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const size_t per = gpt_pow2_workspace_bytes(H, W);
    int64_t chunk = (int64_t)((size_t)(128u << 20) / per);
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if (chunk > 32768) chunk = 32768;
    const int nblk = gpt_pow2_col_blocks(H, W, c->total_cus), nobs = gpt_pow2_obs_blocks(H, W);
    if ((rc = ensure(c, c->ws, c->ws_bytes, per * (size_t)chunk)) != GPT_OK) return rc;
    if ((rc = ensure(c, c->partials, c->partials_bytes, sizeof(double) * GPT_PART * (size_t)(nblk + nobs) * (size_t)chunk)) !=
        GPT_OK)
        return rc;
    double* obs_partials = c->partials + GPT_PART * (size_t)nblk * (size_t)chunk;
    const int64_t hw = (int64_t)H * W, hwf = (int64_t)H * (W / 2 + 1);
    if (!grad_cov) {
        if ((rc = ensure(c, c->sink, c->sink_bytes, sizeof(double) * (size_t)chunk * hwf))) return rc;
    }
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = batch - b0 < chunk ? batch - b0 : chunk;
        LatArgs si = a; // gp_inv_rfft2
        si.batch = nb;
        si.x = z + b0 * z_bs; si.x_bs = z_bs;
        si.loc = loc ? loc + b0 * loc_bs : nullptr; si.loc_bs = loc_bs;
        si.cov = cov + b0 * cov_bs; si.cov_bs = cov_bs;
        si.min_eig = min_eig ? min_eig + b0 : nullptr;
        HIP_TRY(c, gpt_pow2_pass(1, OP_GP_INV_RFFT2, si, c->ws, c->partials, c->total_cus, c->stream));
        ObsRowArgs o;
        memset(&o, 0, sizeof o);
        o.family = family; o.oh = oh; o.ow = ow;
        o.obs = obs + b0 * obs_bs; o.obs_bs = obs_bs;
        o.theta = theta ? theta + b0 : nullptr;
        o.f_out = f_out ? f_out + b0 * hw : nullptr;
        o.fbar_out = fbar_out ? fbar_out + b0 * hw : nullptr;
        o.partials = obs_partials;
        HIP_TRY(c, gpt_pow2_obs_rows(si, c->ws, c->partials, c->total_cus, o, c->stream));
        HIP_TRY(c, gpt_launch_finalize(obs_partials, nobs, nb, 4, 0.0, sums + 4 * b0, nullptr, c->stream));
        LatArgs sv = a; // its reverse mode: the workspace holds the row transforms of d lp / d f
        sv.batch = nb;
        sv.x = si.x; sv.x_bs = z_bs;
        sv.cov = si.cov; sv.cov_bs = cov_bs;
        sv.out = grad_z + b0 * hw;
        sv.out_cov = grad_cov ? grad_cov + b0 * hwf : c->sink;
        sv.zprior = psum ? 1 : 0;
        sv.out_lpdf = psum ? psum + b0 : nullptr;
        HIP_TRY(c, gpt_pow2_pass(1, OP_INV_VJP, sv, c->ws, c->partials, c->total_cus, c->stream));
    }
    return GPT_OK;
}

// Lattices that fit one workgroup's LDS: the whole density in ONE launch (gpt_small_obs.hip). cov != null: the spectrum
// from memory; params != null: evaluated from the kernel parameters where needed (no cov, no d/dcov: params_bar).
int small_obs_launch(gpt_ctx* c, bool armed, int family, const double* z, int64_t z_bs, const double* loc, int64_t loc_bs,
                     const double* cov, int64_t cov_bs, const double* params, int matern, double nu, double nugget,
                     const double* obs, int64_t obs_bs, int64_t oh, int64_t ow, const double* theta, int prior, int64_t H,
                     int64_t W, int64_t batch, int dim1, double* lp, double* f_out, double* grad_z, double* grad_loc,
                     double* grad_loc_sum, double* grad_cov, double* params_bar, double* grad_theta, double* min_eig) {
    int rc;
    LatArgs a = base_args(H, W, batch, dim1);
    if ((rc = make_plan(c, a.W, a.pw, false)) != GPT_OK) return rc;
    if ((rc = make_plan(c, a.H, a.ph, false)) != GPT_OK) return rc;
    a.x = z; a.x_bs = z_bs;
    a.cov = cov; a.cov_bs = cov_bs;
    a.out = grad_z;
    a.out_cov = grad_cov; // (may be null: the reverse mode then keeps d/dcov to itself, put_cbar)
    a.zprior = prior ? 1 : 0;
    a.min_eig = min_eig;
    if (params) {
        a.kpar = params;
        a.kmatern = matern ? 1 : 0;
        a.knu = nu;
        a.kgamma = matern ? std::tgamma(nu + (dim1 ? 0.5 : 1.0)) / std::tgamma(nu) : 0.0; // fft1.stan:173, fft2.stan:245
        a.knugget = nugget;
    }
    SmallObsArgs o;
    o.loc = loc; o.loc_bs = loc_bs;
    o.obs = obs; o.obs_bs = obs_bs;
    o.oh = (int)oh; o.ow = (int)ow; o.family = family;
    o.theta = theta;
    o.f_out = f_out;
    o.grad_loc = grad_loc;
    o.lp = lp;
    o.grad_theta = grad_theta;
    o.grad_loc_sum = grad_loc_sum;
    o.kbar = params_bar;
    unsigned long long* flag = nullptr;
    if (armed && batch == 1 && c->done_dev) { // (the host entry point's request, gpt_ctx::done_*)
        flag = c->done_dev;
        c->done_published = true;
        ++c->done_seq;
    }
    HIP_TRY(c, gpt_launch_small_obs(a, o, c->stream, flag, c->done_seq));
    return GPT_OK;
}

int check_obs_args(gpt_ctx* c, int family, const void* z, const void* spec, const void* obs, const void* lp, const void* grad_z,
                   const double* theta, int64_t oh, int64_t ow, int64_t H, int64_t W) {
    if (!z || !spec || !obs || !lp || !grad_z)
        return fail(c, GPT_ERR_SHAPE, "z, cov (or params), obs, lp and grad_z must not be null");
    if (family != GPT_OBS_NORMAL && family != GPT_OBS_POISSON_LOG && family != GPT_OBS_NEG_BINOMIAL_2_LOG)
        return fail(c, GPT_ERR_SHAPE, "unknown observation family");
    if (family != GPT_OBS_POISSON_LOG && !theta) return fail(c, GPT_ERR_SHAPE, "this observation family needs theta");
    if (oh < 1 || ow < 1 || oh > H || ow > W)
        return fail(c, GPT_ERR_SHAPE, "the observed window must lie inside the lattice");
    return GPT_OK;
}

// GPT_OBS_SMALL_FUSED=0: the composed route for lattices that fit a workgroup too (A/B timing)
bool small_obs_enabled() {
    static const bool off = [] { const char* e = getenv("GPT_OBS_SMALL_FUSED"); return e && e[0] == '0'; }();
    return !off;
}

} // namespace

extern "C" {

// Non-centred model density: f = gp_inv_rfft2(z, loc, cov) (fft2.stan:126-130), the observation
// model of the docs workloads on the device (gpt_obs.hip), the reverse mode of the transform, and optionally the
// white-noise prior of trees.stan:37. Three transforms' worth of passes + two streaming kernels; f and d/df never
// leave HBM.
int gpt_gp_inv_rfft2_obs_lpdf_vjp(gpt_ctx* c, int family, const double* z, int64_t z_bs, const double* loc,
                                  int64_t loc_bs, const double* cov, int64_t cov_bs, const double* obs, int64_t obs_bs,
                                  int64_t oh, int64_t ow, const double* theta, int prior, int64_t H, int64_t W,
                                  int64_t batch, int dim1, double* lp, double* f_out, double* grad_z, double* grad_loc,
                                  double* grad_loc_sum, double* grad_cov, double* grad_theta, double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if ((rc = check_obs_args(c, family, z, cov, obs, lp, grad_z, theta, oh, ow, H, W))) return rc;
    // (a request of the host entry point to let the LAST launch publish its completion, gpt_ctx::done_*: only the
    // one-launch route below may take it up -- the composed route's first lattice launch must not)
    const bool armed = c->done_armed;
    c->done_armed = false;
    if (batch == 0) return GPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t hw = H * W, hwf = H * (W / 2 + 1);
    if (small_obs_enabled() && path_for_shape(H, W) == 0)
        return small_obs_launch(c, armed, family, z, z_bs, loc, loc_bs, cov, cov_bs, nullptr, 0, 0.0, 0.0, obs, obs_bs, oh, ow, theta,
                                prior, H, W, batch, dim1, lp, f_out, grad_z, grad_loc, grad_loc_sum, grad_cov, nullptr,
                                grad_theta, min_eig);
    // GPT_OBS_FUSED=1 (read per call): one row kernel does inverse rows -> observation model -> forward rows in place
    // on the workspace. Same results; measured slower than the three composed steps on MI355X (200
    // VGPRs leave two waves per SIMD to hide the densities' dependent chains), so it is off unless asked for.
    const char* fused_env = getenv("GPT_OBS_FUSED");
    const bool allow_fused = fused_env && fused_env[0] == '1';
    const int nblk = gpt_obs_blocks((int)H, (int)W);
    if (allow_fused && !dim1 && path_for_shape(H, W) == 1 && gpt_pow2_obs_supported((int)H, (int)W)) {
        if ((rc = ensure(c, c->oscr, c->oscr_bytes, sizeof(double) * 5 * (size_t)batch)) != GPT_OK) return rc;
        double* sums = c->oscr;
        double* psum = sums + 4 * batch;
        if ((rc = run_obs_fused(c, family, z, z_bs, loc, loc_bs, cov, cov_bs, obs, obs_bs, (int)oh, (int)ow, theta, (int)H,
                                (int)W, batch, f_out, grad_loc, grad_z, grad_cov, min_eig, sums, prior ? psum : nullptr)) !=
            GPT_OK)
            return rc;
        HIP_TRY(c, gpt_launch_obs_combine(sums, prior ? psum : nullptr, batch, lp, grad_theta, grad_loc_sum, nullptr,
                                          c->stream));
        return GPT_OK;
    }
    // composed: three library calls; lattices per round keep library-owned f / d/df at <= ~1 GiB together
    int64_t chunk = (int64_t)((size_t)(1u << 30) / (16 * (size_t)hw));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if (chunk > 32768) chunk = 32768;
    const size_t nscr = (f_out ? 0 : (size_t)chunk * hw) + (grad_loc ? 0 : (size_t)chunk * hw) + 5 * (size_t)chunk;
    if ((rc = ensure(c, c->oscr, c->oscr_bytes, sizeof(double) * nscr)) != GPT_OK) return rc;
    double* sums = c->oscr;
    double* psum = sums + 4 * chunk;
    double* scr = psum + chunk;
    double* f_scr = f_out ? nullptr : scr;
    double* g_scr = grad_loc ? nullptr : scr + (f_out ? 0 : (size_t)chunk * hw);
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = batch - b0 < chunk ? batch - b0 : chunk;
        double* f = f_out ? f_out + b0 * hw : f_scr;
        double* fbar = grad_loc ? grad_loc + b0 * hw : g_scr;
        const double* zb = z + b0 * z_bs;
        const double* cb = cov + b0 * cov_bs;
        if ((rc = gpt_gp_inv_rfft2(c, zb, z_bs, loc ? loc + b0 * loc_bs : nullptr, loc_bs, cb, cov_bs, H, W, nb, dim1, f,
                                   min_eig ? min_eig + b0 : nullptr)) != GPT_OK)
            return rc;
        if ((rc = ensure(c, c->partials, c->partials_bytes, sizeof(double) * GPT_PART * (size_t)nblk * (size_t)nb)) != GPT_OK)
            return rc;
        ObsArgs o;
        memset(&o, 0, sizeof o);
        o.H = (int)H; o.W = (int)W; o.oh = (int)oh; o.ow = (int)ow; o.family = family;
        o.batch = nb;
        o.f = f;
        o.obs = obs + b0 * obs_bs; o.obs_bs = obs_bs;
        o.theta = theta ? theta + b0 : nullptr;
        o.fbar = fbar;
        HIP_TRY(c, gpt_launch_obs(o, c->partials, c->stream));
        HIP_TRY(c, gpt_launch_finalize(c->partials, nblk, nb, 4, 0.0, sums, nullptr, c->stream));
        if ((rc = inv_rfft2_vjp(c, zb, z_bs, cb, cov_bs, fbar, hw, H, W, nb, dim1, grad_z + b0 * hw,
                                grad_cov ? grad_cov + b0 * hwf : nullptr, prior ? psum : nullptr)) != GPT_OK)
            return rc;
        HIP_TRY(c, gpt_launch_obs_combine(sums, prior ? psum : nullptr, nb, lp + b0, grad_theta ? grad_theta + b0 : nullptr,
                                          grad_loc_sum ? grad_loc_sum + b0 : nullptr, nullptr, c->stream));
    }
    return GPT_OK;
}

// The same density with the spectrum given by its kernel parameters (the inverse-transform half:
// fft2.stan:206-247 feeding fft2.stan:126-130): params [batch][5] = (sigma, length_scale[2], period[2]) ([batch][3] for
// the 1-D functions), matern / nu / nugget as for gpt_gp_rfft2_lpdf_kernel_vjp; params_bar the reverse mode down to them.
// trees.stan rebuilds gp_periodic_matern_cov_rfft2 on every gradient evaluation: behind the Stan interface that is three
// calls (spectrum, density, the spectrum's reverse mode) and two H x (W/2+1) arrays each way; here one call and five
// numbers. Lattices that fit a workgroup's LDS never hold the spectrum anywhere (gpt_small_obs.hip, KMODE); every other
// shape composes the three calls on library-owned scratch, same results.
int gpt_gp_inv_rfft2_obs_lpdf_kernel_vjp(gpt_ctx* c, int family, int matern, double nu, const double* z, int64_t z_bs,
                                         const double* loc, int64_t loc_bs, const double* params, double nugget,
                                         const double* obs, int64_t obs_bs, int64_t oh, int64_t ow, const double* theta,
                                         int prior, int64_t H, int64_t W, int64_t batch, int dim1, double* lp,
                                         double* grad_z, double* grad_loc_sum, double* params_bar, double* grad_theta,
                                         double* min_eig) {
    int rc = check_shape(c, H, W, batch, dim1);
    if (rc) return rc;
    if ((rc = check_obs_args(c, family, z, params, obs, lp, grad_z, theta, oh, ow, H, W))) return rc;
    if (matern && !(nu > 0)) return fail(c, GPT_ERR_SHAPE, "nu must be positive");
    const bool armed = c->done_armed;
    c->done_armed = false;
    if (batch == 0) return GPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // the in-kernel evaluation knows exp-quad and half-integer Matern smoothness (nu = 0.5, 1.5, ... 5.5)
    const bool nu_ok = !matern || (nu <= 5.5 && nu - std::floor(nu) == 0.5 /* kern_eval */);
    if (small_obs_enabled() && nu_ok && path_for_shape(H, W) == 0)
        return small_obs_launch(c, armed, family, z, z_bs, loc, loc_bs, nullptr, 0, params, matern, nu, nugget, obs, obs_bs, oh, ow,
                                theta, prior, H, W, batch, dim1, lp, nullptr, grad_z, nullptr, grad_loc_sum, nullptr, params_bar,
                                grad_theta, min_eig);
    // composed: cov and d/dcov live in scratch, a chunk of lattices at a time (as gpt_gp_rfft2_lpdf_kernel_vjp)
    const size_t hwf = (size_t)H * (size_t)(W / 2 + 1);
    int64_t chunk = (int64_t)((size_t)(128u << 20) / (hwf * sizeof(double)));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    if ((rc = ensure(c, c->kscr, c->kscr_bytes, 2 * sizeof(double) * hwf * (size_t)chunk)) != GPT_OK) return rc;
    double* cov = c->kscr;
    double* dcov = c->kscr + hwf * (size_t)chunk;
    const int npar = dim1 ? 3 : 5;
    for (int64_t b0 = 0; b0 < batch; b0 += chunk) {
        const int64_t nb = batch - b0 < chunk ? batch - b0 : chunk;
        ElemArgs e = elem_args(H, W, nb, dim1);
        e.par = params + b0 * npar;
        e.out = cov;
        e.nu = nu;
        e.gamma_ratio = matern ? std::tgamma(nu + (dim1 ? 0.5 : 1.0)) / std::tgamma(nu) : 0.0; // fft1.stan:173, fft2.stan:245
        e.nugget = nugget;
        if ((rc = run_elem(c, matern ? EL_MATERN : EL_EXPQUAD, e, false, 0, 0.0, nullptr, nullptr)) != GPT_OK) return rc;
        if ((rc = gpt_gp_inv_rfft2_obs_lpdf_vjp(c, family, z + b0 * z_bs, z_bs, loc ? loc + b0 * loc_bs : nullptr, loc_bs, cov,
                                                (int64_t)hwf, obs + b0 * obs_bs, obs_bs, oh, ow, theta ? theta + b0 : nullptr,
                                                prior, H, W, nb, dim1, lp + b0, nullptr, grad_z + b0 * H * W, nullptr,
                                                grad_loc_sum ? grad_loc_sum + b0 : nullptr, params_bar ? dcov : nullptr,
                                                grad_theta ? grad_theta + b0 : nullptr, min_eig ? min_eig + b0 : nullptr)))
            return rc;
        if (params_bar) {
            if (matern)
                rc = gpt_gp_periodic_matern_cov_rfft2_vjp(c, nu, H, W, params + b0 * npar, dcov, (int64_t)hwf, nb, dim1,
                                                          params_bar + b0 * npar);
            else
                rc = gpt_gp_periodic_exp_quad_cov_rfft2_vjp(c, H, W, params + b0 * npar, dcov, (int64_t)hwf, nb, dim1,
                                                            params_bar + b0 * npar);
            if (rc) return rc;
        }
    }
    return GPT_OK;
}

} // extern "C"
