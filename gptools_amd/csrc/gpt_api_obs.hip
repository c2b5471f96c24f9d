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
    if (!z || !cov || !obs || !lp || !grad_z) return fail(c, GPT_ERR_SHAPE, "z, cov, obs, lp and grad_z must not be null");
    if (family != GPT_OBS_NORMAL && family != GPT_OBS_POISSON_LOG && family != GPT_OBS_NEG_BINOMIAL_2_LOG)
        return fail(c, GPT_ERR_SHAPE, "unknown observation family");
    if (family != GPT_OBS_POISSON_LOG && !theta) return fail(c, GPT_ERR_SHAPE, "this observation family needs theta");
    if (oh < 1 || ow < 1 || oh > H || ow > W)
        return fail(c, GPT_ERR_SHAPE, "the observed window must lie inside the lattice");
    // (a request of the host entry point to let the LAST launch publish its completion, gpt_ctx::done_*: only the
    // one-launch route below may take it up -- the composed route's first lattice launch must not)
    const bool armed = c->done_armed;
    c->done_armed = false;
    if (batch == 0) return GPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const int64_t hw = H * W, hwf = H * (W / 2 + 1);
    // Lattices that fit one workgroup's LDS (the docs models: 35 x 60, n = 100 ... 200): the whole density in ONE launch,
    // the lattice never leaves LDS (gpt_small_obs.hip; GPT_OBS_SMALL_FUSED=0: the composed route, for A/B timing).
    static const bool small_fused_off = [] { const char* e = getenv("GPT_OBS_SMALL_FUSED"); return e && e[0] == '0'; }();
    if (!small_fused_off && path_for_shape(H, W) == 0) {
        LatArgs a = base_args(H, W, batch, dim1);
        if ((rc = make_plan(c, a.W, a.pw, false)) != GPT_OK) return rc;
        if ((rc = make_plan(c, a.H, a.ph, false)) != GPT_OK) return rc;
        a.x = z; a.x_bs = z_bs;
        a.cov = cov; a.cov_bs = cov_bs;
        a.out = grad_z;
        a.out_cov = grad_cov;
        if (!grad_cov) {
            if ((rc = ensure(c, c->sink, c->sink_bytes, sizeof(double) * (size_t)batch * hwf))) return rc;
            a.out_cov = c->sink;
        }
        a.zprior = prior ? 1 : 0;
        a.min_eig = min_eig;
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
        unsigned long long* flag = nullptr;
        if (armed && batch == 1 && c->done_dev) {
            flag = c->done_dev;
            c->done_published = true;
            ++c->done_seq;
        }
        HIP_TRY(c, gpt_launch_small_obs(a, o, c->stream, flag, c->done_seq));
        return GPT_OK;
    }
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

} // extern "C"
