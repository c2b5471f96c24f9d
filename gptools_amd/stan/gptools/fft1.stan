// Declarations only: the bodies live in libgptools_hip.so (MI355X HIP kernels) and are bound by the C++
// user header include/gptools_hip.hpp. Compile with `stanc --allow-undefined` (gptools_amd.stan.compile_model
// does). Signatures are those of the reference's stan/gptools/fft1.stan, so models compile unchanged.
vector gp_evaluate_rfft_scale(vector cov_rfft, int n);
vector gp_unpack_rfft(complex_vector x, int n);
vector gp_rfft(vector y, vector loc, vector cov_rfft);
real gp_rfft_log_abs_det_jac(vector cov_rfft, int n);
real gp_rfft_log_abs_det_jacobian(vector cov_rfft, int n) {
    reject("`gp_rfft_log_abs_det_jacobian` has been renamed to `gp_rfft_log_abs_det_jac` to comply with upcoming changes (see https://github.com/stan-dev/stanc3/issues/1470 for details).");
}
real gp_rfft_lpdf(vector y, vector loc, vector cov_rfft);
complex_vector gp_pack_rfft(vector z);
vector gp_inv_rfft(vector z, vector loc, vector cov_rfft);
vector gp_periodic_exp_quad_cov_rfft(int n, real sigma, real length_scale, real period);
vector gp_periodic_matern_cov_rfft(real nu, int n, real sigma, real length_scale, real period);
