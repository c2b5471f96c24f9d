// Declarations only (see fft1.stan). Signatures are those of the reference's stan/gptools/fft2.stan.
matrix gp_evaluate_rfft2_scale(matrix cov_rfft2, int width);
matrix gp_unpack_rfft2(complex_matrix z, int m);
complex_matrix gp_pack_rfft2(matrix z);
matrix gp_rfft2(matrix y, matrix loc, matrix cov_rfft2);
matrix gp_inv_rfft2(matrix z, matrix loc, matrix cov_rfft2);
real gp_rfft2_log_abs_det_jac(matrix cov_rfft2, int width);
real gp_rfft2_log_abs_det_jacobian(matrix cov_rfft2, int width) {
    reject("`gp_rfft2_log_abs_det_jacobian` has been renamed to `gp_rfft2_log_abs_det_jac` to comply with upcoming changes (see https://github.com/stan-dev/stanc3/issues/1470 for details).");
}
real gp_rfft2_lpdf(matrix y, matrix loc, matrix cov_rfft2);
matrix gp_periodic_exp_quad_cov_rfft2(int height, int width, real sigma, vector length_scale,
                                      vector period);
matrix gp_periodic_matern_cov_rfft2(real nu, int height, int width, real sigma, vector length_scale,
                                    vector period);
