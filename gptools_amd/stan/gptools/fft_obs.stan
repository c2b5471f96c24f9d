// Extension (no counterpart in the reference; declarations only, defined in gptools_hip.hpp): the log density of a
// non-centred model evaluated in one trip to the GPU. Include it next to gptools/fft.stan in a model that wants it:
//
//     #include gptools/fft_obs.stan
//     ...
//     // instead of   f = gp_inv_rfft2(z, rep_matrix(loc, H, W), rfft2_cov);  to_vector(z) ~ std_normal();
//     //              frequency[i, j] ~ neg_binomial_2(exp(f[i, j]), 1 / kappa)  over the unmasked quadrats
//     // (docs/trees/trees.stan:31-49 of the reference):
//     target += gp_inv_rfft2_obs_lpdf(z | 2, to_matrix(frequency), loc, rfft2_cov, 1 / kappa, 1);
//
// family: 0 normal (theta = standard deviation), 1 poisson_log (theta unused), 2 neg_binomial_2_log (theta = precision).
// obs: the data on the top-left corner of the lattice; negative counts and not-a-number are masked points.
// prior: 1 adds std_normal_lpdf(to_vector(z)). All normalising constants are kept.
real gp_inv_rfft2_obs_lpdf(matrix z, int family, matrix obs, real loc, matrix cov_rfft2, real theta, int prior);
real gp_inv_rfft_obs_lpdf(vector z, int family, vector obs, real loc, vector cov_rfft, real theta, int prior);

// The same with the spectrum given by its kernel instead of an array -- what docs/trees/trees.stan:26-29 computes first:
//     rfft2_cov = gp_periodic_matern_cov_rfft2(1.5, H, W, sigma, [length_scale, length_scale]', [H, W]') + epsilon
// -- so that one call replaces the spectrum, the density and (in reverse mode) the spectrum's adjoint, and five numbers
// travel where two H x (W/2+1) matrices did:
//     target += gp_inv_rfft2_obs_matern_lpdf(z | 2, to_matrix(frequency), loc, 1.5, sigma, ls, per, epsilon, 1 / kappa, 1);
// nu is data (a literal in every reference model). *_exp_quad_*: the squared-exponential kernel of fft2.stan:206-213.
real gp_inv_rfft2_obs_matern_lpdf(matrix z, int family, matrix obs, real loc, real nu, real sigma, vector length_scale,
                                  vector period, real nugget, real theta, int prior);
real gp_inv_rfft2_obs_exp_quad_lpdf(matrix z, int family, matrix obs, real loc, real sigma, vector length_scale, vector period,
                                    real nugget, real theta, int prior);
