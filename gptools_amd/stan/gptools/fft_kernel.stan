// Extension (no counterpart in the reference; declarations only, defined in gptools_hip.hpp): the log density of the
// Fourier-basis Gaussian process with the spectrum given by its kernel instead of an array. A model that writes
//
//     cov_rfft2 = gp_periodic_matern_cov_rfft2(nu, H, W, sigma, length_scale, period) + epsilon;   // fft2.stan:228-247
//     y ~ gp_rfft2(loc, cov_rfft2);                                                                // fft2.stan:189-192
//
// makes three trips to the GPU per gradient evaluation (spectrum, density, the spectrum's adjoint) and moves two
// H x (W/2+1) matrices each way; after `#include gptools/fft_kernel.stan`
//
//     y ~ gp_rfft2_matern(loc, nu, sigma, length_scale, period, epsilon);
//
// is one trip, and only y, loc, five numbers and their adjoints travel. nu and the nugget epsilon are data (literals in
// every reference model). Same densities, same normalising constants as the two-step form.
real gp_rfft2_matern_lpdf(matrix y, matrix loc, real nu, real sigma, vector length_scale, vector period, real nugget);
real gp_rfft2_exp_quad_lpdf(matrix y, matrix loc, real sigma, vector length_scale, vector period, real nugget);
real gp_rfft_matern_lpdf(vector y, vector loc, real nu, real sigma, real length_scale, real period, real nugget);
real gp_rfft_exp_quad_lpdf(vector y, vector loc, real sigma, real length_scale, real period, real nugget);
