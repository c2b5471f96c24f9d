// Real-FFT layer of the reference's stan/gptools/util.stan (lines 255-330) as declarations; bodies in
// libgptools_hip.so via include/gptools_hip.hpp. The reference's debugging helpers (assert_close, ...) and
// gp_conditional_loc_scale are NOT part of the accelerated path: a model that needs them keeps the
// reference's util.stan with its lines 255-330 replaced by the five declarations below (INTEGRATION.md).
complex_vector rfft(vector y);
complex_vector expand_rfft(complex_vector y, int n);
vector inv_rfft(complex_vector z, int n);
complex_matrix rfft2(matrix y);
matrix inv_rfft2(complex_matrix z, int m);
