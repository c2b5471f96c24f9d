"""
Pin the CPU oracle (oracle/gp_oracle.py) against golden vectors produced by the reference's own NumPy
mirror (tests/golden/make_golden.py), mirroring python/util/tests/test_fft.py and the FFT configurations
of python/stan/tests/test_stan_functions.py:152-513 of the reference. Tolerance: rtol 1e-12 (the
reference's own tests use rtol 1e-7 / atol 1e-6).
"""
import numpy as np
import pytest
from conftest import KERNEL_NAMES, SHAPES_2D, SIZES_1D, load_golden, variants

import gp_oracle as o

RTOL = 1e-12
ATOL = 1e-12


def close(a, b, rtol=RTOL, atol=ATOL):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("n", SIZES_1D)
def test_fft1_golden(n):
    g = load_golden(f"fft1_n{n}.npz")
    y, loc, z = g["y"], g["loc"], g["z"]
    close(o.rfft(y), g["rfft_y"])
    close(o.inv_rfft(g["rfft_y"], n), g["irfft_rfft_y"])
    close(o.inv_rfft(g["rfft_y"], n), y)
    assert np.array_equal(o.gp_unpack_rfft(g["rfft_y"], n), g["unpack_rfft_y"])  # layout: bit-exact
    assert np.array_equal(o.gp_pack_rfft(z), g["pack_z"])
    close(o.expand_rfft(g["rfft_y"], n), g["expand_rfft_y"])
    for name, tag in variants(g):
        k = f"{name}_{tag}"
        cov = g[f"{k}_cov_rfft"]
        close(o.gp_evaluate_rfft_scale(cov, n), g[f"{k}_scale"])
        close(o.gp_rfft(y, loc, cov), g[f"{k}_transform_rfft"])
        close(o.gp_inv_rfft(z, loc, cov), g[f"{k}_transform_irfft"])
        close(o.gp_rfft_log_abs_det_jac(cov, n), g[f"{k}_ladj"])
        close(o.gp_rfft_lpdf(y, loc, cov), g[f"{k}_log_prob"])
        close(o.gp_rfft_lpdf(g["yb"], loc, cov), g[f"{k}_log_prob_batch"])
        # round trip (test_fft.py:50-62)
        close(o.gp_rfft(o.gp_inv_rfft(z, loc, cov), loc, cov), z, rtol=1e-9, atol=1e-9)
    for name in KERNEL_NAMES:
        if f"{name}_b_mvn_logpdf" in g:  # dense cross-check (test_fft.py:39-47)
            close(o.gp_rfft_lpdf(y, loc, g[f"{name}_b_cov_rfft"]), g[f"{name}_b_mvn_logpdf"], rtol=1e-9)


@pytest.mark.parametrize("shape", SHAPES_2D, ids=lambda s: f"{s[0]}x{s[1]}")
def test_fft2_golden(shape):
    h, w = shape
    g = load_golden(f"fft2_{h}x{w}.npz")
    y, loc, z = g["y"], g["loc"], g["z"]
    close(o.rfft2(y), g["rfft2_y"])
    close(o.inv_rfft2(g["rfft2_y"], w), g["irfft2_rfft2_y"])
    close(o.inv_rfft2(g["rfft2_y"], w), y)
    assert np.array_equal(o.gp_unpack_rfft2(g["rfft2_y"], w), g["unpack_rfft2_y"])
    assert np.array_equal(o.gp_pack_rfft2(z), g["pack2_z"])
    for name, tag in variants(g):
        k = f"{name}_{tag}"
        cov = g[f"{k}_cov_rfft2"]
        close(o.gp_evaluate_rfft2_scale(cov, w), g[f"{k}_scale"])
        close(o.gp_rfft2(y, loc, cov), g[f"{k}_transform_rfft2"])
        close(o.gp_inv_rfft2(z, loc, cov), g[f"{k}_transform_irfft2"])
        close(o.gp_rfft2_log_abs_det_jac(cov, w), g[f"{k}_ladj"])
        close(o.gp_rfft2_lpdf(y, loc, cov), g[f"{k}_log_prob"])
        close(o.gp_rfft2_lpdf(g["yb"], loc, cov), g[f"{k}_log_prob_batch"])
        close(o.gp_rfft2(o.gp_inv_rfft2(z, loc, cov), loc, cov), z, rtol=1e-9, atol=1e-9)
    for name in KERNEL_NAMES:
        if f"{name}_b_mvn_logpdf" in g:
            close(o.gp_rfft2_lpdf(y, loc, g[f"{name}_b_cov_rfft2"]), g[f"{name}_b_mvn_logpdf"], rtol=1e-9)


def test_kernels_golden():
    g = load_golden("kernels.npz")
    for idx, (m, n, sigma, l1, l2, p1, p2) in enumerate(g["cases"]):
        m, n = int(m), int(n)
        ls, per = np.array([l1, l2]), np.array([p1, p2])
        close(o.gp_periodic_exp_quad_cov_rfft2(m, n, sigma, ls, per), g[f"c{idx}_expquad2"])
        close(o.gp_periodic_matern_cov_rfft2(1.5, m, n, sigma, ls, per), g[f"c{idx}_matern32_2"])
        close(o.gp_periodic_matern_cov_rfft2(2.5, m, n, sigma, ls, per), g[f"c{idx}_matern52_2"])
        close(o.gp_periodic_exp_quad_cov_rfft(n, sigma, l1, p1), g[f"c{idx}_expquad1"])
        close(o.gp_periodic_matern_cov_rfft(1.5, n, sigma, l1, p1), g[f"c{idx}_matern32_1"])
        close(o.gp_periodic_matern_cov_rfft(2.5, n, sigma, l1, p1), g[f"c{idx}_matern52_1"])


def test_rejects():
    """Stan `reject` paths: fft1.stan:19-21, fft2.stan:21-23, util.stan:274-276."""
    with pytest.raises(o.NotPositiveDefinite, match="not positive-definite"):
        o.gp_evaluate_rfft_scale(np.array([1.0, -0.5, 1.0]), 4)
    with pytest.raises(o.NotPositiveDefinite, match="minimum eigenvalue is -6"):
        o.gp_evaluate_rfft2_scale(np.array([[1.0, -1.0], [1.0, 1.0], [1.0, 1.0], [1.0, 1.0]]), 3)
    with pytest.raises(ValueError, match="expected complex vector with 3 elements but got 4"):
        o.expand_rfft(np.zeros(4, dtype=complex), 4)
    # c == 0 passes the strict `< 0` check (quirk 1 of SURVEY 8a)
    with np.errstate(divide="ignore"):
        assert np.isinf(o.gp_rfft_log_abs_det_jac(np.array([1.0, 0.0, 1.0]), 4))
