"""
GPU parity of the any-length path (gpt_anyn.hip): axes no other path takes -- 1-D series of more than ~5000 points,
non-power-of-two axes longer than 4096, power-of-two shapes outside the tuned kernels. The reference takes any n
(util.stan:263-265, fft1.stan:64-66); its own tests stop at n = 100 (test_stan_functions.py), so these cases are held
against the CPU oracle (NumPy pocketfft) on the same seeded inputs at relative 1e-10, value and every gradient, and
layouts bit-exactly. Line plans covered: four-step on smooth lengths (8192, 6000 = 2 x 3000, 100000 = 25 x 4000,
65536 = 16 x 4096), Bluestein on lengths with a large prime factor (5209, 5003 rows, 10007, 8186 = 2 x 4093, 4099), and a plain
workgroup transform on the short axis.
"""
import numpy as np
import pytest

import gp_oracle as o

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def close(actual, desired, rtol=RTOL, scale=None):
    actual = actual.cpu().numpy() if hasattr(actual, "cpu") else np.asarray(actual)
    desired = np.asarray(desired)
    s = np.max(np.abs(desired)) if scale is None else scale
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=rtol * max(s, 1e-300))


@pytest.fixture(scope="module")
def fft(gpu_lib):
    from gptools_amd import fft as f

    return f


SIZES_1D = [8192, 6000, 5209, 10007, 8186, 65536, 100000]


@pytest.mark.parametrize("n", SIZES_1D)
def test_long_series_vs_oracle(fft, n):
    """Every 1-D function of fft1.stan on a long series, batch of 3 chains, loc shared."""
    assert fft.path_for_shape(1, n) == 3
    rng = np.random.default_rng(31 * n + 5)
    y = rng.normal(size=(3, n))
    loc = rng.normal(size=n)
    cov = o.gp_periodic_matern_cov_rfft(1.5, n, 1.2, n / 50.0, float(n)) + 1e-3
    cov = cov * np.exp(0.1 * rng.normal(size=cov.shape))
    z = rng.normal(size=(3, n))
    bar = rng.normal(size=(3, n))
    close(fft.rfft(y), o.rfft(y))
    close(fft.inv_rfft(o.rfft(y), n), y)
    assert np.array_equal(fft.gp_unpack_rfft(o.rfft(y), n).cpu().numpy(), o.gp_unpack_rfft(o.rfft(y), n))
    close(fft.gp_rfft(y, loc, cov), o.gp_rfft(y, loc, cov))
    close(fft.gp_inv_rfft(z, loc, cov), o.gp_inv_rfft(z, loc, cov))
    lp, gy, gc = fft.gp_rfft_lpdf_vjp(y, loc, cov)
    lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y, loc, cov)
    close(lp, lp_o)
    close(gy, gy_o)
    close(gc, gc_o)
    close(fft.gp_rfft_lpdf(y, loc, cov), lp_o)
    zb, cb = fft.gp_inv_rfft_vjp(z, cov, bar)
    zb_o, _, cb_o = o.gp_inv_rfft_vjp(z, loc, cov, bar)
    close(zb, zb_o)
    close(cb, cb_o)
    # reverse mode of gp_rfft: the map is linear in y - loc, so <zbar, gp_rfft(y)> == <ybar, y - loc>
    yb, _ = fft.gp_rfft_vjp(y, loc, cov, bar)
    lhs = np.sum(bar * o.gp_rfft(y, loc, cov), axis=-1)
    rhs = np.sum(yb.cpu().numpy() * (y - loc), axis=-1)
    np.testing.assert_allclose(rhs, lhs, rtol=1e-9)
    # round trip (test_fft.py:50-62)
    close(fft.gp_rfft(fft.gp_inv_rfft(z, loc, cov), loc, cov), z, rtol=1e-8)
    bad = cov.copy()
    bad[n // 3] = -1.0
    with pytest.raises(ValueError, match="not positive-definite"):
        fft.gp_rfft_lpdf(y, loc, bad)


SHAPES_2D = [(5000, 6), (6, 5000), (4099, 8), (8, 8186), (4100, 300), (16384, 32), (32, 32768), (3, 10007), (5003, 5),
             (6000, 4500),
             (2053, 12), (10, 4093), (1031, 1033)]  # lines that fit a workgroup but have a large prime factor: Bluestein


@pytest.mark.parametrize("shape", SHAPES_2D, ids=lambda s: f"{s[0]}x{s[1]}")
def test_long_axis_lattice_vs_oracle(fft, shape):
    """Every lattice function of fft2.stan on shapes with a long axis, batch of 2, against the oracle."""
    h, w = shape
    assert fft.path_for_shape(h, w) == 3
    rng = np.random.default_rng(100 * h + w)
    nb = 1 if h * w > 10_000_000 else 2
    y = rng.normal(size=(nb, h, w))
    loc = rng.normal(size=(h, w))
    cov = o.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.3, [h / 40.0, w / 30.0], [float(h), float(w)]) + 1e-3
    cov = cov * np.exp(0.1 * rng.normal(size=cov.shape))  # break the row symmetry
    z = rng.normal(size=(nb, h, w))
    bar = rng.normal(size=(nb, h, w))
    close(fft.rfft2(y), o.rfft2(y))
    close(fft.inv_rfft2(o.rfft2(y), w), y)
    close(fft.gp_rfft2(y, loc, cov), o.gp_rfft2(y, loc, cov))
    close(fft.gp_inv_rfft2(z, loc, cov), o.gp_inv_rfft2(z, loc, cov))
    lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, loc, cov)
    lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y, loc, cov)
    close(lp, lp_o)
    close(gy, gy_o)
    close(gc, gc_o)
    close(fft.gp_rfft2_lpdf(y, loc, cov), lp_o)
    zb, cb = fft.gp_inv_rfft2_vjp(z, cov, bar)
    zb_o, _, cb_o = o.gp_inv_rfft2_vjp(z, loc, cov, bar)
    close(zb, zb_o)
    close(cb, cb_o)
    yb, cb2 = fft.gp_rfft2_vjp(y, loc, cov, bar)
    yb_o, _, cb2_o = o.gp_rfft2_vjp(y, loc, cov, bar)
    close(yb, yb_o)
    close(cb2, cb2_o)
    bad = cov.copy()
    bad[h - 1, 0] = -1.0
    with pytest.raises(ValueError, match="not positive-definite"):
        fft.gp_rfft2_lpdf(y, loc, bad)


def test_million_point_series(fft):
    """n = 2^20 + 2^18 (four-step 320 x 4096) and n = 1000003 (prime: Bluestein on 2^21 points): value and gradient
    against the oracle, many chains per call crossing the scratch chunk."""
    for n, nb in ((1310720, 30), (1000003, 2)):
        assert fft.path_for_shape(1, n) == 3
        rng = np.random.default_rng(n)
        y = rng.normal(size=(nb, n))
        cov = o.gp_periodic_exp_quad_cov_rfft(n, 0.8, n / 200.0, float(n)) + 1e-2
        lp, gy, gc = fft.gp_rfft_lpdf_vjp(y, None, cov)
        for i in (0, nb - 1):
            lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y[i], np.zeros(n), cov)
            close(lp[i], lp_o)
            close(gy[i], gy_o)
        lp_all, _, gc_o = o.gp_rfft_lpdf_vjp(y, np.zeros(n), cov)
        close(lp, lp_all)
        close(gc, gc_o)


def test_no_alloc_entry_rejects_any_length_shapes(fft):
    """The allocation-free entry point cannot serve this path (its plans and scratch are allocated): clear error."""
    import torch

    ctx = fft.get_context()
    n = 8192
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    cov = torch.ones(n // 2 + 1, dtype=torch.float64, device="cuda")
    out = torch.zeros(n, dtype=torch.float64, device="cuda")
    lp = torch.zeros(1, dtype=torch.float64, device="cuda")
    with pytest.raises(NotImplementedError, match="any-length"):
        ctx.call("gpt_gp_rfft2_lpdf_vjp_noalloc", y.data_ptr(), n, None, 0, cov.data_ptr(), n // 2 + 1, 1, n, 1, 1,
                 lp.data_ptr(), out.data_ptr(), None, None)
