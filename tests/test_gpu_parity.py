"""
GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(gptools_amd.fft -> libgptools_hip.so), against
  * the golden vectors generated from the reference's NumPy mirror (tests/golden), and
  * the CPU oracle (oracle/gp_oracle.py) on the same seeded inputs, including the reverse-mode gradients.
Mirrors python/util/tests/test_fft.py and python/stan/tests/test_stan_functions.py:152-513 of the reference
(one check per Stan function, odd/even x odd/even shapes, dense-MVN cross-check, reject paths).

Tolerances: the north star asks for relative 1e-10 on log densities and gradients; packing layouts are
checked bit-exactly (pure data movement).
"""
import numpy as np
import pytest
from conftest import KERNEL_NAMES, SHAPES_2D, SIZES_1D, load_golden, variants

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


@pytest.mark.parametrize("n", SIZES_1D)
def test_fft1_golden(fft, n):
    import torch

    g = load_golden(f"fft1_n{n}.npz")
    y, loc, z = g["y"], g["loc"], g["z"]
    close(fft.rfft(y), g["rfft_y"])
    close(fft.inv_rfft(g["rfft_y"], n), g["irfft_rfft_y"])
    assert np.array_equal(fft.gp_unpack_rfft(g["rfft_y"], n).cpu().numpy(), g["unpack_rfft_y"])
    assert np.array_equal(fft.gp_pack_rfft(z).cpu().numpy(), g["pack_z"])
    close(fft.expand_rfft(g["rfft_y"], n), g["expand_rfft_y"])
    for name, tag in variants(g):
        k = f"{name}_{tag}"
        cov = g[f"{k}_cov_rfft"]
        close(fft.gp_evaluate_rfft_scale(cov, n), g[f"{k}_scale"])
        close(fft.gp_rfft(y, loc, cov), g[f"{k}_transform_rfft"])
        close(fft.gp_inv_rfft(z, loc, cov), g[f"{k}_transform_irfft"])
        close(fft.gp_rfft_log_abs_det_jac(cov, n), g[f"{k}_ladj"])
        close(fft.gp_rfft_lpdf(y, loc, cov), g[f"{k}_log_prob"])
        close(fft.gp_rfft_lpdf(g["yb"], loc, cov), g[f"{k}_log_prob_batch"])
        # NumPy-mirror spelling of the same calls
        close(fft.evaluate_log_prob_rfft(y, loc, cov_rfft=cov), g[f"{k}_log_prob"])
        # round trip (test_fft.py:50-62)
        close(fft.gp_rfft(fft.gp_inv_rfft(z, loc, cov), loc, cov), z, rtol=1e-8)
    for name in KERNEL_NAMES:
        if f"{name}_b_mvn_logpdf" in g:
            close(fft.gp_rfft_lpdf(y, loc, g[f"{name}_b_cov_rfft"]), g[f"{name}_b_mvn_logpdf"], rtol=1e-9)
    assert torch.cuda.is_available()


@pytest.mark.parametrize("shape", SHAPES_2D, ids=lambda s: f"{s[0]}x{s[1]}")
def test_fft2_golden(fft, shape):
    h, w = shape
    g = load_golden(f"fft2_{h}x{w}.npz")
    y, loc, z = g["y"], g["loc"], g["z"]
    close(fft.rfft2(y), g["rfft2_y"])
    close(fft.inv_rfft2(g["rfft2_y"], w), g["irfft2_rfft2_y"])
    assert np.array_equal(fft.gp_unpack_rfft2(g["rfft2_y"], w).cpu().numpy(), g["unpack_rfft2_y"])
    assert np.array_equal(fft.gp_pack_rfft2(z).cpu().numpy(), g["pack2_z"])
    for name, tag in variants(g):
        k = f"{name}_{tag}"
        cov = g[f"{k}_cov_rfft2"]
        close(fft.gp_evaluate_rfft2_scale(cov, w), g[f"{k}_scale"])
        close(fft.gp_rfft2(y, loc, cov), g[f"{k}_transform_rfft2"])
        close(fft.gp_inv_rfft2(z, loc, cov), g[f"{k}_transform_irfft2"])
        close(fft.gp_rfft2_log_abs_det_jac(cov, w), g[f"{k}_ladj"])
        close(fft.gp_rfft2_lpdf(y, loc, cov), g[f"{k}_log_prob"])
        close(fft.gp_rfft2_lpdf(g["yb"], loc, cov), g[f"{k}_log_prob_batch"])
        close(fft.evaluate_log_prob_rfft2(y, loc, cov_rfft2=cov), g[f"{k}_log_prob"])
        close(fft.gp_rfft2(fft.gp_inv_rfft2(z, loc, cov), loc, cov), z, rtol=1e-8)
    for name in KERNEL_NAMES:
        if f"{name}_b_mvn_logpdf" in g:
            close(fft.gp_rfft2_lpdf(y, loc, g[f"{name}_b_cov_rfft2"]), g[f"{name}_b_mvn_logpdf"], rtol=1e-9)


def test_kernels_golden(fft):
    g = load_golden("kernels.npz")
    for idx, (m, n, sigma, l1, l2, p1, p2) in enumerate(g["cases"]):
        m, n = int(m), int(n)
        ls, per = [l1, l2], [p1, p2]
        close(fft.gp_periodic_exp_quad_cov_rfft2(m, n, sigma, ls, per), g[f"c{idx}_expquad2"])
        close(fft.gp_periodic_matern_cov_rfft2(1.5, m, n, sigma, ls, per), g[f"c{idx}_matern32_2"])
        close(fft.gp_periodic_matern_cov_rfft2(2.5, m, n, sigma, ls, per), g[f"c{idx}_matern52_2"])
        close(fft.gp_periodic_exp_quad_cov_rfft(n, sigma, l1, p1), g[f"c{idx}_expquad1"])
        close(fft.gp_periodic_matern_cov_rfft(1.5, n, sigma, l1, p1), g[f"c{idx}_matern32_1"])
        close(fft.gp_periodic_matern_cov_rfft(2.5, n, sigma, l1, p1), g[f"c{idx}_matern52_1"])


GRAD_SHAPES = [(5, 7), (5, 8), (6, 7), (6, 8), (2, 2), (3, 2), (2, 3), (1, 8), (1, 7), (35, 60), (64, 64), (27, 25),
               (100, 100)]  # (the largest square the one-workgroup kernel takes: only without its LDS stride padding)


@pytest.mark.parametrize("shape", GRAD_SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
def test_fft2_gradients_vs_oracle(fft, shape):
    """Reverse mode of every differentiable function, batch of 3, loc shared, cov NOT row-symmetric."""
    h, w = shape
    rng = np.random.default_rng(7000 + 131 * h + w)
    y = rng.normal(size=(3, h, w))
    loc = rng.normal(size=(h, w))
    cov = np.exp(rng.normal(size=(h, w // 2 + 1)))
    z = rng.normal(size=(3, h, w))
    bar = rng.normal(size=(3, h, w))
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
    close(fft.gp_rfft2_log_abs_det_jac_vjp(cov, w), o.gp_rfft2_log_abs_det_jac_vjp(cov, w))
    # per-lattice cov (batched) agrees with the shared one
    covb = np.stack([cov, cov * 1.5, cov + 0.25])
    lpb, gyb, gcb = fft.gp_rfft2_lpdf_vjp(y, loc, covb)
    for i in range(3):
        l_i, gy_i, gc_i = o.gp_rfft2_lpdf_vjp(y[i], loc, covb[i])
        close(lpb[i], l_i)
        close(gyb[i], gy_i)
        close(gcb[i], gc_i)


@pytest.mark.parametrize("n", [4, 5, 7, 8, 100, 128, 134, 176])
def test_fft1_gradients_vs_oracle(fft, n):
    rng = np.random.default_rng(8000 + n)
    y = rng.normal(size=(4, n))
    loc = rng.normal(size=n)
    cov = np.exp(rng.normal(size=n // 2 + 1))
    z = rng.normal(size=(4, n))
    bar = rng.normal(size=(4, n))
    lp, gy, gc = fft.gp_rfft_lpdf_vjp(y, loc, cov)
    lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y, loc, cov)
    close(lp, lp_o)
    close(gy, gy_o)
    close(gc, gc_o)
    zb, cb = fft.gp_inv_rfft_vjp(z, cov, bar)
    zb_o, _, cb_o = o.gp_inv_rfft_vjp(z, loc, cov, bar)
    close(zb, zb_o)
    close(cb, cb_o)
    close(fft.gp_rfft_log_abs_det_jac_vjp(cov, n), o.gp_rfft_log_abs_det_jac_vjp(cov, n))


def test_kernel_gradients_vs_oracle(fft):
    rng = np.random.default_rng(9000)
    for (h, w) in [(7, 9), (8, 10), (35, 60)]:
        sigma, ls, per = 1.3, np.array([0.7, 1.1]), np.array([5.0, 7.0])
        cbar = rng.normal(size=(h, w // 2 + 1))
        for nu in (None, 1.5, 2.5):
            if nu is None:
                got = fft.gp_periodic_exp_quad_cov_rfft2_vjp(h, w, sigma, ls, per, cbar)
                want = o.gp_periodic_exp_quad_cov_rfft2_vjp(h, w, sigma, ls, per, cbar)
            else:
                got = fft.gp_periodic_matern_cov_rfft2_vjp(nu, h, w, sigma, ls, per, cbar)
                want = o.gp_periodic_matern_cov_rfft2_vjp(nu, h, w, sigma, ls, per, cbar)
            scale = max(np.max(np.abs(np.ravel(x))) for x in want)
            for a, b in zip(got, want):
                close(a, b, scale=scale)
        cbar1 = rng.normal(size=w // 2 + 1)
        got = fft.gp_periodic_exp_quad_cov_rfft_vjp(w, sigma, 0.7, 5.0, cbar1)
        want = o.gp_periodic_exp_quad_cov_rfft_vjp(w, sigma, 0.7, 5.0, cbar1)
        for a, b in zip(got, want):
            close(a, b, scale=max(abs(float(x)) for x in want))
        got = fft.gp_periodic_matern_cov_rfft_vjp(1.5, w, sigma, 0.7, 5.0, cbar1)
        want = o.gp_periodic_matern_cov_rfft_vjp(1.5, w, sigma, 0.7, 5.0, cbar1)
        for a, b in zip(got, want):
            close(a, b, scale=max(abs(float(x)) for x in want))


def test_rejects(fft):
    """Stan `reject` paths (fft1.stan:19-21, fft2.stan:21-23, util.stan:274-276) surface as exceptions."""
    with pytest.raises(ValueError, match="not positive-definite"):
        fft.gp_evaluate_rfft_scale(np.array([1.0, -0.5, 1.0]), 4)
    with pytest.raises(ValueError, match="minimum eigenvalue is -6"):
        fft.gp_evaluate_rfft2_scale(np.array([[1.0, -1.0], [1.0, 1.0], [1.0, 1.0], [1.0, 1.0]]), 3)
    bad = np.ones((4, 4))
    bad[3, 0] = -1e-3  # a row of the edge column the forward transform never reads still rejects
    with pytest.raises(ValueError, match="not positive-definite"):
        fft.gp_rfft2_lpdf(np.zeros((4, 6)), np.zeros((4, 6)), bad)
    with pytest.raises(ValueError, match="expected complex vector with 3 elements but got 4"):
        fft.expand_rfft(np.zeros(4, dtype=complex), 4)
    # c == 0 passes the strict `< 0` check and yields -inf/NaN downstream (quirk 1 of SURVEY 8a)
    out = fft.gp_rfft_log_abs_det_jac(np.array([1.0, 0.0, 1.0]), 4)
    assert np.isinf(float(out))


def test_host_entry_points_col_major(fft):
    """The host-pointer ABI the Stan header binds: column-major (Eigen) matrices in and out."""
    import ctypes

    from gptools_amd import _lib

    lib = _lib.load()
    ctx = fft.get_context()
    rng = np.random.default_rng(11)
    # (the last two are large enough for the pinned, chunked staging ring: one exact chunk, and chunks with a remainder)
    for (h, w, dim1) in [(5, 8, 0), (6, 7, 0), (35, 60, 0), (1, 9, 1), (1, 128, 1), (1024, 1024, 0), (1100, 1000, 0)]:
        wf = w // 2 + 1
        y = rng.normal(size=(h, w))
        loc = rng.normal(size=(h, w))
        cov = np.exp(rng.normal(size=(h, wf)))
        F = lambda a: np.asfortranarray(a)  # noqa: E731
        yF, locF, covF = F(y), F(loc), F(cov)
        lp = np.zeros(1)
        gy = np.zeros((h, w), order="F")
        gc = np.zeros((h, wf), order="F")
        P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        rc = lib.gpt_gp_rfft2_lpdf_host(ctx.handle, _lib.GPT_COL_MAJOR, P(yF), P(locF), P(covF), h, w, dim1, P(lp),
                                        P(gy), P(gc))
        assert rc == 0, ctx.error()
        if dim1:
            lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y[0], loc[0], cov[0])
            gy_o, gc_o = gy_o[None], gc_o[None]
        else:
            lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y, loc, cov)
        close(lp[0], lp_o)
        close(gy, gy_o)
        close(gc, gc_o)
        z = np.zeros((h, w), order="F")
        rc = lib.gpt_gp_rfft2_host(ctx.handle, _lib.GPT_COL_MAJOR, P(yF), P(locF), P(covF), h, w, dim1, P(z))
        assert rc == 0, ctx.error()
        close(z, o.gp_rfft(y[0], loc[0], cov[0])[None] if dim1 else o.gp_rfft2(y, loc, cov))
        f = np.zeros((h, w), order="F")
        rc = lib.gpt_gp_inv_rfft2_host(ctx.handle, _lib.GPT_COL_MAJOR, P(yF), P(locF), P(covF), h, w, dim1, P(f))
        assert rc == 0, ctx.error()
        close(f, o.gp_inv_rfft(y[0], loc[0], cov[0])[None] if dim1 else o.gp_inv_rfft2(y, loc, cov))
        # real-FFT layer and layout helpers on host pointers
        if not dim1:
            zc = np.zeros((h, wf), dtype=complex, order="F")
            assert lib.gpt_rfft2_host(ctx.handle, _lib.GPT_COL_MAJOR, P(yF), h, w, P(zc)) == 0, ctx.error()
            close(zc, o.rfft2(y))
            back = np.zeros((h, w), order="F")
            assert lib.gpt_inv_rfft2_host(ctx.handle, _lib.GPT_COL_MAJOR, P(zc), h, w, P(back)) == 0, ctx.error()
            close(back, y)
            un = np.zeros((h, w), order="F")
            assert lib.gpt_gp_unpack_rfft2_host(ctx.handle, _lib.GPT_COL_MAJOR, P(zc), h, w, 0, P(un)) == 0
            assert np.array_equal(un, o.gp_unpack_rfft2(np.ascontiguousarray(zc), w))
            pk = np.zeros((h, wf), dtype=complex, order="F")
            assert lib.gpt_gp_pack_rfft2_host(ctx.handle, _lib.GPT_COL_MAJOR, P(un), h, w, 0, P(pk)) == 0
            assert np.array_equal(pk, o.gp_pack_rfft2(un))
            sc = np.zeros((h, wf), order="F")
            assert lib.gpt_gp_evaluate_rfft2_scale_host(ctx.handle, _lib.GPT_COL_MAJOR, P(covF), h, w, 0, P(sc)) == 0
            close(sc, o.gp_evaluate_rfft2_scale(cov, w))
        else:
            ex = np.zeros(w, dtype=complex)
            half = np.ascontiguousarray(o.rfft(y[0]))
            assert lib.gpt_expand_rfft_host(ctx.handle, P(half), w, P(ex)) == 0, ctx.error()
            close(ex, np.fft.fft(y[0]))
        # reject -> status 1 and the reference's message
        covF[0, 0] = -1.0
        rc = lib.gpt_gp_rfft2_lpdf_host(ctx.handle, _lib.GPT_COL_MAJOR, P(yF), P(locF), P(covF), h, w, dim1, P(lp),
                                        None, None)
        assert rc == _lib.GPT_ERR_NOT_PD
        assert "not positive-definite" in ctx.error()
        assert lib.gpt_last_min_eig(ctx.handle) < 0


@pytest.mark.parametrize("shape", [(35, 60), (110, 110), (256, 256), (512, 256), (1, 176), (1, 8192), (6, 5000)],
                         ids=lambda s: f"{s[0]}x{s[1]}")
def test_host_entry_points_row_major_small_calls(fft, shape):
    """Row-major host arrays through the copy-free small-call route of the host ABI: the transform kernels of every path
    (one workgroup per lattice, generic, power-of-two three-pass, any-length) read their inputs from, and write their
    results to, the pinned device-mapped bounce area directly. Twice in a row (the area is reused)."""
    import ctypes

    from gptools_amd import _lib

    lib = _lib.load()
    ctx = fft.get_context()
    h, w = shape
    dim1 = int(h == 1)
    wf = w // 2 + 1
    rng = np.random.default_rng(13 * h + w)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    for rep in range(2):
        y, loc, z = rng.normal(size=(h, w)), rng.normal(size=(h, w)), rng.normal(size=(h, w))
        cov = np.exp(0.3 * rng.normal(size=(h, wf)))
        lp, gy, gc = np.zeros(1), np.zeros((h, w)), np.zeros((h, wf))
        rc = lib.gpt_gp_rfft2_lpdf_host(ctx.handle, _lib.GPT_ROW_MAJOR, P(y), P(loc), P(cov), h, w, dim1, P(lp), P(gy), P(gc))
        assert rc == 0, ctx.error()
        if dim1:
            lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y[0], loc[0], cov[0])
            gy_o, gc_o = gy_o[None], gc_o[None]
        else:
            lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y, loc, cov)
        close(lp[0], lp_o)
        close(gy, gy_o)
        close(gc, gc_o)
        out = np.zeros((h, w))
        assert lib.gpt_gp_rfft2_host(ctx.handle, _lib.GPT_ROW_MAJOR, P(y), P(loc), P(cov), h, w, dim1, P(out)) == 0, ctx.error()
        close(out, o.gp_rfft(y[0], loc[0], cov[0])[None] if dim1 else o.gp_rfft2(y, loc, cov))
        assert lib.gpt_gp_inv_rfft2_host(ctx.handle, _lib.GPT_ROW_MAJOR, P(z), P(loc), P(cov), h, w, dim1, P(out)) == 0, ctx.error()
        close(out, o.gp_inv_rfft(z[0], loc[0], cov[0])[None] if dim1 else o.gp_inv_rfft2(z, loc, cov))
        # value only, no loc
        assert lib.gpt_gp_rfft2_lpdf_host(ctx.handle, _lib.GPT_ROW_MAJOR, P(y), None, P(cov), h, w, dim1, P(lp), None, None) == 0
        zero = np.zeros((h, w))
        close(lp[0], (o.gp_rfft_lpdf(y[0], zero[0], cov[0]) if dim1 else o.gp_rfft2_lpdf(y, zero, cov)))


# ---- three-pass paths (lattices too large for one workgroup's LDS) ----------------------------------
BIG_SHAPES = [(256, 256), (256, 512), (512, 256), (1024, 256), (2048, 512), (512, 2048), (8192, 256), (256, 8192),
              (256, 16384),  # every line length of the power-of-two kernels
              (1024, 1024), (2048, 1024), (1024, 2048), (8192, 1024), (1024, 4096), (4096, 2048),  # sub-line kernels
              (110, 110), (90, 125), (128, 128), (301, 200), (5, 3000), (2500, 6),  # generic three-pass path
              # ... its column-tile geometries (gpt_generic.hip make_geo): 4 columns x 512 threads, 2 columns x 1024 threads in
              # nine tiles (tile numbering by XCD with a ragged last eighth), 4 of 6 possible columns, 8 columns x 256 threads
              (1000, 40), (3000, 36), (600, 50), (192, 200), (160, 1000)]


@pytest.mark.parametrize("shape", BIG_SHAPES, ids=lambda s: f"{s[0]}x{s[1]}")
def test_three_pass_path_vs_oracle(fft, shape, monkeypatch):
    """Every lattice function on the three-pass kernels, batch of 2, against the oracle."""
    # (launches of a few small power-of-two lattices would take the generic kernels, gpt_api.hip few_pow2_lattices: this test
    # is about each path's own kernels; the re-routed launches are test_few_small_power_of_two_lattices below)
    monkeypatch.setenv("GPT_POW2_SMALL_TO_GENERIC", "0")
    h, w = shape
    assert fft.path_for_shape(h, w) in (1, 2)
    rng = np.random.default_rng(100 * h + w)
    y = rng.normal(size=(2, h, w))
    loc = rng.normal(size=(h, w))
    cov = o.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.3, [h / 40.0, w / 30.0], [float(h), float(w)]) + 1e-3
    cov = cov * np.exp(0.1 * rng.normal(size=cov.shape))  # break the row symmetry
    z = rng.normal(size=(2, h, w))
    bar = rng.normal(size=(2, h, w))
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


@pytest.mark.parametrize("shape", [(256, 256), (256, 512), (512, 256), (1024, 256), (1024, 1024), (256, 4096)],
                         ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("batch", [1, 3])
def test_few_small_power_of_two_lattices(fft, shape, batch, monkeypatch):
    """A launch of one or a few small power-of-two lattices runs on the generic path's kernels (lower latency; gpt_api.hip
    few_pow2_lattices), GPT_POW2_SMALL_TO_GENERIC=0 keeps it on the power-of-two kernels: both against the oracle."""
    h, w = shape
    rng = np.random.default_rng(5 * h + w + batch)
    y = rng.normal(size=(batch, h, w))
    loc = rng.normal(size=(h, w))
    cov = np.exp(0.3 * rng.normal(size=(batch, h, w // 2 + 1)))
    z = rng.normal(size=(batch, h, w))
    want = [o.gp_rfft2_lpdf_vjp(y[i], loc, cov[i]) for i in range(batch)]
    f_o = o.gp_inv_rfft2(z[batch - 1], loc, cov[batch - 1])
    z_o = o.gp_rfft2(y[0], loc, cov[0])
    for knob in ("1", "0"):
        monkeypatch.setenv("GPT_POW2_SMALL_TO_GENERIC", knob)
        lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, loc, cov)
        for i in range(batch):
            close(lp[i], want[i][0])
            close(gy[i], want[i][1])
            close(gc[i], want[i][2])
        close(fft.gp_inv_rfft2(z, loc, cov)[batch - 1], f_o)
        close(fft.gp_rfft2(y, loc, cov)[0], z_o)


def test_many_large_one_workgroup_lattices_take_the_generic_path(fft):
    """Lattices that fill more than half a CU's LDS on the one-workgroup path go to the generic three-pass path when a
    launch holds 768 or more of them (gpt_api.hip many_large_small_lattices): same results as the same lattices in small
    launches, and as the oracle."""
    h, w, b = 72, 72, 800
    rng = np.random.default_rng(72800)
    y = rng.normal(size=(b, h, w))
    loc = rng.normal(size=(h, w))
    cov = np.exp(0.3 * rng.normal(size=(b, h, w // 2 + 1)))
    lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, loc, cov)
    for lo in (0, 400, 790):
        lp_s, gy_s, gc_s = fft.gp_rfft2_lpdf_vjp(y[lo:lo + 10], loc, cov[lo:lo + 10])  # (one workgroup per lattice)
        close(lp[lo:lo + 10], lp_s.cpu())
        close(gy[lo:lo + 10], gy_s.cpu())
        close(gc[lo:lo + 10], gc_s.cpu())
    for i in (0, 799):
        lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y[i], loc, cov[i])
        close(lp[i], lp_o)
        close(gy[i], gy_o)
        close(gc[i], gc_o)
    z = rng.normal(size=(b, h, w))
    f = fft.gp_inv_rfft2(z, loc, cov)
    close(f[:4], fft.gp_inv_rfft2(z[:4], loc, cov[:4]).cpu())
    close(f[799], o.gp_inv_rfft2(z[799], loc, cov[799]))
    close(fft.gp_rfft2(y, loc, cov)[795:], o.gp_rfft2(y[795:], loc, cov[795:]))


def test_headline_shape_4096(fft):
    """BASELINE config 4 (4096 x 4096): value and gradients against the oracle on one lattice, plus
    size-independent properties: transform round trip and linearity of the gradient in (y - loc)."""
    import torch

    h = w = 4096
    rng = np.random.default_rng(4096)
    cov = o.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.3, [8.0, 12.0], [float(h), float(w)]) + 1e-3
    loc = rng.normal(size=(h, w))
    z = rng.normal(size=(h, w))
    dev = torch.device("cuda:0")
    cov_d, loc_d, z_d = (torch.as_tensor(x, device=dev) for x in (cov, loc, z))
    y_d = fft.gp_inv_rfft2(z_d, loc_d, cov_d)
    # round trip: gp_rfft2(gp_inv_rfft2(z)) == z (test_fft.py:114-126 at full size)
    close(fft.gp_rfft2(y_d, loc_d, cov_d), z, rtol=1e-9)
    lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y_d, loc_d, cov_d)
    y = y_d.cpu().numpy()
    close(y, o.gp_inv_rfft2(z, loc, cov))
    lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y, loc, cov)
    close(lp, lp_o)
    close(gy, gy_o)
    close(gc, gc_o)
    # lpdf = -1/2 |z|^2 - n/2 log(2 pi) + log|J| with z the whitened coefficients
    ladj = fft.gp_rfft2_log_abs_det_jac(cov_d, w)
    want = -0.5 * float((z_d * z_d).sum()) - 0.5 * h * w * np.log(2 * np.pi) + float(ladj)
    close(lp, want, rtol=1e-9)
    # the gradient w.r.t. y is linear in (y - loc): grad(loc + 2 (y - loc)) == 2 grad(y)
    _, gy2, _ = fft.gp_rfft2_lpdf_vjp(loc_d + 2.0 * (y_d - loc_d), loc_d, cov_d, need_cov_grad=False)
    close(gy2, 2.0 * gy.cpu().numpy(), rtol=1e-9)


def test_subline_path_variants(fft):
    """The sub-line kernels (fused value + gradient at heights >= 512) with everything the plain cases leave out:
    a spectrum per lattice, no loc, no d/dcov, more lattices than one workspace chunk holds, and the reject of a
    negative spectrum entry in a row above H/2 of an edge column (never read by the transform, still checked)."""
    h, w = 1024, 2048
    b = 5
    rng = np.random.default_rng(77)
    y = rng.normal(size=(b, h, w))
    loc = rng.normal(size=(h, w))
    base = o.gp_periodic_matern_cov_rfft2(2.5, h, w, 0.9, [h / 50.0, w / 35.0], [float(h), float(w)]) + 1e-3
    cov = base * np.exp(0.1 * rng.normal(size=(b,) + base.shape))
    lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, loc, cov)
    for i in range(b):
        lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y[i], loc, cov[i])
        close(lp[i], lp_o)
        close(gy[i], gy_o)
        close(gc[i], gc_o)
    lp2, gy2, gc2 = fft.gp_rfft2_lpdf_vjp(y, None, cov, need_cov_grad=False)
    assert gc2 is None
    lp_o, gy_o, _ = o.gp_rfft2_lpdf_vjp(y[b - 1], np.zeros((h, w)), cov[b - 1])
    close(lp2[b - 1], lp_o)
    close(gy2[b - 1], gy_o)
    # 40 lattices of 2048 x 1024 (16 MiB of packed spectrum each): several workspace chunks
    h2, w2 = 2048, 1024
    y2 = rng.normal(size=(40, h2, w2))
    cov2 = o.gp_periodic_matern_cov_rfft2(1.5, h2, w2, 1.1, [20.0, 15.0], [float(h2), float(w2)]) + 1e-3
    lp3, gy3, gc3 = fft.gp_rfft2_lpdf_vjp(y2, None, cov2)
    for i in (0, 7, 39):
        lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y2[i], np.zeros((h2, w2)), cov2)
        close(lp3[i], lp_o)
        close(gy3[i], gy_o)
        close(gc3[i], gc_o)
    bad = cov.copy()
    bad[3, h - 2, 0] = -1.0
    with pytest.raises(ValueError, match="not positive-definite"):
        fft.gp_rfft2_lpdf_vjp(y, loc, bad)
    bad = cov.copy()
    bad[1, 5, 77] = -2.0
    with pytest.raises(ValueError, match="not positive-definite"):
        fft.gp_rfft2_lpdf_vjp(y, loc, bad)


@pytest.mark.parametrize("shape", [(1024, 1024), (512, 2048), (35, 60), (100, 100), (128,),
                                   (300, 210), (1000, 40), (121, 130), (2500, 6),  # (generic path: lean / full kernels)
                                   (256, 256), (256, 1024)], ids=str)  # (no sub-line column kernel: generic path's kernels)
@pytest.mark.parametrize("kind,nu", [("matern", 1.5), ("matern", 2.5), ("matern", 0.9), ("expquad", 0.0)])
def test_kernel_fused_lpdf(fft, shape, kind, nu):
    """gp_rfft2_lpdf over gp_periodic_*_cov_rfft2 + nugget without the spectrum in memory (power-of-two lattices of
    >= 512 rows: evaluated inside the column pass; other shapes: composed), against the oracle's composition of the
    three reference functions and the chain rule."""
    dim1 = len(shape) == 1
    rng = np.random.default_rng(11)
    b = 2
    y = rng.normal(size=(b,) + shape)
    loc = rng.normal(size=shape)
    sigma = np.array([1.3, 0.7])
    eps = 1e-3
    if dim1:
        n = shape[0]
        ls, per = np.array([n / 12.0, n / 20.0]), np.array([float(n), 1.5 * n])
        lp, gy, (sb, lb, pb) = fft.gp_rfft_lpdf_kernel_vjp(y, loc, kind, nu, sigma, ls, per, nugget=eps)
    else:
        h, w = shape
        ls = np.array([[h / 40.0, w / 30.0], [h / 25.0, w / 50.0]])
        per = np.array([[float(h), float(w)], [1.2 * h, 0.9 * w]])
        lp, gy, (sb, lb, pb) = fft.gp_rfft2_lpdf_kernel_vjp(y, loc, kind, nu, sigma, ls, per, nugget=eps)
    for i in range(b):
        if dim1:
            cov = (o.gp_periodic_exp_quad_cov_rfft(n, sigma[i], ls[i], per[i]) if kind == "expquad"
                   else o.gp_periodic_matern_cov_rfft(nu, n, sigma[i], ls[i], per[i])) + eps
            lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y[i], loc, cov)
            bars = (o.gp_periodic_exp_quad_cov_rfft_vjp(n, sigma[i], ls[i], per[i], gc_o) if kind == "expquad"
                    else o.gp_periodic_matern_cov_rfft_vjp(nu, n, sigma[i], ls[i], per[i], gc_o))
        else:
            cov = (o.gp_periodic_exp_quad_cov_rfft2(h, w, sigma[i], ls[i], per[i]) if kind == "expquad"
                   else o.gp_periodic_matern_cov_rfft2(nu, h, w, sigma[i], ls[i], per[i])) + eps
            lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y[i], loc, cov)
            bars = (o.gp_periodic_exp_quad_cov_rfft2_vjp(h, w, sigma[i], ls[i], per[i], gc_o) if kind == "expquad"
                    else o.gp_periodic_matern_cov_rfft2_vjp(nu, h, w, sigma[i], ls[i], per[i], gc_o))
        close(lp[i], lp_o)
        close(gy[i], gy_o)
        # the parameter gradients are sums of ~n terms of both signs: tolerance relative to the sum of magnitudes
        scale = float(np.abs(gc_o * cov).sum())
        for got, want, p in zip((sb[i], lb[i], pb[i]), bars, (sigma[i], ls[i], per[i])):
            np.testing.assert_allclose(np.asarray(got.cpu()), want, rtol=1e-9,
                                       atol=1e-10 * scale / float(np.min(np.abs(p))))
