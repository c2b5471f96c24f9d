"""Small calls of the host-pointer ABI (what a Stan chain makes, thousands of times): the one-workgroup kernel publishes
its own completion into pinned memory and the host spins on that instead of synchronising the stream (gpt_ctx::done_*,
gpt_api_host.hip), and small column-major matrices change layout on the CPU inside the copy that happens anyway. Many
calls in a row with fresh inputs every time -- a flag seen too early, or a sequence number off by one, shows up as the
previous call's results -- mixed with calls that take the ordinary synchronisation (larger lattices, other entry points),
every result against the oracle."""
import ctypes

import numpy as np
import pytest

import gp_oracle as o

pytestmark = pytest.mark.gpu

RTOL = 1e-10  # as tests/test_gpu_parity.py


def close(a, b):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(1.0, float(np.max(np.abs(b)))) if b.size else 1.0
    assert np.max(np.abs(a - b)) <= RTOL * scale, (np.max(np.abs(a - b)), scale)


@pytest.fixture(scope="module")
def fft():
    from gptools_amd import fft as f
    return f


@pytest.mark.parametrize("layout_name", ["row", "col"])
@pytest.mark.parametrize("shape", [(1, 100), (1, 128), (35, 60), (64, 64), (7, 9)], ids=lambda s: f"{s[0]}x{s[1]}")
def test_many_small_host_calls_in_a_row(fft, shape, layout_name):
    from gptools_amd import _lib

    lib = _lib.load()
    ctx = fft.get_context()
    layout = _lib.GPT_ROW_MAJOR if layout_name == "row" else _lib.GPT_COL_MAJOR
    h, w = shape
    dim1 = int(h == 1)
    wf = w // 2 + 1
    rng = np.random.default_rng(1000 * h + w)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    order = "C" if layout == _lib.GPT_ROW_MAJOR else "F"
    arr = lambda a: np.array(a, order=order)  # noqa: E731  (the caller's layout)
    for rep in range(60):
        y, loc = rng.normal(size=(h, w)), rng.normal(size=(h, w))
        cov = np.exp(0.3 * rng.normal(size=(h, wf)))
        lp = np.full(1, np.nan)
        gy, gc = np.full((h, w), np.nan, order=order), np.full((h, wf), np.nan, order=order)
        ya, la, ca = arr(y), arr(loc), arr(cov)
        rc = lib.gpt_gp_rfft2_lpdf_host(ctx.handle, layout, P(ya), P(la), P(ca), h, w, dim1, P(lp), P(gy), P(gc))
        assert rc == 0, ctx.error()
        if dim1:
            lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y[0], loc[0], cov[0])
            gy_o, gc_o = gy_o[None], gc_o[None]
        else:
            lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y, loc, cov)
        close(lp[0], lp_o)
        close(gy, gy_o)
        close(gc, gc_o)
        if rep % 7 == 3:  # the transform pair, between log-density calls
            out = np.full((h, w), np.nan, order=order)
            assert lib.gpt_gp_rfft2_host(ctx.handle, layout, P(ya), P(la), P(ca), h, w, dim1, P(out)) == 0, ctx.error()
            close(out, o.gp_rfft(y[0], loc[0], cov[0])[None] if dim1 else o.gp_rfft2(y, loc, cov))
            z = arr(rng.normal(size=(h, w)))
            assert lib.gpt_gp_inv_rfft2_host(ctx.handle, layout, P(z), P(la), P(ca), h, w, dim1, P(out)) == 0, ctx.error()
            close(out, o.gp_inv_rfft(np.asarray(z)[0], loc[0], cov[0])[None] if dim1 else o.gp_inv_rfft2(np.asarray(z), loc, cov))
        if rep % 11 == 5:  # a call that waits the ordinary way (three-pass path), then back
            H2, W2 = 256, 256
            y2, l2 = rng.normal(size=(H2, W2)), rng.normal(size=(H2, W2))
            c2 = np.exp(0.3 * rng.normal(size=(H2, W2 // 2 + 1)))
            lp2 = np.full(1, np.nan)
            assert lib.gpt_gp_rfft2_lpdf_host(ctx.handle, _lib.GPT_ROW_MAJOR, P(y2), P(l2), P(c2), H2, W2, 0, P(lp2), None,
                                              None) == 0, ctx.error()
            close(lp2[0], o.gp_rfft2_lpdf(y2, l2, c2))
        if rep % 13 == 6:  # the reference's reject still arrives through the flag path (min_eig sits beside the results)
            bad = cov.copy()
            bad[0, 0] = -1.0
            rc = lib.gpt_gp_rfft2_lpdf_host(ctx.handle, layout, P(ya), P(la), P(arr(bad)), h, w, dim1, P(lp), None, None)
            assert rc == _lib.GPT_ERR_NOT_PD
            assert "not positive-definite" in ctx.error()


@pytest.mark.parametrize("shape", [(1, 100), (35, 60), (64, 64), (256, 256), (120, 150)], ids=lambda s: f"{s[0]}x{s[1]}")
@pytest.mark.parametrize("kind,nu", [("matern", 1.5), ("expquad", 0.0)])
def test_kernel_parameter_host_forms(fft, shape, kind, nu):
    """gpt_gp_rfft2_lpdf_kernel_host and gpt_gp_inv_rfft2_obs_lpdf_kernel_host (column-major, as the Stan header calls
    them): (sigma, length_scale, period) + nugget in, the density and every adjoint out in one call; several calls in a
    row with fresh inputs (one-launch kernels publish their own completion); against the oracle's composition of the
    reference functions."""
    from gptools_amd import _lib

    lib = _lib.load()
    ctx = fft.get_context()
    h, w = shape
    dim1 = int(h == 1)
    rng = np.random.default_rng(77 * h + w)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    F = lambda a: np.asfortranarray(a)  # noqa: E731
    eps = 1e-3
    for rep in range(8):
        y, loc = rng.normal(size=(h, w)), rng.normal(size=(h, w))
        sigma = float(rng.gamma(4.0, 0.3))
        if dim1:
            ls, per = float(rng.uniform(2.0, 6.0)), float(w)
            par = np.array([sigma, ls, per])
            cov = (o.gp_periodic_exp_quad_cov_rfft(w, sigma, ls, per) if kind == "expquad"
                   else o.gp_periodic_matern_cov_rfft(nu, w, sigma, ls, per)) + eps
            lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y[0], loc[0], cov)
            gy_o = gy_o[None]
            bars = (o.gp_periodic_exp_quad_cov_rfft_vjp(w, sigma, ls, per, gc_o) if kind == "expquad"
                    else o.gp_periodic_matern_cov_rfft_vjp(nu, w, sigma, ls, per, gc_o))
        else:
            ls, per = rng.uniform(1.5, 5.0, size=2), np.array([float(h), float(w)])
            par = np.concatenate([[sigma], ls, per])
            cov = (o.gp_periodic_exp_quad_cov_rfft2(h, w, sigma, ls, per) if kind == "expquad"
                   else o.gp_periodic_matern_cov_rfft2(nu, h, w, sigma, ls, per)) + eps
            lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y, loc, cov)
            bars = (o.gp_periodic_exp_quad_cov_rfft2_vjp(h, w, sigma, ls, per, gc_o) if kind == "expquad"
                    else o.gp_periodic_matern_cov_rfft2_vjp(nu, h, w, sigma, ls, per, gc_o))
        want_bar = np.concatenate([np.ravel(x) for x in bars])
        scale = float(np.abs(gc_o * cov).sum())
        lp, gy, pb = np.full(1, np.nan), np.full((h, w), np.nan, order="F"), np.full(par.size, np.nan)
        ya, la = F(y), F(loc)
        rc = lib.gpt_gp_rfft2_lpdf_kernel_host(ctx.handle, _lib.GPT_COL_MAJOR, int(kind == "matern"), nu, P(ya), P(la), P(par), eps, h, w,
                                               dim1, P(lp), P(gy), P(pb))
        assert rc == 0, ctx.error()
        close(lp[0], lp_o)
        close(gy, gy_o)
        np.testing.assert_allclose(pb, want_bar, rtol=1e-9, atol=1e-9 * scale / float(np.min(np.abs(par))))
        if dim1:
            continue
        # the non-centred model density from the same parameters
        oh, ow = h - h // 4, w - w // 3
        obs = rng.poisson(3.0, size=(oh, ow)).astype(float)
        obs[rng.random((oh, ow)) < 0.1] = -1.0
        z = rng.normal(size=(h, w))
        lp2, gl, gt = np.full(1, np.nan), np.full(1, np.nan), np.full(1, np.nan)
        gz, kb = np.full((h, w), np.nan, order="F"), np.full(5, np.nan)
        za, oa = F(z), F(obs)
        rc = lib.gpt_gp_inv_rfft2_obs_lpdf_kernel_host(ctx.handle, _lib.GPT_COL_MAJOR, o.OBS_NEG_BINOMIAL_2_LOG, int(kind == "matern"), nu,
                                                       P(za), None, 0.4, P(par), eps, P(oa), oh, ow, 2.2, 1, h, w, 0, P(lp2), P(gz), P(gl),
                                                       P(kb), P(gt))
        assert rc == 0, ctx.error()
        lpo, zbar, lsum, kbars, thbar = o.gp_inv_rfft2_obs_lpdf_kernel_vjp(o.OBS_NEG_BINOMIAL_2_LOG, z, np.full((h, w), 0.4), kind, nu, sigma,
                                                                           ls, per, obs, 2.2, nugget=eps)
        close(lp2[0], lpo)
        close(gz, zbar)
        close(gl[0], lsum)
        close(gt[0], thbar)
        cbar = o.gp_inv_rfft2_obs_lpdf_vjp(o.OBS_NEG_BINOMIAL_2_LOG, z, np.full((h, w), 0.4), cov, obs, 2.2)[5]
        np.testing.assert_allclose(kb, np.concatenate([np.ravel(x) for x in kbars]), rtol=1e-9,
                                   atol=1e-9 * float(np.abs(cbar * cov).sum()) / float(np.min(np.abs(par))))
