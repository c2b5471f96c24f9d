"""
Cross-chain batching on the GPU: chains = Python threads (ctypes releases the GIL while a
chain waits inside the library), every chain its own lattice and spectrum, against the oracle; chains leaving,
a chain with a non-positive-definite spectrum, column-major (Stan) matrices, the 1-D functions.
"""
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def _close(a, b):
    b = np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=RTOL, atol=RTOL * max(float(np.abs(b).max()), 1e-300))


def _run_chains(nchains, fn):
    errors = []

    def main(c):
        try:
            fn(c)
        except Exception as exc:  # pragma: no cover - surfaced below
            errors.append((c, exc))

    threads = [threading.Thread(target=main, args=(c,)) for c in range(nchains)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a chain is stuck inside the batcher"
    if errors:
        raise errors[0][1]


@pytest.mark.parametrize("col_major", [False, True])
def test_batched_chains_match_oracle(col_major):
    import gp_oracle as o
    from gptools_amd import batch

    h, w, nchains, rounds = 35, 60, 8, 3
    rng = np.random.default_rng(5)
    base = o.gp_periodic_matern_cov_rfft2(1.5, h, w, 1.2, [h / 9.0, w / 7.0], [float(h), float(w)]) + 1e-3
    y = rng.normal(size=(rounds, nchains, h, w))
    loc = rng.normal(size=(nchains, h, w))
    cov = base * np.exp(0.1 * rng.normal(size=(rounds, nchains) + base.shape))
    b = batch.Batcher(nchains, h, w, col_major=col_major)
    out = {}

    def chain(c):
        for r in range(rounds):
            want_grad = not (c == 1 and r == 1)  # one value-only call rides along with the gradient calls
            out[(c, r)] = b.lpdf(c, y[r, c], loc[c] if c != 2 else None, cov[r, c], grad=want_grad)

    _run_chains(nchains, chain)
    assert b.launches == rounds  # one fused launch per round, not one per chain
    for (c, r), res in out.items():
        lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y[r, c], loc[c] if c != 2 else np.zeros((h, w)), cov[r, c])
        if c == 1 and r == 1:
            _close(res, lp_o)
            continue
        _close(res[0], lp_o)
        _close(res[1], gy_o)
        _close(res[2], gc_o)
    b.close()


def test_chains_leave_and_reject():
    import gp_oracle as o
    from gptools_amd import batch
    from gptools_amd.fft import NotPositiveDefinite

    n, nchains = 176, 6  # 1-D functions, n = 16 * 11 (docs/padding sweep)
    rng = np.random.default_rng(6)
    cov = o.gp_periodic_matern_cov_rfft(2.5, n, 0.8, 12.0, float(n)) + 1e-3
    y = rng.normal(size=(4, nchains, n))
    loc = rng.normal(size=n)
    b = batch.Batcher(nchains, None, n)
    got, rejected = {}, []

    def chain(c):
        rounds = 2 if c < 2 else 4  # chains 0 and 1 finish early and leave
        for r in range(rounds):
            cc = cov.copy()
            if c == 3 and r == 1:
                cc[5] = -1.0  # this chain's proposal is rejected; the others are unaffected
            try:
                got[(c, r)] = b.lpdf(c, y[r, c], loc, cc, grad=True, grad_cov=False)
            except NotPositiveDefinite:
                rejected.append((c, r))
        b.leave(c)

    _run_chains(nchains, chain)
    assert rejected == [(3, 1)]
    assert b.launches == 4
    for (c, r), (lp, gy, gc) in got.items():
        lp_o, gy_o, _ = o.gp_rfft_lpdf_vjp(y[r, c], loc, cov)
        _close(lp, lp_o)
        _close(gy, gy_o)
        assert gc is None
    with pytest.raises(ValueError):
        b.lpdf(0, y[0, 0], loc, cov)  # slot 0 has left
    b.close()


def test_timeout_when_a_chain_died_without_leaving():
    """A chain that ends without `leave` must not block the others for ever: with a timeout the waiting call gives up
    with TimeoutError, and once the dead chain's slot has left, the same slot evaluates normally."""
    import gp_oracle as o
    from gptools_amd import batch

    n = 144
    rng = np.random.default_rng(8)
    cov = o.gp_periodic_exp_quad_cov_rfft(n, 1.1, 9.0, float(n)) + 1e-3
    y = rng.normal(size=n)
    b = batch.Batcher(2, None, n, timeout=0.3)
    with pytest.raises(TimeoutError, match="did not call"):
        b.lpdf(0, y, None, cov)  # slot 1 never calls
    assert b.launches == 0
    b.leave(1)
    lp, gy, gc = b.lpdf(0, y, None, cov)
    lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y, np.zeros(n), cov)
    _close(lp, lp_o)
    _close(gy, gy_o)
    _close(gc, gc_o)
    b.close()
