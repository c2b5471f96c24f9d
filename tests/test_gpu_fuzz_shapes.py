"""
Seeded sweep over lattice shapes across the boundaries between the kernel families (one workgroup per lattice /
power-of-two / generic / any-length): value and gradients of the log density and both transforms against the
oracle at relative 1e-10. The reference takes any shape (fft2.stan:13-192, fft1.stan:14-143); its own tests stop at small
even / odd pairs (test_stan_functions.py:152-408), so this is where odd corners of the path selection get exercised:
widths 2 and 3, heights 1 and 2, primes, lengths just below and above 4096 / 5120, shapes whose lines fit a workgroup
but whose lattice does not.
"""
import numpy as np
import pytest

import gp_oracle as o

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def close(actual, desired, rtol=RTOL):
    actual = actual.cpu().numpy() if hasattr(actual, "cpu") else np.asarray(actual)
    desired = np.asarray(desired)
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=rtol * max(float(np.max(np.abs(desired))), 1e-300))


def _shapes():
    rng = np.random.default_rng(20250611)
    out = [(2, 2), (1, 2), (2, 3), (3, 2), (1, 3), (2, 4096), (2, 4097), (3, 5119), (1, 5120), (1, 5121), (4096, 2), (4097, 3),
           (64, 80), (80, 64), (81, 81), (256, 255), (255, 256), (257, 256), (256, 258), (512, 384), (97, 101), (2, 8192),
           (8192, 2), (127, 254), (131, 262), (4095, 4), (4, 4095), (1, 4099), (1, 12289), (3, 16384), (1, 131071)]
    for _ in range(12):
        out.append((int(rng.integers(1, 40)), int(rng.integers(2, 7000))))
        out.append((int(rng.integers(2, 7000)), int(rng.integers(2, 40))))
        out.append((int(rng.integers(40, 420)), int(rng.integers(40, 420))))
    return out


@pytest.fixture(scope="module")
def fft(gpu_lib):
    from gptools_amd import fft as f

    return f


@pytest.mark.parametrize("shape", _shapes(), ids=lambda s: f"{s[0]}x{s[1]}")
def test_shape_sweep(fft, shape):
    h, w = shape
    assert fft.path_for_shape(h, w) >= 0
    rng = np.random.default_rng(7 * h + 13 * w)
    nb = 2
    cov = np.exp(0.4 * rng.normal(size=(h, w // 2 + 1))) + 0.05
    if h == 1:  # the 1-D functions (fft1.stan)
        n = w
        y, loc, z = rng.normal(size=(nb, n)), rng.normal(size=n), rng.normal(size=(nb, n))
        c1 = cov[0]
        lp, gy, gc = fft.gp_rfft_lpdf_vjp(y, loc, c1)
        lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y, loc, c1)
        close(lp, lp_o)
        close(gy, gy_o)
        close(gc, gc_o)
        close(fft.gp_inv_rfft(z, loc, c1), o.gp_inv_rfft(z, loc, c1))
        close(fft.gp_rfft(y, loc, c1), o.gp_rfft(y, loc, c1))
        return
    y, loc, z = rng.normal(size=(nb, h, w)), rng.normal(size=(h, w)), rng.normal(size=(nb, h, w))
    lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, loc, cov)
    lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y, loc, cov)
    close(lp, lp_o)
    close(gy, gy_o)
    close(gc, gc_o)
    close(fft.gp_inv_rfft2(z, loc, cov), o.gp_inv_rfft2(z, loc, cov))
    # (no round trip here: with a spectrum that is not symmetric in the rows the two transforms are not inverses of each
    # other on the edge columns, fft2.stan:65,127-128)
    close(fft.gp_rfft2(y, loc, cov), o.gp_rfft2(y, loc, cov))
    close(fft.rfft2(y), o.rfft2(y))
