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


def _generic_shapes():
    """Shapes of the generic three-pass path (gpt_generic.hip): column tiles of 16 / 8 / 4 / 2 / 1 columns with ragged last
    tiles, tile counts that are and are not multiples of eight (the XCD numbering), workgroups of 128 ... 1024 threads,
    5-smooth lines (lean kernels) mixed with lines that have factors 7, 11, 13, 16 or a larger prime."""
    rng = np.random.default_rng(20261111)
    out = [(1200, 36), (1201, 36), (2400, 18), (4050, 10), (36, 1200), (150, 3000), (1024, 90), (640, 650), (125, 729),
           (343, 130), (169, 176), (121, 1331), (720, 34), (34, 720), (1999, 14)]
    def smooth(n, top):  # largest prime factor <= top
        for f in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43):
            while f <= top and n % f == 0:
                n //= f
        return n == 1

    lens = [n for n in range(100, 1300) if smooth(n, 43)]  # (a prime factor above 127 sends a shape to the any-length path)
    while len(out) < 60:
        h, w = int(rng.choice(lens)), int(rng.choice(lens))
        if h * w <= 900_000:
            out.append((h, w))
    return out


@pytest.mark.parametrize("shape", _generic_shapes(), ids=lambda s: f"{s[0]}x{s[1]}")
def test_generic_path_sweep(fft, shape):
    h, w = shape
    if fft.path_for_shape(h, w) != 2:
        pytest.skip("not a generic-path shape")
    rng = np.random.default_rng(11 * h + 17 * w)
    nb = 3
    cov = np.exp(0.4 * rng.normal(size=(nb, h, w // 2 + 1))) + 0.05  # one spectrum per lattice
    y, loc, z = rng.normal(size=(nb, h, w)), rng.normal(size=(h, w)), rng.normal(size=(nb, h, w))
    lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, loc, cov)
    for b in range(nb):
        lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y[b], loc, cov[b])
        close(lp[b], lp_o)
        close(gy[b], gy_o)
        close(gc[b], gc_o)
    close(fft.gp_inv_rfft2(z, loc, cov)[nb - 1], o.gp_inv_rfft2(z[nb - 1], loc, cov[nb - 1]))
    close(fft.gp_rfft2(y, loc, cov)[nb - 1], o.gp_rfft2(y[nb - 1], loc, cov[nb - 1]))
