"""
Generate the golden vectors under tests/golden/ from the REFERENCE's own NumPy mirror.

Run in the build container only (the reference checkout is not available on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py [/root/reference]

It imports ``gptools.util.fft`` / ``gptools.util.kernels`` from ``<reference>/python/util`` (numpy + scipy
only), evaluates them on seeded inputs and stores inputs + outputs as small ``.npz`` files. Nothing of
the reference's source is written anywhere: the fixtures hold numbers only.

Covered reference functions (python/util/gptools/util/fft/fft1.py, fft2.py, kernels.py) and the tests of
the reference they correspond to:
  * unpack/pack, scale, transform_rfft(2), transform_irfft(2), log_abs_det_jac, log_prob
    (python/util/tests/test_fft.py:32-126, python/stan/tests/test_stan_functions.py:152-408)
  * np.fft.rfft/irfft/rfft2/irfft2 pins of the raw transforms (test_stan_functions.py:152-177, 266-295)
  * dense multivariate-normal log-density for n <= 100 (test_fft.py:39-47,86-96)
  * ExpQuadKernel / MaternKernel(1.5|2.5).evaluate_rfft in 1-D and 2-D (test_stan_functions.py:410-513)
"""
import os
import sys

import numpy as np
from scipy import stats

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, os.path.join(REF, "python", "util"))
sys.dont_write_bytecode = True

from gptools.util import coordgrid, fft, kernels  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

SIZES_1D = [4, 5, 7, 8, 9, 10, 21, 22, 100, 128, 134, 144, 160, 176, 192]
SHAPES_2D = [(4, 6), (4, 7), (5, 6), (5, 7), (5, 8), (6, 7), (6, 8), (7, 9), (8, 10), (35, 60),
             (64, 64), (2, 2), (3, 2), (2, 3), (16, 12), (12, 16), (27, 25), (1, 8), (1, 7)]


def kernel_set(period):
    # Length scales are kept short relative to the period: the wrapped real-space kernels of variant
    # (b) stop being positive definite for long ones (their spectrum is a naive discretisation).
    return {
        "expquad": kernels.ExpQuadKernel(1.3, 0.06 * np.mean(period), period),
        "matern32": kernels.MaternKernel(1.5, 0.9, 0.05 * np.mean(period), period),
        "matern52": kernels.MaternKernel(2.5, 1.7, 0.04 * np.mean(period), period),
    }


def make_1d(n, rng):
    out = {"n": n}
    y, loc, z = rng.normal(size=(3, n))
    yb = rng.normal(size=(3, n))  # a batch sharing loc and cov
    out.update(y=y, loc=loc, z=z, yb=yb)
    out["rfft_y"] = np.fft.rfft(y)
    out["irfft_rfft_y"] = np.fft.irfft(out["rfft_y"], n)
    out["unpack_rfft_y"] = fft.fft1.unpack_rfft(out["rfft_y"], n)
    out["pack_z"] = fft.fft1.pack_rfft(z)
    out["expand_rfft_y"] = fft.expand_rfft(out["rfft_y"], n)
    for name, kernel in kernel_set(float(n)).items():
        # (a) analytic spectrum + nugget (how the docs models build cov_rfft)
        cov_a = kernel.evaluate_rfft(n) + 1e-3
        # (b) spectrum of the wrapped real-space kernel + nugget (how the reference's tests build it)
        x = np.arange(n, dtype=float)[:, None]
        dense = (kernel + kernels.DiagonalKernel(0.1, float(n))).evaluate(x)
        cov_b = np.fft.rfft(dense[0]).real
        for tag, cov in (("a", cov_a), ("b", cov_b)):
            k = f"{name}_{tag}"
            out[f"{k}_cov_rfft"] = cov
            out[f"{k}_scale"] = fft.fft1.evaluate_rfft_scale(cov_rfft=cov, size=n)
            out[f"{k}_transform_rfft"] = fft.transform_rfft(y, loc, cov_rfft=cov)
            out[f"{k}_transform_irfft"] = fft.transform_irfft(z, loc, cov_rfft=cov)
            out[f"{k}_ladj"] = fft.evaluate_rfft_log_abs_det_jac(n, cov_rfft=cov)
            out[f"{k}_log_prob"] = fft.evaluate_log_prob_rfft(y, loc, cov_rfft=cov)
            out[f"{k}_log_prob_batch"] = fft.evaluate_log_prob_rfft(yb, loc, cov_rfft=cov)
        if n <= 100:
            out[f"{name}_b_mvn_logpdf"] = stats.multivariate_normal(loc, dense).logpdf(y)
        out[f"{name}_evaluate_rfft"] = kernel.evaluate_rfft(n)
    return out


def make_2d(shape, rng):
    h, w = shape
    n = h * w
    out = {"shape": np.array(shape)}
    y, loc, z = rng.normal(size=(3, h, w))
    yb = rng.normal(size=(3, h, w))
    out.update(y=y, loc=loc, z=z, yb=yb)
    out["rfft2_y"] = np.fft.rfft2(y)
    out["irfft2_rfft2_y"] = np.fft.irfft2(out["rfft2_y"], shape)
    out["unpack_rfft2_y"] = fft.fft2.unpack_rfft2(out["rfft2_y"], shape)
    out["pack2_z"] = fft.fft2.pack_rfft2(z)
    period = np.array([float(h), float(w)])
    for name, kernel in kernel_set(period).items():
        cov_a = kernel.evaluate_rfft(shape) + 1e-3
        variants = [("a", cov_a)]
        if n <= 2500:
            xs = coordgrid(np.arange(h, dtype=float), np.arange(w, dtype=float))
            dense = (kernel + kernels.DiagonalKernel(0.1, period)).evaluate(xs)
            cov_b = np.fft.rfft2(dense[0].reshape(shape)).real
            variants.append(("b", cov_b))
        for tag, cov in variants:
            k = f"{name}_{tag}"
            out[f"{k}_cov_rfft2"] = cov
            out[f"{k}_scale"] = fft.fft2.evaluate_rfft2_scale(cov_rfft2=cov, width=w)
            out[f"{k}_transform_rfft2"] = fft.transform_rfft2(y, loc, cov_rfft2=cov)
            out[f"{k}_transform_irfft2"] = fft.transform_irfft2(z, loc, cov_rfft2=cov)
            out[f"{k}_ladj"] = fft.evaluate_rfft2_log_abs_det_jac(w, cov_rfft2=cov)
            out[f"{k}_log_prob"] = fft.evaluate_log_prob_rfft2(y, loc, cov_rfft2=cov)
            out[f"{k}_log_prob_batch"] = fft.evaluate_log_prob_rfft2(yb, loc, cov_rfft2=cov)
        if n <= 100:
            out[f"{name}_b_mvn_logpdf"] = stats.multivariate_normal(loc.ravel(), dense).logpdf(y.ravel())
        out[f"{name}_evaluate_rfft"] = kernel.evaluate_rfft(shape)
    return out


def make_kernels(rng):
    """Spectral-kernel goldens on anisotropic length scales/periods (test_stan_functions.py:410-513)."""
    out = {}
    cases = []
    for idx, (m, n) in enumerate([(7, 9), (7, 10), (8, 9), (8, 10), (35, 60), (64, 48)]):
        sigma = rng.gamma(10, 0.1)
        ls = rng.gamma(10, 0.1, 2)
        period = rng.gamma(100, 0.1, 2)
        cases.append((m, n, sigma, *ls, *period))
        out[f"c{idx}_expquad2"] = kernels.ExpQuadKernel(sigma, ls, period=period).evaluate_rfft([m, n])
        out[f"c{idx}_matern32_2"] = kernels.MaternKernel(1.5, sigma, ls, period).evaluate_rfft([m, n])
        out[f"c{idx}_matern52_2"] = kernels.MaternKernel(2.5, sigma, ls, period).evaluate_rfft([m, n])
        out[f"c{idx}_expquad1"] = kernels.ExpQuadKernel(sigma, ls[0], period=period[0]).evaluate_rfft([n])
        out[f"c{idx}_matern32_1"] = kernels.MaternKernel(1.5, sigma, ls[0], period[0]).evaluate_rfft([n])
        out[f"c{idx}_matern52_1"] = kernels.MaternKernel(2.5, sigma, ls[0], period[0]).evaluate_rfft([n])
    out["cases"] = np.array(cases)
    return out


def main():
    np.seterr(divide="raise", over="raise", invalid="raise")
    for n in SIZES_1D:
        rng = np.random.default_rng(1000 + n)
        np.savez_compressed(os.path.join(OUT, f"fft1_n{n}.npz"), **make_1d(n, rng))
    for shape in SHAPES_2D:
        rng = np.random.default_rng(2000 + 97 * shape[0] + shape[1])
        np.savez_compressed(os.path.join(OUT, f"fft2_{shape[0]}x{shape[1]}.npz"), **make_2d(shape, rng))
    np.savez_compressed(os.path.join(OUT, "kernels.npz"), **make_kernels(np.random.default_rng(3000)))
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print(f"wrote {len(SIZES_1D) + len(SHAPES_2D) + 1} files, {total / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
