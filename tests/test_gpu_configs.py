"""
GPU parity of BASELINE.json's configurations 2 and 3 as they are stated (through the C ABI, against the CPU oracle,
relative 1e-10 on values and every gradient):

  config 2  docs/trees (docs/trees/trees.md:55-74, trees.stan:14-49): the 25 x 50 tachve tree counts
            (tests/golden/tachve.csv = the reference's data file data/tachve.csv, a data fixture) padded by 10 to
            35 x 60, 20 % of the quadrats held out (-1), Matern-3/2 spectrum on the padded grid, non-centred
            parameterisation, negative-binomial observation model with the white-noise prior -- one density + gradients
            per chain, several chains per launch;
  config 3  docs/padding (docs/padding/padding.md:28-31,113-117; padded.stan:19-34): the padded sweep
            n = 128 + 16 q, q = 0..4, with 64 CHAINS FUSED IN ONE LAUNCH, every chain its own spectrum: the 1-D functions
            (one-workgroup path) and their 2-D analogue (128 + 16 q)^2 (generic three-pass path).
"""
import os

import numpy as np
import pytest
from conftest import GOLDEN, load_golden

import gp_oracle as o

pytestmark = pytest.mark.gpu

RTOL = 1e-10
CHAINS = 64
SWEEP = [128, 144, 160, 176, 192]


def close(actual, desired, rtol=RTOL, what=""):
    actual = actual.cpu().numpy() if hasattr(actual, "cpu") else np.asarray(actual)
    desired = np.asarray(desired)
    s = max(float(np.max(np.abs(desired))), 1e-300)
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=rtol * s, err_msg=what)


@pytest.fixture(scope="module")
def fft(gpu_lib):
    from gptools_amd import fft as f

    return f


def _chain_kernel_params(rng, chains):
    sigma = rng.gamma(10.0, 0.1, size=chains)
    ls = rng.gamma(20.0, 0.25, size=chains)  # ~5 lattice units, as the padding study's kernels
    return sigma, ls


@pytest.mark.parametrize("n", SWEEP)
def test_config3_padding_sweep_1d_64_chains(fft, n):
    """padded.stan: cov_rfft = gp_periodic_matern_cov_rfft(1.5, n, sigma, length_scale, n) + epsilon per chain, then the
    centred density (gp_rfft_lpdf) and the non-centred transform (gp_inv_rfft) with their reverse modes."""
    rng = np.random.default_rng(300 + n)
    sigma, ls = _chain_kernel_params(rng, CHAINS)
    cov = np.stack([o.gp_periodic_matern_cov_rfft(1.5, n, sigma[b], ls[b], float(n)) + 1e-9 for b in range(CHAINS)])
    close(fft.gp_periodic_matern_cov_rfft(1.5, n, sigma, ls, float(n)) + 1e-9, cov)
    y = rng.normal(size=(CHAINS, n))
    z = rng.normal(size=(CHAINS, n))
    loc = rng.normal(size=n)
    bar = rng.normal(size=(CHAINS, n))
    lp, gy, gc = fft.gp_rfft_lpdf_vjp(y, loc, cov)  # ONE launch over the 64 chains
    f = fft.gp_inv_rfft(z, loc, cov)
    zb, cb = fft.gp_inv_rfft_vjp(z, cov, bar)
    assert lp.shape == (CHAINS,)
    for b in range(CHAINS):
        lp_o, gy_o, gc_o = o.gp_rfft_lpdf_vjp(y[b], loc, cov[b])
        close(lp[b], lp_o, what=f"chain {b} lpdf")
        close(gy[b], gy_o, what=f"chain {b} d/dy")
        close(gc[b], gc_o, what=f"chain {b} d/dcov")
        close(f[b], o.gp_inv_rfft(z[b], loc, cov[b]), what=f"chain {b} gp_inv_rfft")
        zb_o, _, cb_o = o.gp_inv_rfft_vjp(z[b], loc, cov[b], bar[b])
        close(zb[b], zb_o, what=f"chain {b} zbar")
        close(cb[b], cb_o, what=f"chain {b} covbar")
    if n in (128, 144, 160, 176, 192):
        # the reference-generated golden of this size, evaluated as one more "chain" of the same launch geometry
        g = load_golden(f"fft1_n{n}.npz")
        for name in ("matern32_a", "expquad_a"):
            close(fft.gp_rfft_lpdf(np.broadcast_to(g["y"], (CHAINS, n)).copy(), g["loc"], g[f"{name}_cov_rfft"]),
                  np.full(CHAINS, g[f"{name}_log_prob"]))


@pytest.mark.parametrize("n", SWEEP)
def test_config3_padding_sweep_2d_64_chains(fft, n):
    """The 2-D analogue (128 + 16 q)^2 of the sweep, 64 chains per launch, a spectrum per chain."""
    assert fft.path_for_shape(n, n) == 2  # generic three-pass path
    rng = np.random.default_rng(900 + n)
    sigma, ls = _chain_kernel_params(rng, CHAINS)
    per = [float(n), float(n)]
    cov = np.stack([o.gp_periodic_matern_cov_rfft2(1.5, n, n, sigma[b], [ls[b], 1.5 * ls[b]], per) + 1e-6
                    for b in range(CHAINS)])
    y = rng.normal(size=(CHAINS, n, n))
    z = rng.normal(size=(CHAINS, n, n))
    loc = rng.normal(size=(n, n))
    lp, gy, gc = fft.gp_rfft2_lpdf_vjp(y, loc, cov)
    f = fft.gp_inv_rfft2(z, loc, cov)
    zt = fft.gp_rfft2(y, loc, cov)
    for b in range(0, CHAINS, 7):  # a sample of the chains against the oracle ...
        lp_o, gy_o, gc_o = o.gp_rfft2_lpdf_vjp(y[b], loc, cov[b])
        close(lp[b], lp_o, what=f"chain {b} lpdf")
        close(gy[b], gy_o, what=f"chain {b} d/dy")
        close(gc[b], gc_o, what=f"chain {b} d/dcov")
        close(f[b], o.gp_inv_rfft2(z[b], loc, cov[b]), what=f"chain {b} gp_inv_rfft2")
        close(zt[b], o.gp_rfft2(y[b], loc, cov[b]), what=f"chain {b} gp_rfft2")
    # ... and every chain's value
    close(lp, [o.gp_rfft2_lpdf(y[b], loc, cov[b]) for b in range(CHAINS)])


def _tachve():
    counts = np.loadtxt(os.path.join(GOLDEN, "tachve.csv"), delimiter=",", comments="#")
    assert counts.shape == (25, 50) and counts.sum() == 5724  # header line of the file: 5724 trees
    return counts


def test_config2_trees_on_its_data(fft):
    """docs/trees: the model density of trees.stan:31-49 (without the three scalar Student-t priors, which are Stan's)
    on the real tree counts, four chains with their own (z, loc, sigma, length_scale, kappa) in one launch."""
    counts = _tachve()
    nrows, ncols = counts.shape
    padding = 10  # trees.md:56-58
    H, W = nrows + padding, ncols + padding
    rng = np.random.default_rng(0)
    train = rng.uniform(size=counts.shape) < 0.8  # trees.md:61-64
    freq = np.where(train, counts, -1.0)  # -1 = held out, trees.md:72
    chains = 4
    z = rng.normal(size=(chains, H, W))
    loc_scalar = rng.normal(size=chains)
    sigma = rng.gamma(10.0, 0.1, size=chains)
    length_scale = np.exp(rng.uniform(np.log(2), np.log(28), size=chains))  # trees.stan:21
    kappa = rng.gamma(5.0, 0.2, size=chains)
    cov = np.stack([o.gp_periodic_matern_cov_rfft2(1.5, H, W, sigma[b], [length_scale[b]] * 2, [float(H), float(W)])
                    for b in range(chains)])  # trees.stan:27-29
    close(fft.gp_periodic_matern_cov_rfft2(1.5, H, W, sigma, np.stack([length_scale] * 2, -1), [float(H), float(W)]), cov)
    loc = np.broadcast_to(loc_scalar[:, None, None], (chains, H, W)).copy()  # rep_matrix(loc, ...), trees.stan:32
    phi = 1.0 / kappa  # trees.stan:46
    res = fft.gp_inv_rfft2_obs_lpdf_vjp("neg_binomial_2_log", z, loc, cov, freq, phi, prior=True)
    for b in range(chains):
        lp, f, zbar, locbar, lsum, cbar, thbar = o.gp_inv_rfft2_obs_lpdf_vjp(o.OBS_NEG_BINOMIAL_2_LOG, z[b], loc[b], cov[b],
                                                                             freq, phi[b], True)
        close(res["lp"][b], lp)
        close(res["f"][b], f)
        close(res["grad_z"][b], zbar)
        close(res["grad_loc_sum"][b], lsum)
        close(res["grad_cov"][b], cbar)
        close(res["grad_theta"][b], thbar)
        # held-out and padded quadrats carry no data: d lp / d f vanishes there
        g = res["grad_loc"][b].cpu().numpy()
        assert not g[nrows:, :].any() and not g[:, ncols:].any() and not g[:nrows, :ncols][~train].any()
        # chain rule down to the kernel parameters (what Stan's autodiff does with rfft2_cov, trees.stan:27-29)
        sb, lb, _ = fft.gp_periodic_matern_cov_rfft2_vjp(1.5, H, W, sigma[b], [length_scale[b]] * 2, [float(H), float(W)],
                                                         res["grad_cov"][b])
        sb_o, lb_o, _ = o.gp_periodic_matern_cov_rfft2_vjp(1.5, H, W, sigma[b], [length_scale[b]] * 2, [float(H), float(W)],
                                                           cbar)
        scale = float(np.abs(cbar * cov[b]).sum())
        np.testing.assert_allclose(float(sb), sb_o, rtol=1e-9, atol=1e-10 * scale / sigma[b])
        np.testing.assert_allclose(np.asarray(lb.cpu()), lb_o, rtol=1e-9, atol=1e-10 * scale / length_scale[b])
