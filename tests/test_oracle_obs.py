"""
CPU checks of the oracle's observation models (oracle/gp_oracle.py: obs_lpdf_vjp and the non-centred model density
built on it). The reference holds no fixtures for these (its tests only run the docs models through
cmdstan), so the densities are pinned against scipy.stats -- an independent implementation of the same published
formulas -- and the gradients against torch autograd in fp64.
"""
import numpy as np
import pytest
from scipy import stats

import gp_oracle as o


def _case(family, rng, shape=(6, 8), window=(4, 5), batch=()):
    H, W = shape
    f = rng.normal(size=batch + shape)
    if family == o.OBS_NORMAL:
        obs = rng.normal(size=window)
        obs[1, 2] = np.nan
        theta = rng.gamma(4.0, 0.3, size=batch)
    else:
        obs = rng.poisson(3.0, size=window).astype(float)
        obs[0, 1] = -1.0  # trees.stan:45
        obs[2, 0] = np.nan
        theta = rng.gamma(4.0, 0.7, size=batch) if family == o.OBS_NEG_BINOMIAL_2_LOG else None
    return f, obs, theta


@pytest.mark.parametrize("family", [o.OBS_NORMAL, o.OBS_POISSON_LOG, o.OBS_NEG_BINOMIAL_2_LOG])
def test_density_against_scipy_stats(family):
    rng = np.random.default_rng(11 + family)
    f, obs, theta = _case(family, rng)
    lp, fbar, thbar = o.obs_lpdf_vjp(family, obs, f, theta)
    oh, ow = obs.shape
    fw = f[:oh, :ow]
    seen = ~np.isnan(obs) & ((family == o.OBS_NORMAL) | ~(obs < 0))
    y = obs[seen]
    if family == o.OBS_NORMAL:
        want = stats.norm.logpdf(y, fw[seen], theta).sum()
    elif family == o.OBS_POISSON_LOG:
        want = stats.poisson.logpmf(y, np.exp(fw[seen])).sum()
    else:
        mu = np.exp(fw[seen])
        want = stats.nbinom.logpmf(y, theta, theta / (theta + mu)).sum()  # mean mu, variance mu + mu^2 / phi
    np.testing.assert_allclose(lp, want, rtol=1e-12)
    assert np.all(fbar[oh:] == 0) and np.all(fbar[:, ow:] == 0) and np.all(fbar[:oh, :ow][~seen] == 0)


@pytest.mark.parametrize("family", [o.OBS_NORMAL, o.OBS_POISSON_LOG, o.OBS_NEG_BINOMIAL_2_LOG])
def test_gradients_against_autograd(family):
    import torch

    rng = np.random.default_rng(23 + family)
    f, obs, theta = _case(family, rng, batch=(3,))
    lp, fbar, thbar = o.obs_lpdf_vjp(family, obs, f, theta)
    oh, ow = obs.shape
    seen = torch.as_tensor(~np.isnan(obs) & ((family == o.OBS_NORMAL) | ~(obs < 0)))
    y = torch.as_tensor(np.where(seen.numpy(), obs, 0.0))
    tf = torch.tensor(f, requires_grad=True)
    tt = None if theta is None else torch.tensor(theta, requires_grad=True)
    fw = tf[..., :oh, :ow]
    if family == o.OBS_NORMAL:
        th = tt[:, None, None]
        t = -0.5 * ((y - fw) / th) ** 2 - torch.log(th) - 0.5 * np.log(2 * np.pi)
    elif family == o.OBS_POISSON_LOG:
        t = y * fw - torch.exp(fw) - torch.lgamma(y + 1)
    else:
        th = tt[:, None, None]
        mu = torch.exp(fw)
        t = (torch.lgamma(y + th) - torch.lgamma(y + 1) - torch.lgamma(th) + y * torch.log(mu / (mu + th))
             + th * torch.log(th / (mu + th)))
    tl = torch.where(seen, t, torch.zeros_like(t)).sum(dim=(-2, -1))
    tl.sum().backward()
    np.testing.assert_allclose(lp, tl.detach().numpy(), rtol=1e-12)
    np.testing.assert_allclose(fbar, tf.grad.numpy(), rtol=1e-11, atol=1e-12)
    if tt is not None:
        np.testing.assert_allclose(thbar, tt.grad.numpy(), rtol=1e-10, atol=1e-12)


def test_model_density_chain_rule_finite_differences():
    """trees.stan:31-49: d/dz, d/dcov and d/d(scalar loc) of the whole non-centred density."""
    rng = np.random.default_rng(5)
    H, W = 6, 8
    z = rng.normal(size=(H, W))
    loc = np.full((H, W), 0.3)
    cov = o.gp_periodic_matern_cov_rfft2(1.5, H, W, 0.8, np.array([1.5, 2.0]), np.array([H, W])) + 1e-2
    obs = rng.poisson(2.0, size=(4, 5)).astype(float)
    obs[1, 1] = -1
    phi = 2.5
    lp, f, zbar, locbar, lsum, cbar, thbar = o.gp_inv_rfft2_obs_lpdf_vjp(o.OBS_NEG_BINOMIAL_2_LOG, z, loc, cov, obs, phi)

    def val(z_=z, loc_=loc, cov_=cov, phi_=phi):
        return o.gp_inv_rfft2_obs_lpdf_vjp(o.OBS_NEG_BINOMIAL_2_LOG, z_, loc_, cov_, obs, phi_)[0]

    eps = 1e-6
    for idx in [(0, 0), (2, 3), (5, 7), (3, 0), (4, 4)]:
        d = np.zeros_like(z)
        d[idx] = eps
        np.testing.assert_allclose((val(z_=z + d) - val(z_=z - d)) / (2 * eps), zbar[idx], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose((val(loc_=loc + eps) - val(loc_=loc - eps)) / (2 * eps), lsum, rtol=1e-6)
    np.testing.assert_allclose((val(phi_=phi + eps) - val(phi_=phi - eps)) / (2 * eps), thbar, rtol=1e-6)
    # interior spectrum entries are free parameters of the half spectrum
    for idx in [(1, 2), (4, 3), (0, 1)]:
        d = np.zeros_like(cov)
        d[idx] = eps
        np.testing.assert_allclose((val(cov_=cov + d) - val(cov_=cov - d)) / (2 * eps), cbar[idx], rtol=1e-5, atol=1e-9)


def test_1d_sibling_matches_2d_row():
    rng = np.random.default_rng(9)
    n = 12
    z = rng.normal(size=n)
    cov = o.gp_periodic_exp_quad_cov_rfft(n, 1.1, 2.0, n) + 1e-3
    y = rng.normal(size=7)
    lp, f, zbar, locbar, lsum, cbar, thbar = o.gp_inv_rfft_obs_lpdf_vjp(o.OBS_NORMAL, z, np.zeros(n), cov, y, 0.7)
    np.testing.assert_allclose(f, o.gp_inv_rfft(z, np.zeros(n), cov))
    want = stats.norm.logpdf(y, f[:7], 0.7).sum() + stats.norm.logpdf(z).sum()
    np.testing.assert_allclose(lp, want, rtol=1e-12)


@pytest.mark.parametrize("kind,nu", [("matern", 1.5), ("expquad", 0.0)])
def test_kernel_parameter_form_finite_differences(kind, nu):
    """gp_inv_rfft2_obs_lpdf_kernel_vjp (the density with the spectrum built from (sigma, length_scale, period) + nugget,
    trees.stan:26-49): a composition of functions pinned above and in tests/test_oracle_golden.py -- here the chain through
    all of them against central differences in every kernel parameter, and its 1-D sibling."""
    rng = np.random.default_rng(8)
    h, w = 6, 8
    z = rng.normal(size=(h, w))
    _, obs, theta = _case(o.OBS_NEG_BINOMIAL_2_LOG, rng)
    sigma, ls, per = 1.2, np.array([1.1, 1.7]), np.array([6.0, 9.0])
    args = lambda s, l, p: (o.OBS_NEG_BINOMIAL_2_LOG, z, np.full((h, w), 0.3), kind, nu, s, l, p, obs, theta)  # noqa: E731
    lp, zbar, lsum, (sb, lb, pb), thbar = o.gp_inv_rfft2_obs_lpdf_kernel_vjp(*args(sigma, ls, per), nugget=1e-3)
    val = lambda s, l, p: o.gp_inv_rfft2_obs_lpdf_kernel_vjp(*args(s, l, p), nugget=1e-3)[0]  # noqa: E731
    eps = 1e-6
    np.testing.assert_allclose(sb, (val(sigma + eps, ls, per) - val(sigma - eps, ls, per)) / (2 * eps), rtol=1e-6)
    for k in range(2):
        d = np.zeros(2)
        d[k] = eps
        np.testing.assert_allclose(lb[k], (val(sigma, ls + d, per) - val(sigma, ls - d, per)) / (2 * eps), rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(pb[k], (val(sigma, ls, per + d) - val(sigma, ls, per - d)) / (2 * eps), rtol=1e-6, atol=1e-8)
    # 1-D sibling on one row
    n = 10
    z1 = rng.normal(size=n)
    obs1 = np.array([2.0, 0.0, -1.0, 3.0, 1.0, np.nan, 4.0])
    a1 = lambda s, l, p: (o.OBS_NEG_BINOMIAL_2_LOG, z1, np.full(n, 0.3), kind, nu, s, l, p, obs1, 1.3)  # noqa: E731
    lp1, _, _, (sb1, lb1, pb1), _ = o.gp_inv_rfft_obs_lpdf_kernel_vjp(*a1(0.9, 1.4, 10.0), nugget=1e-3)
    v1 = lambda s, l, p: o.gp_inv_rfft_obs_lpdf_kernel_vjp(*a1(s, l, p), nugget=1e-3)[0]  # noqa: E731
    np.testing.assert_allclose(sb1, (v1(0.9 + eps, 1.4, 10.0) - v1(0.9 - eps, 1.4, 10.0)) / (2 * eps), rtol=1e-6)
    np.testing.assert_allclose(lb1, (v1(0.9, 1.4 + eps, 10.0) - v1(0.9, 1.4 - eps, 10.0)) / (2 * eps), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(pb1, (v1(0.9, 1.4, 10.0 + eps) - v1(0.9, 1.4, 10.0 - eps)) / (2 * eps), rtol=1e-6, atol=1e-8)
