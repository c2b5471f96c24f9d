"""
CPU ORACLE (test infrastructure only -- never imported by the product path).

NumPy restatement of the Fourier-method Gaussian-process path of onnela-lab/gptools, written to follow
the reference's Stan functions index for index. Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the shipped path (``gptools_amd``) runs on
hand-written HIP kernels and fails loudly without them.

Every function cites the reference file:line it restates (paths relative to the reference checkout).
Indices in comments are 0-based; Stan is 1-based and inclusive.

Third-party arithmetic: the reference calls Stan Math's ``fft``/``inv_fft``/``fft2``/``inv_fft2``
(cmdstan 2.36.0, Eigen FFT underneath; call sites ``stan/gptools/util.stan:264,292,303,329``). Stan Math
is not part of the reference checkout, so the published definition of the DFT is restated with
``numpy.fft`` (full complex transform, then the same truncation/real-part steps the Stan code applies).

Parity is pinned: ``tests/test_oracle_golden.py`` checks every function below against golden vectors
generated from the reference's own NumPy mirror (``tests/golden/make_golden.py``), against the dense
multivariate-normal density the reference's tests use, and (gradients -- which the reference never writes
down, Stan's autodiff produces them) against finite differences and torch autograd in fp64.

All functions accept leading batch dimensions ``(..., n)`` / ``(..., H, W)``.
"""
from __future__ import annotations

import math

import numpy as np
from scipy import special

LOG2 = math.log(2.0)
LOG2PI = math.log(2.0 * math.pi)


class NotPositiveDefinite(ValueError):
    """Stan ``reject`` of ``gp_evaluate_rfft(2)_scale`` (fft1.stan:19-21, fft2.stan:21-23)."""


# ---------------------------------------------------------------------------------------------------
# Real-FFT layer: stan/gptools/util.stan:255-330
# ---------------------------------------------------------------------------------------------------

def rfft(y):
    """util.stan:263-265 -- ``fft(y)[:n %/% 2 + 1]`` (full complex FFT, then truncate)."""
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[-1]
    return np.fft.fft(y.astype(np.complex128), axis=-1)[..., : n // 2 + 1]


def expand_rfft(y, n):
    """util.stan:271-281 -- Hermitian completion; rejects on a size mismatch (util.stan:274-276)."""
    y = np.asarray(y)
    nrfft = n // 2 + 1
    if y.shape[-1] != nrfft:
        raise ValueError(f"expected complex vector with {nrfft} elements but got {y.shape[-1]}")
    ncomplex = (n - 1) // 2
    result = np.empty(y.shape[:-1] + (n,), dtype=np.complex128)
    result[..., :nrfft] = y
    # result[nrfft + 1:n] = conj(reverse(y[2:1 + ncomplex]))  (1-based)
    result[..., nrfft:] = np.conj(y[..., 1 : 1 + ncomplex][..., ::-1])
    return result


def inv_rfft(z, n):
    """util.stan:291-293 -- ``get_real(inv_fft(expand_rfft(z, n)))``; ``inv_fft`` carries the 1/n."""
    return np.fft.ifft(expand_rfft(z, n), axis=-1).real


def rfft2(y):
    """util.stan:302-304 -- ``fft2(y)[:, :m %/% 2 + 1]``."""
    y = np.asarray(y, dtype=np.float64)
    w = y.shape[-1]
    return np.fft.fft2(y.astype(np.complex128), axes=(-2, -1))[..., : w // 2 + 1]


def inv_rfft2(z, m):
    """util.stan:315-330 -- fill the redundant half, ``get_real(inv_fft2(x))``.

    Columns 0 and (m even) m/2 are used as given for all rows; the real part drops whatever is not
    Hermitian in them.
    """
    z = np.asarray(z)
    n = z.shape[-2]
    mrfft = m // 2 + 1
    mcomplex = (m - 1) // 2
    if z.shape[-1] != mrfft:
        raise ValueError(f"expected complex matrix with {mrfft} columns but got {z.shape[-1]}")
    x = np.empty(z.shape[:-1] + (m,), dtype=np.complex128)
    x[..., :mrfft] = z
    # util.stan:322-324: x[i, mrfft + 1:m] = conj(reverse(z[i, 2:1 + mcomplex]))
    x[..., mrfft:] = np.conj(z[..., 1 : 1 + mcomplex][..., ::-1])
    # util.stan:326-328: x[2:, i] = reverse(x[2:, i]) for the filled columns
    x[..., 1:, mrfft:] = x[..., 1:, mrfft:][..., ::-1, :]
    return np.fft.ifft2(x, axes=(-2, -1)).real


# ---------------------------------------------------------------------------------------------------
# 1-D GP layer: stan/gptools/fft1.stan
# ---------------------------------------------------------------------------------------------------

def _reject_min(r):
    # Stan's min() over the whole container, evaluated per lattice of a batch here.
    minval = np.min(r) if r.size else np.inf
    if minval < 0:
        raise NotPositiveDefinite(
            f"covariance matrix is not positive-definite (minimum eigenvalue is {minval})"
        )


def gp_evaluate_rfft_scale(cov_rfft, n):
    """fft1.stan:14-30. The rejected ``minval`` is of ``n * cov_rfft / 2`` (fft1.stan:16-20)."""
    cov_rfft = np.asarray(cov_rfft, dtype=np.float64)
    nrfft = n // 2 + 1
    if cov_rfft.shape[-1] != nrfft:
        raise ValueError(f"expected vector with {nrfft} elements but got {cov_rfft.shape[-1]}")
    result = n * cov_rfft / 2
    _reject_min(result)
    result[..., 0] *= 2
    if n % 2 == 0:
        result[..., nrfft - 1] *= 2
    return np.sqrt(result)


def gp_unpack_rfft(x, n):
    """fft1.stan:44-51 -- ``[Re_0 .. Re_{n//2}, Im_1 .. Im_{(n-1)//2}]``."""
    x = np.asarray(x)
    ncomplex = (n - 1) // 2
    nrfft = n // 2 + 1
    z = np.empty(x.shape[:-1] + (n,), dtype=np.float64)
    z[..., :nrfft] = x.real[..., :nrfft]
    z[..., nrfft:] = x.imag[..., 1 : 1 + ncomplex]
    return z


def gp_pack_rfft(z):
    """fft1.stan:108-118 -- inverse of :func:`gp_unpack_rfft`."""
    z = np.asarray(z, dtype=np.float64)
    n = z.shape[-1]
    ncomplex = (n - 1) // 2
    nrfft = n // 2 + 1
    fft = z[..., :nrfft].astype(np.complex128)
    fft[..., 1 : ncomplex + 1] += 1j * z[..., nrfft:]
    return fft


def gp_rfft(y, loc, cov_rfft):
    """fft1.stan:64-66."""
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[-1]
    return gp_unpack_rfft(rfft(y - loc) / gp_evaluate_rfft_scale(cov_rfft, n), n)


def gp_rfft_log_abs_det_jac(cov_rfft, n):
    """fft1.stan:78-82."""
    rfft_scale = gp_evaluate_rfft_scale(cov_rfft, n)
    ls = np.log(rfft_scale)
    return (
        -ls[..., : n // 2 + 1].sum(axis=-1)
        - ls[..., 1 : (n + 1) // 2].sum(axis=-1)
        - LOG2 * ((n - 1) // 2)
        + n * math.log(n) / 2
    )


def std_normal_lpdf(z, axes):
    """Stan Math ``std_normal_lpdf`` called as a function: constants are kept (fft1.stan:100)."""
    z = np.asarray(z)
    size = int(np.prod([z.shape[a] for a in axes]))
    return -0.5 * np.sum(z * z, axis=axes) - 0.5 * size * LOG2PI


def gp_rfft_lpdf(y, loc, cov_rfft):
    """fft1.stan:96-101."""
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[-1]
    z = gp_rfft(y, loc, cov_rfft)
    return std_normal_lpdf(z, (-1,)) + gp_rfft_log_abs_det_jac(cov_rfft, n)


def gp_inv_rfft(z, loc, cov_rfft):
    """fft1.stan:139-143."""
    z = np.asarray(z, dtype=np.float64)
    n = z.shape[-1]
    rfft_scale = gp_evaluate_rfft_scale(cov_rfft, n)
    return inv_rfft(rfft_scale * gp_pack_rfft(z), n) + loc


def gp_periodic_exp_quad_cov_rfft(n, sigma, length_scale, period):
    """fft1.stan:154-158."""
    nrfft = n // 2 + 1
    k = np.arange(nrfft, dtype=np.float64)
    return (
        n * sigma**2 * length_scale / period * math.sqrt(2 * math.pi)
        * np.exp(-2 * (math.pi * k * length_scale / period) ** 2)
    )


def gp_periodic_matern_cov_rfft(nu, n, sigma, length_scale, period):
    """fft1.stan:170-176 (``tgamma`` is Stan Math's gamma function)."""
    nrfft = n // 2 + 1
    k = np.arange(nrfft, dtype=np.float64)
    return (
        sigma**2 * n * math.sqrt(2 * math.pi / nu) * special.gamma(nu + 0.5) / special.gamma(nu)
        * (1 + 2 / nu * (math.pi * length_scale / period * k) ** 2) ** -(nu + 0.5)
        * length_scale / period
    )


# ---------------------------------------------------------------------------------------------------
# 2-D GP layer: stan/gptools/fft2.stan
# ---------------------------------------------------------------------------------------------------

def gp_evaluate_rfft2_scale(cov_rfft2, width):
    """fft2.stan:13-40."""
    cov_rfft2 = np.asarray(cov_rfft2, dtype=np.float64)
    height = cov_rfft2.shape[-2]
    n = width * height
    fftwidth = width // 2 + 1
    fftheight = height // 2 + 1
    if cov_rfft2.shape[-1] != fftwidth:
        raise ValueError(f"expected matrix with {fftwidth} columns but got {cov_rfft2.shape[-1]}")
    fftscale = n * cov_rfft2 / 2
    _reject_min(fftscale)
    fftscale[..., 0, 0] *= 2
    if height % 2 == 0:
        fftscale[..., fftheight - 1, 0] *= 2
    if width % 2 == 0:
        fftscale[..., 0, fftwidth - 1] *= 2
    if width % 2 == 0 and height % 2 == 0:
        fftscale[..., fftheight - 1, fftwidth - 1] *= 2
    return np.sqrt(fftscale)


def gp_unpack_rfft2(z, m):
    """fft2.stan:55-74 -- the packing contract; rows > H/2 of the edge columns are never read."""
    z = np.asarray(z)
    height = z.shape[-2]
    fftwidth = m // 2 + 1
    fftheight = height // 2 + 1
    wcomplex = (m - 1) // 2
    result = np.empty(z.shape[:-2] + (height, m), dtype=np.float64)
    result[..., :, 0] = gp_unpack_rfft(z[..., :fftheight, 0], height)
    result[..., :, 1 : wcomplex + 1] = z.real[..., :, 1 : wcomplex + 1]
    result[..., :, 1 + wcomplex : 2 * wcomplex + 1] = z.imag[..., :, 1 : wcomplex + 1]
    if m % 2 == 0:
        result[..., :, m - 1] = gp_unpack_rfft(z[..., :fftheight, fftwidth - 1], height)
    return result


def gp_pack_rfft2(z):
    """fft2.stan:85-100."""
    z = np.asarray(z, dtype=np.float64)
    height, width = z.shape[-2:]
    ncomplex = (width - 1) // 2
    result = np.empty(z.shape[:-2] + (height, width // 2 + 1), dtype=np.complex128)
    result[..., :, 0] = expand_rfft(gp_pack_rfft(z[..., :, 0]), height)
    result[..., :, 1 : ncomplex + 1] = (
        z[..., :, 1 : ncomplex + 1] + 1j * z[..., :, ncomplex + 1 : 2 * ncomplex + 1]
    )
    if width % 2 == 0:
        result[..., :, width // 2] = expand_rfft(gp_pack_rfft(z[..., :, width - 1]), height)
    return result


def gp_rfft2(y, loc, cov_rfft2):
    """fft2.stan:112-114."""
    y = np.asarray(y, dtype=np.float64)
    w = y.shape[-1]
    return gp_unpack_rfft2(rfft2(y - loc) / gp_evaluate_rfft2_scale(cov_rfft2, w), w)


def gp_inv_rfft2(z, loc, cov_rfft2):
    """fft2.stan:126-130."""
    z = np.asarray(z, dtype=np.float64)
    w = z.shape[-1]
    y = gp_pack_rfft2(z) * gp_evaluate_rfft2_scale(cov_rfft2, w)
    return inv_rfft2(y, w) + loc


def gp_rfft2_log_abs_det_jac(cov_rfft2, width):
    """fft2.stan:143-173. (``width == 1`` double counts column 0, fft2.stan:155,162: require W >= 2.)"""
    cov_rfft2 = np.asarray(cov_rfft2, dtype=np.float64)
    height = cov_rfft2.shape[-2]
    n = width * height
    fftwidth = width // 2 + 1
    fftheight = height // 2 + 1
    ls = np.log(gp_evaluate_rfft2_scale(cov_rfft2, width))
    # idx = (height % 2) ? fftheight : fftheight - 1 ; sums over [1:fftheight] and [2:idx] (1-based)
    idx = fftheight if height % 2 else fftheight - 1
    ladj = -(ls[..., :fftheight, 0].sum(axis=-1) + ls[..., 1:idx, 0].sum(axis=-1))
    ladj = ladj - 2 * ls[..., :, 1 : fftwidth - 1].sum(axis=(-2, -1))
    if width % 2:
        ladj = ladj - 2 * ls[..., :, fftwidth - 1].sum(axis=-1)
    else:
        ladj = ladj - (
            ls[..., :fftheight, fftwidth - 1].sum(axis=-1) + ls[..., 1:idx, fftwidth - 1].sum(axis=-1)
        )
    nterms = (n - 1) // 2
    if height % 2 == 0 and width % 2 == 0:
        nterms -= 1
    return ladj - LOG2 * nterms + n * math.log(n) / 2


def gp_rfft2_lpdf(y, loc, cov_rfft2):
    """fft2.stan:189-192."""
    y = np.asarray(y, dtype=np.float64)
    w = y.shape[-1]
    z = gp_rfft2(y, loc, cov_rfft2)
    return std_normal_lpdf(z, (-2, -1)) + gp_rfft2_log_abs_det_jac(cov_rfft2, w)


def gp_periodic_exp_quad_cov_rfft2(height, width, sigma, length_scale, period):
    """fft2.stan:206-213 -- outer product of the two 1-D spectra, the height one Hermitian-expanded."""
    rfftm = gp_periodic_exp_quad_cov_rfft(height, sigma, length_scale[0], period[0])
    rfftn = gp_periodic_exp_quad_cov_rfft(width, 1.0, length_scale[1], period[1])
    return np.outer(expand_rfft(rfftm, height).real, rfftn)


def gp_periodic_matern_cov_rfft2(nu, height, width, sigma, length_scale, period):
    """fft2.stan:228-247."""
    nrfft = width // 2 + 1
    result = np.empty((height, nrfft), dtype=np.float64)
    ndim = 2
    col_part = (np.arange(nrfft, dtype=np.float64) * length_scale[1] / period[1]) ** 2
    for i in range(height // 2 + 1):  # fft2.stan:237-243, krow = i
        result[i] = 1 + 2 / nu * math.pi**2 * ((i * length_scale[0] / period[0]) ** 2 + col_part)
        if i > 0:
            result[height - i] = result[i]
    return (
        sigma**2 * height * width * 2**ndim * (math.pi / (2 * nu)) ** (ndim / 2)
        * special.gamma(nu + ndim / 2) / special.gamma(nu)
        * result ** -(nu + ndim / 2)
        * (length_scale[0] / period[0]) * (length_scale[1] / period[1])
    )


# ---------------------------------------------------------------------------------------------------
# Reverse-mode gradients.
#
# The reference contains no gradient code: Stan's autodiff differentiates the functions above. The
# closed forms below are the exact derivatives of those functions (derivations in DESIGN.md); they are
# checked against finite differences and torch autograd of an fp64 restatement in tests/.
# ---------------------------------------------------------------------------------------------------

def _edge_cols(width):
    """Columns of the half spectrum produced by purely real row transforms: 0 and (W even) W/2."""
    return [0] + ([width // 2] if width % 2 == 0 else [])


def _entry_kinds2(height, width):
    """kind[i, j] of every half-spectrum entry as read by fft2.stan:55-74 / 143-173:
    2 = complex pair (Re and Im both whitened), 1 = real point (Re only), 0 = never read."""
    fftwidth = width // 2 + 1
    kind = np.full((height, fftwidth), 2, dtype=np.int64)
    hc = (height - 1) // 2
    for j in _edge_cols(width):
        kind[:, j] = 0
        kind[0, j] = 1
        kind[1 : 1 + hc, j] = 2
        if height % 2 == 0:
            kind[height // 2, j] = 1
    return kind


def gp_rfft2_lpdf_vjp(y, loc, cov_rfft2):
    """Value and gradients of fft2.stan:189-192 w.r.t. ``y`` and ``cov_rfft2`` (``d/dloc = -d/dy``).

    lpdf = -sum_pairs(|F|^2/(n c) + log c) - 1/2 sum_reals((Re F)^2/(n c) + log c) - n/2 log(2 pi)
    over the entries fft2.stan actually reads (``_entry_kinds2``).
    """
    y = np.asarray(y, dtype=np.float64)
    c = np.asarray(cov_rfft2, dtype=np.float64)
    height, width = y.shape[-2:]
    n = height * width
    lpdf = gp_rfft2_lpdf(y, loc, c)
    F = rfft2(y - loc)
    kind = _entry_kinds2(height, width)
    c_b = np.broadcast_to(c, F.shape[:-2] + c.shape[-2:])
    pair = kind == 2
    real = kind == 1
    with np.errstate(divide="ignore", invalid="ignore"):
        gc = np.where(pair, (F.real**2 + F.imag**2) / (n * c_b**2) - 1 / c_b, 0.0)
        gc = np.where(real, 0.5 * (F.real**2 / (n * c_b**2) - 1 / c_b), gc)
    # d lpdf / d y = -irfft2(F / c_eff) with the edge columns completed from their rows <= H/2.
    G = F / c_b
    hc = (height - 1) // 2
    for j in _edge_cols(width):
        # real points carry Re F only; rows > H/2 mirror rows 1..hc
        G[..., 0, j] = F[..., 0, j].real / c_b[..., 0, j]
        if height % 2 == 0:
            G[..., height // 2, j] = F[..., height // 2, j].real / c_b[..., height // 2, j]
        G[..., height - hc :, j] = np.conj(G[..., 1 : 1 + hc, j][..., ::-1])
    gy = -inv_rfft2(G, width)
    return lpdf, gy, gc


def gp_rfft_lpdf_vjp(y, loc, cov_rfft):
    """Value and gradients of fft1.stan:96-101 w.r.t. ``y`` and ``cov_rfft``."""
    y = np.asarray(y, dtype=np.float64)
    c = np.asarray(cov_rfft, dtype=np.float64)
    n = y.shape[-1]
    lpdf = gp_rfft_lpdf(y, loc, c)
    F = rfft(y - loc)
    c_b = np.broadcast_to(c, F.shape[:-1] + c.shape[-1:])
    real = np.zeros(n // 2 + 1, dtype=bool)
    real[0] = True
    if n % 2 == 0:
        real[n // 2] = True
    gc = np.where(
        real,
        0.5 * (F.real**2 / (n * c_b**2) - 1 / c_b),
        (F.real**2 + F.imag**2) / (n * c_b**2) - 1 / c_b,
    )
    G = np.where(real, F.real / c_b + 0j, F / c_b)
    gy = -inv_rfft(G, n)
    return lpdf, gy, gc


def gp_inv_rfft2_vjp(z, loc, cov_rfft2, fbar):
    """VJP of fft2.stan:126-130: given ``fbar = dL/df`` return ``(zbar, locbar, covbar)``."""
    z = np.asarray(z, dtype=np.float64)
    c = np.asarray(cov_rfft2, dtype=np.float64)
    fbar = np.asarray(fbar, dtype=np.float64)
    height, width = z.shape[-2:]
    n = height * width
    wc = (width - 1) // 2
    hc = (height - 1) // 2
    hf = height // 2 + 1
    s = gp_evaluate_rfft2_scale(c, width)
    s_b = np.broadcast_to(s, z.shape[:-2] + s.shape[-2:])
    c_b = np.broadcast_to(c, s_b.shape)
    P = gp_pack_rfft2(z)
    R = rfft2(fbar)
    # interior columns: every (i, j) stands for itself and its mirror -> 2/n
    sbar = (2.0 / n) * (P * np.conj(R)).real
    zbar = np.empty_like(z)
    zbar[..., :, 1 : 1 + wc] = (2.0 / n) * s_b[..., :, 1 : 1 + wc] * R.real[..., :, 1 : 1 + wc]
    zbar[..., :, 1 + wc : 1 + 2 * wc] = (2.0 / n) * s_b[..., :, 1 : 1 + wc] * R.imag[..., :, 1 : 1 + wc]
    for j, jz in zip(_edge_cols(width), [0, width - 1]):
        # edge columns: all H rows of s are used (fft2.stan:127-128), each exactly once -> 1/n
        sbar[..., :, j] = (1.0 / n) * (P[..., :, j] * np.conj(R[..., :, j])).real
        sj = s_b[..., :, j]
        ssum = sj[..., 1 : 1 + hc] + sj[..., height - hc :][..., ::-1]  # s[k] + s[H-k], k = 1..hc
        zbar[..., 0, jz] = sj[..., 0] * R.real[..., 0, j] / n
        zbar[..., 1 : 1 + hc, jz] = ssum * R.real[..., 1 : 1 + hc, j] / n
        if height % 2 == 0:
            zbar[..., height // 2, jz] = sj[..., height // 2] * R.real[..., height // 2, j] / n
        zbar[..., hf:, jz] = ssum * R.imag[..., 1 : 1 + hc, j] / n
    with np.errstate(divide="ignore", invalid="ignore"):
        cbar = sbar * s_b / (2 * c_b)
    return zbar, fbar.copy(), cbar


def gp_inv_rfft_vjp(z, loc, cov_rfft, fbar):
    """VJP of fft1.stan:139-143."""
    z = np.asarray(z, dtype=np.float64)
    c = np.asarray(cov_rfft, dtype=np.float64)
    fbar = np.asarray(fbar, dtype=np.float64)
    n = z.shape[-1]
    nc = (n - 1) // 2
    nrfft = n // 2 + 1
    s = gp_evaluate_rfft_scale(c, n)
    s_b = np.broadcast_to(s, z.shape[:-1] + s.shape[-1:])
    c_b = np.broadcast_to(c, s_b.shape)
    P = gp_pack_rfft(z)
    R = rfft(fbar)
    kappa = np.full(nrfft, 2.0 / n)
    kappa[0] = 1.0 / n
    if n % 2 == 0:
        kappa[n // 2] = 1.0 / n
    sbar = kappa * (P * np.conj(R)).real
    zbar = np.empty_like(z)
    zbar[..., :nrfft] = kappa * s_b * R.real
    zbar[..., nrfft:] = (2.0 / n) * s_b[..., 1 : 1 + nc] * R.imag[..., 1 : 1 + nc]
    with np.errstate(divide="ignore", invalid="ignore"):
        cbar = sbar * s_b / (2 * c_b)
    return zbar, fbar.copy(), cbar


def gp_rfft2_log_abs_det_jac_vjp(cov_rfft2, width):
    """Gradient of fft2.stan:143-173 w.r.t. ``cov_rfft2``: -(number of whitened entries)/(2 c)."""
    c = np.asarray(cov_rfft2, dtype=np.float64)
    kind = _entry_kinds2(c.shape[-2], width)
    return -0.5 * kind / c


def gp_rfft_log_abs_det_jac_vjp(cov_rfft, n):
    """Gradient of fft1.stan:78-82 w.r.t. ``cov_rfft``."""
    c = np.asarray(cov_rfft, dtype=np.float64)
    kind = np.full(n // 2 + 1, 2.0)
    kind[0] = 1.0
    if n % 2 == 0:
        kind[n // 2] = 1.0
    return -0.5 * kind / c


def gp_periodic_exp_quad_cov_rfft2_vjp(height, width, sigma, length_scale, period, cbar):
    """VJP of fft2.stan:206-213: ``cbar -> (sigma_bar, length_scale_bar[2], period_bar[2])``."""
    c = gp_periodic_exp_quad_cov_rfft2(height, width, sigma, length_scale, period)
    ki = np.minimum(np.arange(height), height - np.arange(height)).astype(np.float64)[:, None]
    kj = np.arange(width // 2 + 1, dtype=np.float64)[None, :]
    w = np.asarray(cbar) * c
    l1, l2 = length_scale
    p1, p2 = period
    f1 = 4 * math.pi**2 * ki**2 * l1**2 / p1**2
    f2 = 4 * math.pi**2 * kj**2 * l2**2 / p2**2
    sigma_bar = np.sum(w * 2 / sigma, axis=(-2, -1))
    ls_bar = np.stack(
        [np.sum(w * (1 - f1) / l1, axis=(-2, -1)), np.sum(w * (1 - f2) / l2, axis=(-2, -1))], axis=-1
    )
    p_bar = np.stack(
        [np.sum(w * (f1 - 1) / p1, axis=(-2, -1)), np.sum(w * (f2 - 1) / p2, axis=(-2, -1))], axis=-1
    )
    return sigma_bar, ls_bar, p_bar


def gp_periodic_matern_cov_rfft2_vjp(nu, height, width, sigma, length_scale, period, cbar):
    """VJP of fft2.stan:228-247 (``nu`` is data in every reference model: no d/dnu)."""
    c = gp_periodic_matern_cov_rfft2(nu, height, width, sigma, length_scale, period)
    ki = np.minimum(np.arange(height), height - np.arange(height)).astype(np.float64)[:, None]
    kj = np.arange(width // 2 + 1, dtype=np.float64)[None, :]
    l1, l2 = length_scale
    p1, p2 = period
    a1 = (ki * l1 / p1) ** 2
    a2 = (kj * l2 / p2) ** 2
    arg = 1 + 2 / nu * math.pi**2 * (a1 + a2)
    w = np.asarray(cbar) * c
    g = (nu + 1) * (4 * math.pi**2 / nu) / arg  # -(nu + 1) * d log(arg) / d log(l) = -g * a
    sigma_bar = np.sum(w * 2 / sigma, axis=(-2, -1))
    ls_bar = np.stack(
        [np.sum(w * (1 - g * a1) / l1, axis=(-2, -1)), np.sum(w * (1 - g * a2) / l2, axis=(-2, -1))],
        axis=-1,
    )
    p_bar = np.stack(
        [np.sum(w * (g * a1 - 1) / p1, axis=(-2, -1)), np.sum(w * (g * a2 - 1) / p2, axis=(-2, -1))],
        axis=-1,
    )
    return sigma_bar, ls_bar, p_bar


def gp_periodic_exp_quad_cov_rfft_vjp(n, sigma, length_scale, period, cbar):
    """VJP of fft1.stan:154-158."""
    c = gp_periodic_exp_quad_cov_rfft(n, sigma, length_scale, period)
    k = np.arange(n // 2 + 1, dtype=np.float64)
    w = np.asarray(cbar) * c
    f = 4 * math.pi**2 * k**2 * length_scale**2 / period**2
    return (
        np.sum(w * 2 / sigma, axis=-1),
        np.sum(w * (1 - f) / length_scale, axis=-1),
        np.sum(w * (f - 1) / period, axis=-1),
    )


def gp_periodic_matern_cov_rfft_vjp(nu, n, sigma, length_scale, period, cbar):
    """VJP of fft1.stan:170-176."""
    c = gp_periodic_matern_cov_rfft(nu, n, sigma, length_scale, period)
    k = np.arange(n // 2 + 1, dtype=np.float64)
    a = (k * length_scale / period) ** 2
    arg = 1 + 2 / nu * math.pi**2 * a
    w = np.asarray(cbar) * c
    g = (nu + 0.5) * (4 * math.pi**2 / nu) / arg
    return (
        np.sum(w * 2 / sigma, axis=-1),
        np.sum(w * (1 - g * a) / length_scale, axis=-1),
        np.sum(w * (g * a - 1) / period, axis=-1),
    )


def gp_rfft2_vjp(y, loc, cov_rfft2, zbar):
    """VJP of fft2.stan:112-114 (``z = gp_rfft2(y, loc, c)``): ``zbar -> (ybar, locbar, covbar)``."""
    y = np.asarray(y, dtype=np.float64)
    c = np.asarray(cov_rfft2, dtype=np.float64)
    zbar = np.asarray(zbar, dtype=np.float64)
    height, width = y.shape[-2:]
    hc = (height - 1) // 2
    s = gp_evaluate_rfft2_scale(c, width)
    s_b = np.broadcast_to(s, y.shape[:-2] + s.shape[-2:])
    c_b = np.broadcast_to(c, s_b.shape)
    F = rfft2(y - loc)
    # dL/dF as a half-spectrum "pack" of zbar / s; the adjoint of x -> rfft2(x) read entry-wise is
    # n * irfft2 of the entries weighted 1/2 where an entry stands for two full-spectrum slots.
    Q = gp_pack_rfft2(zbar)  # Re/Im parts in place; edge columns Hermitian-expanded
    D = Q / s_b
    kind = _entry_kinds2(height, width)
    # covbar: z = Re/Im(F)/s -> dz/ds = -z/s ; ds/dc = s/(2c)
    sbar = -(Q.real * F.real + Q.imag * F.imag) / s_b**2
    sbar = np.where(kind == 0, 0.0, sbar)
    sbar = np.where(kind == 1, -(Q.real * F.real) / s_b**2, sbar)
    with np.errstate(divide="ignore", invalid="ignore"):
        cbar = sbar * s_b / (2 * c_b)
    # ybar[a,b] = sum_entries Re-part: zbar_re/s * cos(..) ... == Re sum_k D_k e^{-i theta}; assemble
    # the full-spectrum weights: interior entries count once (weight n/2 after irfft2's mirroring),
    # edge entries rows 1..hc likewise, real points weight n.
    Dh = np.where(kind == 2, 0.5 * D, 0.0)
    Dh = np.where(kind == 1, D.real + 0j, Dh)
    for j in _edge_cols(width):
        Dh[..., height - hc :, j] = np.conj(Dh[..., 1 : 1 + hc, j][..., ::-1])
    # Re F = sum x cos(theta), Im F = -sum x sin(theta)  =>  ybar = Re sum_entries D e^{+i theta}
    #   = n * irfft2(Dh): a pair entry stands for two full-spectrum slots (hence the 1/2 above).
    ybar = height * width * inv_rfft2(Dh, width)
    return ybar, -ybar, cbar


# ---------------------------------------------------------------------------------------------------
# Observation models of the docs workloads on the latent field
# ---------------------------------------------------------------------------------------------------
# The densities are Stan Math's (third party, cmdstan 2.36.0; not in the reference checkout): their published
# definitions are restated here with scipy.special. Call sites in the reference: docs/trees/trees.stan:46
# (neg_binomial_2), docs/padding/padded.stan:34 (normal),
# docs/poisson_regression/poisson_regression_fourier_non_centered.stan:26 (poisson_log). Pinned against
# scipy.stats and fp64 autograd in tests/test_oracle_obs.py; the reference holds no fixtures for them
# (its tests only check that the docs models run: python/stan/tests/test_examples.py).

OBS_NORMAL, OBS_POISSON_LOG, OBS_NEG_BINOMIAL_2_LOG = 0, 1, 2


def obs_lpdf_vjp(family, obs, f, theta=None):
    """Sum over the observed corner of ``family_lpdf(obs | f, theta)`` with d/df (zero where unobserved) and d/dtheta.

    ``obs`` is ``(..., h, w)`` with ``h <= H``, ``w <= W`` (the unpadded corner, trees.stan:43-44); NaN, and for the
    count families a negative value (trees.stan:45), masks a point. ``theta`` has the batch shape.
    """
    f = np.asarray(f, dtype=np.float64)
    obs = np.asarray(obs, dtype=np.float64)
    oh, ow = obs.shape[-2:]
    bshape = f.shape[:-2]
    y = np.broadcast_to(obs, bshape + (oh, ow))
    fw = f[..., :oh, :ow]
    seen = ~np.isnan(y)
    if family != OBS_NORMAL:
        seen &= ~(y < 0)
    ys = np.where(seen, y, 0.0)
    th = None if theta is None else np.broadcast_to(np.asarray(theta, dtype=np.float64), bshape)[..., None, None]
    if family == OBS_NORMAL:
        r = (ys - fw) / th
        lp = -0.5 * r * r - np.log(th) - 0.5 * LOG2PI
        df = r / th
        dth = (r * r - 1.0) / th
    elif family == OBS_POISSON_LOG:
        mu = np.exp(fw)
        lp = ys * fw - mu - special.gammaln(ys + 1.0)
        df = ys - mu
        dth = np.zeros_like(fw)
    elif family == OBS_NEG_BINOMIAL_2_LOG:
        # neg_binomial_2(n | mu = exp(f), phi): lchoose(n + phi - 1, n) + n log(mu / (mu + phi)) + phi log(phi / (mu + phi))
        lphi = np.log(th)
        lse = np.logaddexp(fw, lphi)
        lp = (special.gammaln(ys + th) - special.gammaln(ys + 1.0) - special.gammaln(th) + ys * (fw - lse)
              + th * (lphi - lse))
        df = ys - (ys + th) * np.exp(fw - lse)
        dth = special.digamma(ys + th) - special.digamma(th) + lphi - lse + 1.0 - (ys + th) * np.exp(-lse)
    else:
        raise ValueError(f"unknown observation family {family}")
    fbar = np.zeros_like(f)
    fbar[..., :oh, :ow] = np.where(seen, df, 0.0)
    return np.where(seen, lp, 0.0).sum(axis=(-2, -1)), fbar, np.where(seen, dth, 0.0).sum(axis=(-2, -1))


def gp_inv_rfft2_obs_lpdf_vjp(family, z, loc, cov_rfft2, obs, theta=None, prior=True):
    """Log density of a non-centred docs model and its gradients (trees.stan:31-49 without the scalar priors):
    ``f = gp_inv_rfft2(z, loc, cov)``; ``lp = sum obs_lpdf(obs | f, theta) [+ std_normal_lpdf(z)]``.
    Returns ``(lp, f, zbar, locbar, locbar_sum, covbar, thetabar)``."""
    z = np.asarray(z, dtype=np.float64)
    f = gp_inv_rfft2(z, loc, cov_rfft2)
    lp, fbar, thbar = obs_lpdf_vjp(family, obs, f, theta)
    zbar, locbar, cbar = gp_inv_rfft2_vjp(z, loc, cov_rfft2, fbar)
    if prior:
        lp = lp + std_normal_lpdf(z, (-2, -1))
        zbar = zbar - z
    return lp, f, zbar, locbar, locbar.sum(axis=(-2, -1)), cbar, thbar


def gp_inv_rfft_obs_lpdf_vjp(family, z, loc, cov_rfft, obs, theta=None, prior=True):
    """1-D sibling (padded.stan:29-34, poisson_regression_fourier_non_centered.stan:20-26)."""
    z = np.asarray(z, dtype=np.float64)
    f = gp_inv_rfft(z, loc, cov_rfft)
    lp, fbar, thbar = obs_lpdf_vjp(family, np.asarray(obs, dtype=np.float64)[..., None, :], f[..., None, :], theta)
    fbar = fbar[..., 0, :]
    zbar, locbar, cbar = gp_inv_rfft_vjp(z, loc, cov_rfft, fbar)
    if prior:
        lp = lp + std_normal_lpdf(z, (-1,))
        zbar = zbar - z
    return lp, f, zbar, locbar, locbar.sum(axis=-1), cbar, thbar


def gp_inv_rfft2_obs_lpdf_kernel_vjp(family, z, loc, kind, nu, sigma, length_scale, period, obs, theta=None, nugget=0.0,
                                     prior=True):
    """The non-centred model density with the spectrum built from its kernel, as docs/trees/trees.stan:26-29 does on
    every gradient evaluation: ``cov = kernel(sigma, length_scale, period) + nugget`` (fft2.stan:206-213 / :228-247), then
    :func:`gp_inv_rfft2_obs_lpdf_vjp`, then the kernel's reverse mode. Returns
    ``(lp, zbar, locbar_sum, (sigma_bar, length_scale_bar, period_bar), thetabar)``."""
    h, w = np.asarray(z).shape[-2:]
    if kind == "expquad":
        cov = gp_periodic_exp_quad_cov_rfft2(h, w, sigma, length_scale, period) + nugget
    else:
        cov = gp_periodic_matern_cov_rfft2(nu, h, w, sigma, length_scale, period) + nugget
    lp, _, zbar, _, lsum, cbar, thbar = gp_inv_rfft2_obs_lpdf_vjp(family, z, loc, cov, obs, theta, prior)
    if kind == "expquad":
        kbar = gp_periodic_exp_quad_cov_rfft2_vjp(h, w, sigma, length_scale, period, cbar)
    else:
        kbar = gp_periodic_matern_cov_rfft2_vjp(nu, h, w, sigma, length_scale, period, cbar)
    return lp, zbar, lsum, kbar, thbar


def gp_inv_rfft_obs_lpdf_kernel_vjp(family, z, loc, kind, nu, sigma, length_scale, period, obs, theta=None, nugget=0.0,
                                    prior=True):
    """1-D sibling (fft1.stan:154-176 feeding fft1.stan:67-70)."""
    n = np.asarray(z).shape[-1]
    if kind == "expquad":
        cov = gp_periodic_exp_quad_cov_rfft(n, sigma, length_scale, period) + nugget
    else:
        cov = gp_periodic_matern_cov_rfft(nu, n, sigma, length_scale, period) + nugget
    lp, _, zbar, _, lsum, cbar, thbar = gp_inv_rfft_obs_lpdf_vjp(family, z, loc, cov, obs, theta, prior)
    if kind == "expquad":
        kbar = gp_periodic_exp_quad_cov_rfft_vjp(n, sigma, length_scale, period, cbar)
    else:
        kbar = gp_periodic_matern_cov_rfft_vjp(nu, n, sigma, length_scale, period, cbar)
    return lp, zbar, lsum, kbar, thbar
