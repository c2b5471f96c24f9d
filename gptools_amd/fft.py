"""
PyTorch-ROCm shim over the C ABI (include/gptools_hip.h): device tensors in, device tensors out.

Function names and argument meaning follow the reference's Stan library (``stan/gptools/fft1.stan``,
``fft2.stan``, ``util.stan:255-330``); the NumPy-mirror names of ``python/util/gptools/util/fft`` are
provided as aliases at the bottom. torch is plumbing here (device memory and streams): every function
calls straight into ``libgptools_hip.so``; there is no eager/CPU fallback.

Shapes: ``y, loc, z`` are ``(..., H, W)`` (``(..., n)`` for the 1-D functions), ``cov_rfft2`` is
``(..., H, W // 2 + 1)``; ``loc`` / ``cov`` without batch dimensions are shared by the whole batch.
"""
from __future__ import annotations

import ctypes
import math
import threading
import weakref

import numpy as np
import torch

from . import _lib

_tls = threading.local()  # .contexts: device index -> Context, one set per thread, freed when the thread ends
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # (device index) -> the current stream's handle


class NotPositiveDefinite(ValueError):
    """The reference's ``reject`` in ``gp_evaluate_rfft(2)_scale`` (fft1.stan:19-21, fft2.stan:21-23)."""


class Context:
    """One library context per (device, calling thread); follows torch's current stream."""

    def __init__(self, device: int):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("gptools_amd needs a GPU: the HIP path is the only path")
        self.device = device
        with torch.cuda.device(device):
            self.stream = torch.cuda.current_stream().cuda_stream
        h = ctypes.c_void_p()
        rc = self.lib.gpt_create(device, ctypes.c_void_p(self.stream), ctypes.byref(h))
        if rc != 0:
            raise RuntimeError(f"gpt_create failed with status {rc}")
        self.handle = h
        self._fns = {}
        # the context owns device memory (workspace, twiddles, staging): give it back when the Context object goes
        # (its thread ended, or close() was called)
        self._finalizer = weakref.finalize(self, self.lib.gpt_destroy, h)
        self._finalizer.atexit = False  # at interpreter exit the process teardown frees everything

    def close(self):
        """Free the library context (workspace and tables on the device) now; the object must not be used afterwards."""
        self._finalizer()
        self.handle = None

    def sync_stream(self):
        # this runs on every call: the raw stream handle, without building a torch.cuda.Stream object around it
        cur = _raw_stream(self.device) if _raw_stream else torch.cuda.current_stream(self.device).cuda_stream
        if cur != self.stream:
            rc = self.lib.gpt_set_stream(self.handle, ctypes.c_void_p(cur))
            if rc != 0:
                raise RuntimeError(self.error())
            self.stream = cur

    def error(self) -> str:
        return self.lib.gpt_last_error(self.handle).decode()

    def call(self, name, *args):
        self.sync_stream()
        fn = self._fns.get(name)
        if fn is None:
            fn = self._fns[name] = getattr(self.lib, name)
        rc = fn(self.handle, *args)
        if rc == 0:
            return
        if rc == _lib.GPT_ERR_SHAPE:
            raise ValueError(self.error())
        if rc == _lib.GPT_ERR_UNSUPPORTED:
            raise NotImplementedError(self.error())
        if rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {self.error()}")


def get_context(device=None) -> Context:
    if isinstance(device, torch.device):
        device = device.index
    if device is None:
        device = torch.cuda.current_device()
    try:
        ctx = _tls.contexts[device]  # the common case: one dictionary look-up per call
        if ctx.handle is not None:
            return ctx
    except (AttributeError, KeyError):
        pass
    table = getattr(_tls, "contexts", None)
    if table is None:
        table = _tls.contexts = {}
    ctx = table[device] = Context(device)
    return ctx


def close_contexts():
    """Free this thread's library contexts (device workspaces); later calls create new ones."""
    for ctx in getattr(_tls, "contexts", {}).values():
        if ctx.handle is not None:
            ctx.close()
    _tls.contexts = {}


def path_for_shape(height: int, width: int) -> int:
    """0 small (one workgroup per lattice), 1 power-of-two three-pass, 2 generic three-pass, 3 any-length (four-step /
    Bluestein line transforms), -1 none."""
    return _lib.load().gpt_path_for_shape(height, width)


# ---------------------------------------------------------------------------------------------------
def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _dev(x, like=None):
    if isinstance(x, torch.Tensor) and x.is_cuda and x.dtype is torch.float64:
        return x  # (the sampler's case: nothing to convert)
    if not isinstance(x, torch.Tensor):
        # through numpy: torch.as_tensor would round Python floats to float32
        a = np.asarray(x)
        x = torch.as_tensor(a if np.iscomplexobj(a) else a.astype(np.float64, copy=False))
    if x.dtype not in (torch.float64, torch.complex128):
        x = x.to(torch.complex128 if x.is_complex() else torch.float64)
    if not x.is_cuda:
        x = x.to(like.device if like is not None else torch.device("cuda", torch.cuda.current_device()))
    return x


def _batched(x, core_ndim, batch_shape=None):
    """-> (contiguous tensor, batch_shape, batch stride in elements or 0 when shared)."""
    x = x.contiguous()
    bshape = tuple(x.shape[: x.ndim - core_ndim])
    core = 1
    for s in x.shape[x.ndim - core_ndim:]:
        core *= s
    if batch_shape is not None and bshape != tuple(batch_shape):
        if bshape == ():
            return x, tuple(batch_shape), 0
        x = x.expand(*batch_shape, *x.shape[x.ndim - core_ndim:]).contiguous()
        bshape = tuple(batch_shape)
    return x, bshape, core


def _check_pd(min_eig, check):
    if check and min_eig.numel():
        mn = float(min_eig.min())
        if mn < 0:
            raise NotPositiveDefinite(
                f"covariance matrix is not positive-definite (minimum eigenvalue is {mn})"
            )


def _numel(shape):
    return int(math.prod(shape))


def _prep(main, loc, cov, dim1):
    """Normalise (main, loc, cov) to batched contiguous device tensors; returns a dict of geometry."""
    main = _dev(main)
    core = 1 if dim1 else 2
    # batch dimensions broadcast like the reference's NumPy mirror (fft2.py: `(..., H, W)` arrays): loc / cov with MORE
    # batch dimensions than the lattice argument expand it
    mb = main.shape[:-core]
    others = [torch.Size(t.shape[:-core]) for t in (loc, cov)
              if hasattr(t, "shape") and len(t.shape) > core and t.shape[:-core] != mb]
    if others:
        full = torch.broadcast_shapes(torch.Size(main.shape[:-core]), *others)
        if tuple(full) != tuple(main.shape[:-core]):
            main = main.expand(*full, *main.shape[-core:])
    main, bshape, m_bs = _batched(main, core)
    H = 1 if dim1 else main.shape[-2]
    W = main.shape[-1]
    g = dict(H=H, W=W, Wf=W // 2 + 1, bshape=bshape, B=_numel(bshape), main=main, m_bs=m_bs, dim1=int(dim1))
    if loc is not None:
        loc = _dev(loc, main)
        if loc.shape[-core:] != main.shape[-core:]:
            loc = loc.expand(*main.shape[-core:])
        g["loc"], _, g["l_bs"] = _batched(loc, core, bshape)
    else:
        g["loc"], g["l_bs"] = None, 0
    if cov is not None:
        cov = _dev(cov, main)
        if cov.is_complex():  # the reference's tests pass np.fft.rfft(cov) unchanged (test_fft.py:45-46)
            cov = cov.real
        if tuple(cov.shape[-core:]) != ((g["Wf"],) if dim1 else (H, g["Wf"])):
            raise ValueError(f"cov_rfft has shape {tuple(cov.shape)}; expected (..., {H}, {g['Wf']})")
        g["cov"], _, g["c_bs"] = _batched(cov, core, bshape)
    return g


def _new(g, *shape):
    return torch.empty((*g["bshape"], *shape), dtype=torch.float64, device=g["main"].device)


# ---- real-FFT layer: util.stan:255-330 -------------------------------------------------------------
def rfft2(y):
    """util.stan:302-304."""
    g = _prep(y, None, None, False)
    out = torch.empty((*g["bshape"], g["H"], g["Wf"]), dtype=torch.complex128, device=g["main"].device)
    get_context(out.device).call("gpt_rfft2", _p(g["main"]), g["H"], g["W"], g["B"], _p(out))
    return out


def rfft(y):
    """util.stan:263-265."""
    y = _dev(y)
    return rfft2(y.unsqueeze(-2)).squeeze(-2)


def inv_rfft2(z, m):
    """util.stan:315-330 (``m`` = number of columns of the signal)."""
    z = _dev(z).to(torch.complex128).contiguous()
    H, Wf = z.shape[-2:]
    if Wf != m // 2 + 1:
        raise ValueError(f"expected complex matrix with {m // 2 + 1} columns but got {Wf}")
    bshape = tuple(z.shape[:-2])
    out = torch.empty((*bshape, H, m), dtype=torch.float64, device=z.device)
    get_context(z.device).call("gpt_inv_rfft2", _p(z), H, m, _numel(bshape), _p(out))
    return out


def inv_rfft(z, n):
    """util.stan:291-293."""
    z = _dev(z)
    return inv_rfft2(z.unsqueeze(-2), n).squeeze(-2)


def expand_rfft(y, n):
    """util.stan:271-281 (rejects a size mismatch, util.stan:274-276)."""
    y = _dev(y).to(torch.complex128).contiguous()
    if y.shape[-1] != n // 2 + 1:
        raise ValueError(f"expected complex vector with {n // 2 + 1} elements but got {y.shape[-1]}")
    bshape = tuple(y.shape[:-1])
    out = torch.empty((*bshape, n), dtype=torch.complex128, device=y.device)
    get_context(y.device).call("gpt_expand_rfft", _p(y), n, _numel(bshape), _p(out))
    return out


# ---- GP layer --------------------------------------------------------------------------------------
def _scale(cov, width, dim1, check):
    cov = _dev(cov)
    if cov.is_complex():
        cov = cov.real
    cov, bshape, c_bs = _batched(cov, 1 if dim1 else 2)
    H = 1 if dim1 else cov.shape[-2]
    if cov.shape[-1] != width // 2 + 1:
        raise ValueError(f"expected {width // 2 + 1} columns but got {cov.shape[-1]}")
    out = torch.empty_like(cov)
    mn = torch.empty(bshape, dtype=torch.float64, device=cov.device)
    get_context(cov.device).call("gpt_gp_evaluate_rfft2_scale", _p(cov), c_bs, H, width, _numel(bshape),
                                 int(dim1), _p(out), _p(mn))
    _check_pd(mn, check)
    return out


def gp_evaluate_rfft_scale(cov_rfft, n, check=True):
    """fft1.stan:14-30."""
    return _scale(cov_rfft, n, True, check)


def gp_evaluate_rfft2_scale(cov_rfft2, width, check=True):
    """fft2.stan:13-40."""
    return _scale(cov_rfft2, width, False, check)


def _unpack(z, width, dim1):
    z = _dev(z).to(torch.complex128).contiguous()
    H = 1 if dim1 else z.shape[-2]
    bshape = tuple(z.shape[: -1 if dim1 else -2])
    out = torch.empty((*bshape, *((width,) if dim1 else (H, width))), dtype=torch.float64, device=z.device)
    get_context(z.device).call("gpt_gp_unpack_rfft2", _p(z), H, width, _numel(bshape), int(dim1), _p(out))
    return out


def gp_unpack_rfft(x, n):
    """fft1.stan:44-51."""
    return _unpack(x, n, True)


def gp_unpack_rfft2(z, m):
    """fft2.stan:55-74."""
    return _unpack(z, m, False)


def _pack(z, dim1):
    z = _dev(z).contiguous()
    W = z.shape[-1]
    H = 1 if dim1 else z.shape[-2]
    bshape = tuple(z.shape[: -1 if dim1 else -2])
    out = torch.empty((*bshape, *((W // 2 + 1,) if dim1 else (H, W // 2 + 1))), dtype=torch.complex128,
                      device=z.device)
    get_context(z.device).call("gpt_gp_pack_rfft2", _p(z), H, W, _numel(bshape), int(dim1), _p(out))
    return out


def gp_pack_rfft(z):
    """fft1.stan:108-118."""
    return _pack(z, True)


def gp_pack_rfft2(z):
    """fft2.stan:85-100."""
    return _pack(z, False)


def _transform(name, main, loc, cov, dim1, check):
    g = _prep(main, loc, cov, dim1)
    out = torch.empty_like(g["main"])
    mn = _new(g)
    get_context(out.device).call(name, _p(g["main"]), g["m_bs"], _p(g["loc"]), g["l_bs"], _p(g["cov"]), g["c_bs"],
                                 g["H"], g["W"], g["B"], g["dim1"], _p(out), _p(mn))
    _check_pd(mn, check)
    return out


def gp_rfft(y, loc, cov_rfft, check=True):
    """fft1.stan:64-66."""
    return _transform("gpt_gp_rfft2", y, loc, cov_rfft, True, check)


def gp_rfft2(y, loc, cov_rfft2, check=True):
    """fft2.stan:112-114."""
    return _transform("gpt_gp_rfft2", y, loc, cov_rfft2, False, check)


def gp_inv_rfft(z, loc, cov_rfft, check=True):
    """fft1.stan:139-143."""
    return _transform("gpt_gp_inv_rfft2", z, loc, cov_rfft, True, check)


def gp_inv_rfft2(z, loc, cov_rfft2, check=True):
    """fft2.stan:126-130."""
    return _transform("gpt_gp_inv_rfft2", z, loc, cov_rfft2, False, check)


def _ladj(cov, width, dim1, check):
    cov = _dev(cov)
    if cov.is_complex():
        cov = cov.real
    cov, bshape, c_bs = _batched(cov, 1 if dim1 else 2)
    H = 1 if dim1 else cov.shape[-2]
    out = torch.empty(bshape, dtype=torch.float64, device=cov.device)
    mn = torch.empty(bshape, dtype=torch.float64, device=cov.device)
    get_context(cov.device).call("gpt_gp_rfft2_log_abs_det_jac", _p(cov), c_bs, H, width, _numel(bshape),
                                 int(dim1), _p(out), _p(mn))
    _check_pd(mn, check)
    return out


def gp_rfft_log_abs_det_jac(cov_rfft, n, check=True):
    """fft1.stan:78-82."""
    return _ladj(cov_rfft, n, True, check)


def gp_rfft2_log_abs_det_jac(cov_rfft2, width, check=True):
    """fft2.stan:143-173."""
    return _ladj(cov_rfft2, width, False, check)


def _ladj_vjp(cov, width, dim1):
    cov = _dev(cov)
    cov, bshape, c_bs = _batched(cov, 1 if dim1 else 2)
    H = 1 if dim1 else cov.shape[-2]
    out = torch.empty_like(cov)
    get_context(cov.device).call("gpt_gp_rfft2_log_abs_det_jac_vjp", _p(cov), c_bs, H, width, _numel(bshape),
                                 int(dim1), _p(out))
    return out


def gp_rfft_log_abs_det_jac_vjp(cov_rfft, n):
    return _ladj_vjp(cov_rfft, n, True)


def gp_rfft2_log_abs_det_jac_vjp(cov_rfft2, width):
    return _ladj_vjp(cov_rfft2, width, False)


def _lpdf(y, loc, cov, dim1, check):
    g = _prep(y, loc, cov, dim1)
    out, mn = _new(g), _new(g)
    get_context(out.device).call("gpt_gp_rfft2_lpdf", _p(g["main"]), g["m_bs"], _p(g["loc"]), g["l_bs"],
                                 _p(g["cov"]), g["c_bs"], g["H"], g["W"], g["B"], g["dim1"], _p(out), _p(mn))
    _check_pd(mn, check)
    return out


def gp_rfft_lpdf(y, loc, cov_rfft, check=True):
    """fft1.stan:96-101."""
    return _lpdf(y, loc, cov_rfft, True, check)


def gp_rfft2_lpdf(y, loc, cov_rfft2, check=True):
    """fft2.stan:189-192."""
    return _lpdf(y, loc, cov_rfft2, False, check)


def _lpdf_vjp(y, loc, cov, dim1, check, need_cov_grad, out=None):
    g = _prep(y, loc, cov, dim1)
    if out is None:
        lp, mn = _new(g), _new(g)
        gy = torch.empty_like(g["main"])
        gc = _new(g, *((g["Wf"],) if dim1 else (g["H"], g["Wf"]))) if need_cov_grad else None
    else:
        lp, gy, gc, mn = out
    get_context(lp.device).call("gpt_gp_rfft2_lpdf_vjp", _p(g["main"]), g["m_bs"], _p(g["loc"]), g["l_bs"],
                                _p(g["cov"]), g["c_bs"], g["H"], g["W"], g["B"], g["dim1"], _p(lp), _p(gy),
                                _p(gc), _p(mn))
    _check_pd(mn, check)
    return lp, gy, gc


def gp_rfft_lpdf_vjp(y, loc, cov_rfft, check=True, need_cov_grad=True):
    """Fused value + gradients of fft1.stan:96-101: ``(lpdf, d/dy, d/dcov)``; ``d/dloc = -d/dy``."""
    return _lpdf_vjp(y, loc, cov_rfft, True, check, need_cov_grad)


def gp_rfft2_lpdf_vjp(y, loc, cov_rfft2, check=True, need_cov_grad=True, out=None):
    """Fused value + gradients of fft2.stan:189-192: ``(lpdf, d/dy, d/dcov)``; ``d/dloc = -d/dy``.

    ``out=(lpdf, grad_y, grad_cov, min_eig)`` reuses caller-owned output tensors (benchmark loops).
    """
    return _lpdf_vjp(y, loc, cov_rfft2, False, check, need_cov_grad, out)


def _inv_vjp(z, cov, fbar, dim1):
    g = _prep(z, fbar, cov, dim1)  # fbar rides in the `loc` slot of _prep (same shape rules)
    zbar = torch.empty_like(g["main"])
    cbar = _new(g, *((g["Wf"],) if dim1 else (g["H"], g["Wf"])))
    get_context(zbar.device).call("gpt_gp_inv_rfft2_vjp", _p(g["main"]), g["m_bs"], _p(g["cov"]), g["c_bs"],
                                  _p(g["loc"]), g["l_bs"], g["H"], g["W"], g["B"], g["dim1"], _p(zbar), _p(cbar))
    return zbar, cbar


def gp_inv_rfft_vjp(z, cov_rfft, fbar):
    """Reverse mode of fft1.stan:139-143: ``fbar -> (zbar, covbar)``; ``locbar = fbar``."""
    return _inv_vjp(z, cov_rfft, fbar, True)


def gp_inv_rfft2_vjp(z, cov_rfft2, fbar):
    """Reverse mode of fft2.stan:126-130: ``fbar -> (zbar, covbar)``; ``locbar = fbar``."""
    return _inv_vjp(z, cov_rfft2, fbar, False)


def _fwd_vjp(y, loc, cov, zbar, dim1):
    g = _prep(y, loc, cov, dim1)
    zb = _dev(zbar, g["main"])
    zb, _, zb_bs = _batched(zb, 1 if dim1 else 2, g["bshape"])
    ybar = torch.empty_like(g["main"])
    cbar = _new(g, *((g["Wf"],) if dim1 else (g["H"], g["Wf"])))
    get_context(ybar.device).call("gpt_gp_rfft2_vjp", _p(g["main"]), g["m_bs"], _p(g["loc"]), g["l_bs"],
                                  _p(g["cov"]), g["c_bs"], _p(zb), zb_bs, g["H"], g["W"], g["B"], g["dim1"],
                                  _p(ybar), _p(cbar))
    return ybar, cbar


def gp_rfft_vjp(y, loc, cov_rfft, zbar):
    """Reverse mode of fft1.stan:64-66: ``zbar -> (ybar, covbar)``; ``locbar = -ybar``."""
    return _fwd_vjp(y, loc, cov_rfft, zbar, True)


def gp_rfft2_vjp(y, loc, cov_rfft2, zbar):
    """Reverse mode of fft2.stan:112-114: ``zbar -> (ybar, covbar)``; ``locbar = -ybar``."""
    return _fwd_vjp(y, loc, cov_rfft2, zbar, False)


# ---- observation models on the latent field ----------------------------------
OBS_FAMILIES = {"normal": 0, "poisson_log": 1, "neg_binomial_2_log": 2}


def _obs_lpdf_vjp(family, z, loc, cov, obs, theta, prior, dim1, check, need_cov_grad, need_f=True):
    g = _prep(z, loc, cov, dim1)
    core = 1 if dim1 else 2
    obs = _dev(obs, g["main"])
    obs, _, o_bs = _batched(obs, core, g["bshape"])
    oh = 1 if dim1 else obs.shape[-2]
    ow = obs.shape[-1]
    fam = OBS_FAMILIES[family]
    th = None
    if fam != 1:
        if theta is None:
            raise ValueError(f"the {family} family needs theta")
        th = _dev(theta, g["main"]).expand(g["bshape"]).contiguous()
        # Stan Math rejects these (normal_lpdf "Scale parameter", neg_binomial_2_log_lpmf "Precision parameter")
        if not bool(((th > 0) & torch.isfinite(th)).all()):
            raise ValueError(("Scale" if fam == 0 else "Precision") + " parameter must be positive finite")
    lp, mn, gls = _new(g), _new(g), _new(g)
    gth = _new(g) if th is not None else None
    f = torch.empty_like(g["main"]) if need_f else None
    gz = torch.empty_like(g["main"])
    gl = torch.empty_like(g["main"]) if need_f else None
    gc = _new(g, *((g["Wf"],) if dim1 else (g["H"], g["Wf"]))) if need_cov_grad else None
    get_context(lp.device).call("gpt_gp_inv_rfft2_obs_lpdf_vjp", fam, _p(g["main"]), g["m_bs"], _p(g["loc"]), g["l_bs"],
                                _p(g["cov"]), g["c_bs"], _p(obs), o_bs, oh, ow, _p(th), int(bool(prior)), g["H"], g["W"],
                                g["B"], g["dim1"], _p(lp), _p(f), _p(gz), _p(gl), _p(gls), _p(gc), _p(gth), _p(mn))
    _check_pd(mn, check)
    return dict(lp=lp, f=f, grad_z=gz, grad_loc=gl, grad_loc_sum=gls, grad_cov=gc, grad_theta=gth)


def gp_inv_rfft2_obs_lpdf_vjp(family, z, loc, cov_rfft2, obs, theta=None, prior=True, check=True, need_cov_grad=True,
                              need_f=True):
    """Log density and gradients of a non-centred docs model on the device: ``f = gp_inv_rfft2(z, loc, cov)``
    (fft2.stan:126-130), ``lp = sum family_lpdf(obs | f, theta) [+ std_normal_lpdf(z)]`` over the ``obs`` corner of the
    (padded) lattice, NaN / negative counts masked (docs/trees/trees.stan:31-49). ``family``: ``"normal"`` (theta =
    standard deviation), ``"poisson_log"``, ``"neg_binomial_2_log"`` (theta = precision; trees.stan passes 1/kappa).
    Returns a dict: ``lp, f, grad_z, grad_loc`` (= d lp / d f), ``grad_loc_sum, grad_cov, grad_theta``. With
    ``need_f=False`` the field and its gradient are not returned (a sampler with a scalar ``loc`` needs neither) and,
    on power-of-two lattices, never reach HBM."""
    return _obs_lpdf_vjp(family, z, loc, cov_rfft2, obs, theta, prior, False, check, need_cov_grad, need_f)


def gp_inv_rfft_obs_lpdf_vjp(family, z, loc, cov_rfft, obs, theta=None, prior=True, check=True, need_cov_grad=True,
                             need_f=True):
    """1-D sibling: docs/padding/padded.stan:29-34 (``obs = y[:observe_first]``),
    docs/poisson_regression/poisson_regression_fourier_non_centered.stan:20-26."""
    return _obs_lpdf_vjp(family, z, loc, cov_rfft, obs, theta, prior, True, check, need_cov_grad, need_f)


# ---- spectral kernels ------------------------------------------------------------------------------
def _params(dim1, sigma, length_scale, period, device=None):
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    parts = [torch.as_tensor(sigma, dtype=torch.float64, device=device)]
    ls = torch.as_tensor(length_scale, dtype=torch.float64, device=device)
    per = torch.as_tensor(period, dtype=torch.float64, device=device)
    if dim1:
        bshape = torch.broadcast_shapes(parts[0].shape, ls.shape, per.shape)
        cols = [parts[0].expand(bshape), ls.expand(bshape), per.expand(bshape)]
    else:
        bshape = torch.broadcast_shapes(parts[0].shape, ls.shape[:-1], per.shape[:-1])
        ls = ls.expand(*bshape, 2)
        per = per.expand(*bshape, 2)
        cols = [parts[0].expand(bshape), ls[..., 0], ls[..., 1], per[..., 0], per[..., 1]]
    return torch.stack(cols, dim=-1).contiguous(), tuple(bshape)


def _lpdf_kernel_vjp(y, loc, kind, nu, sigma, length_scale, period, nugget, dim1, check, out=None):
    g = _prep(y, loc, None, dim1)
    par, pshape = _params(dim1, sigma, length_scale, period, g["main"].device)
    if pshape != g["bshape"]:
        par = par.expand(*g["bshape"], par.shape[-1]).contiguous()
    if out is None:
        lp, mn = _new(g), _new(g)
        gy = torch.empty_like(g["main"])
        pbar = torch.empty_like(par)
    else:
        lp, gy, pbar, mn = out
    get_context(lp.device).call("gpt_gp_rfft2_lpdf_kernel_vjp", int(kind != "expquad"), float(nu), _p(g["main"]),
                                g["m_bs"], _p(g["loc"]), g["l_bs"], _p(par), float(nugget), g["H"], g["W"], g["B"],
                                g["dim1"], _p(lp), _p(gy), _p(pbar), _p(mn))
    _check_pd(mn, check)
    if dim1:
        return lp, gy, (pbar[..., 0], pbar[..., 1], pbar[..., 2])
    return lp, gy, (pbar[..., 0], pbar[..., 1:3], pbar[..., 3:5])


def gp_rfft2_lpdf_kernel_vjp(y, loc, kind, nu, sigma, length_scale, period, nugget=0.0, check=True, out=None):
    """``gp_rfft2_lpdf(y | loc, kernel + nugget)`` (fft2.stan:189-192 over :206-247) without materialising the spectrum:
    ``(lpdf, d/dy, (d/dsigma, d/dlength_scale, d/dperiod))``. ``kind``: ``"expquad"`` or ``"matern"`` (``nu``).
    ``out=(lpdf, grad_y, params_bar[..., 5], min_eig)`` reuses caller-owned output tensors (sampler / benchmark loops)."""
    return _lpdf_kernel_vjp(y, loc, kind, nu, sigma, length_scale, period, nugget, False, check, out)


def gp_rfft_lpdf_kernel_vjp(y, loc, kind, nu, sigma, length_scale, period, nugget=0.0, check=True):
    """1-D sibling (fft1.stan:96-101 over :154-176)."""
    return _lpdf_kernel_vjp(y, loc, kind, nu, sigma, length_scale, period, nugget, True, check)


def _obs_lpdf_kernel_vjp(family, z, loc, kind, nu, sigma, length_scale, period, nugget, obs, theta, prior, dim1, check):
    g = _prep(z, loc, None, dim1)
    core = 1 if dim1 else 2
    par, pshape = _params(dim1, sigma, length_scale, period, g["main"].device)
    if pshape != g["bshape"]:
        par = par.expand(*g["bshape"], par.shape[-1]).contiguous()
    obs = _dev(obs, g["main"])
    obs, _, o_bs = _batched(obs, core, g["bshape"])
    oh = 1 if dim1 else obs.shape[-2]
    ow = obs.shape[-1]
    fam = OBS_FAMILIES[family]
    th = None
    if fam != 1:
        if theta is None:
            raise ValueError(f"the {family} family needs theta")
        th = _dev(theta, g["main"]).expand(g["bshape"]).contiguous()
        if not bool(((th > 0) & torch.isfinite(th)).all()):
            raise ValueError(("Scale" if fam == 0 else "Precision") + " parameter must be positive finite")
    lp, mn, gls = _new(g), _new(g), _new(g)
    gth = _new(g) if th is not None else None
    gz = torch.empty_like(g["main"])
    pbar = torch.empty_like(par)
    get_context(lp.device).call("gpt_gp_inv_rfft2_obs_lpdf_kernel_vjp", fam, int(kind != "expquad"), float(nu),
                                _p(g["main"]), g["m_bs"], _p(g["loc"]), g["l_bs"], _p(par), float(nugget), _p(obs), o_bs, oh,
                                ow, _p(th), int(bool(prior)), g["H"], g["W"], g["B"], g["dim1"], _p(lp), _p(gz), _p(gls),
                                _p(pbar), _p(gth), _p(mn))
    _check_pd(mn, check)
    kbar = (pbar[..., 0], pbar[..., 1], pbar[..., 2]) if dim1 else (pbar[..., 0], pbar[..., 1:3], pbar[..., 3:5])
    return dict(lp=lp, grad_z=gz, grad_loc_sum=gls, grad_kernel=kbar, grad_theta=gth)


def gp_inv_rfft2_obs_lpdf_kernel_vjp(family, z, loc, kind, nu, sigma, length_scale, period, obs, theta=None, nugget=0.0,
                                     prior=True, check=True):
    """The non-centred model density of :func:`gp_inv_rfft2_obs_lpdf_vjp` with the spectrum given by its kernel
    (fft2.stan:206-247 + ``nugget``, as docs/trees/trees.stan:26-29 builds it on every gradient evaluation) instead of
    an array: one call, and gradients down to ``(sigma, length_scale[2], period[2])``. ``kind``: ``"expquad"`` or
    ``"matern"`` (``nu``). Returns a dict: ``lp, grad_z, grad_loc_sum, grad_kernel = (d/dsigma, d/dlength_scale,
    d/dperiod), grad_theta``. Lattices that fit one workgroup never hold the spectrum anywhere."""
    return _obs_lpdf_kernel_vjp(family, z, loc, kind, nu, sigma, length_scale, period, nugget, obs, theta, prior, False, check)


def gp_inv_rfft_obs_lpdf_kernel_vjp(family, z, loc, kind, nu, sigma, length_scale, period, obs, theta=None, nugget=0.0,
                                    prior=True, check=True):
    """1-D sibling (fft1.stan:154-176 feeding fft1.stan:67-70; padded.stan, poisson_regression_fourier_non_centered.stan)."""
    return _obs_lpdf_kernel_vjp(family, z, loc, kind, nu, sigma, length_scale, period, nugget, obs, theta, prior, True, check)


def _spec(kind, nu, height, width, sigma, length_scale, period, dim1):
    par, bshape = _params(dim1, sigma, length_scale, period)
    Wf = width // 2 + 1
    out = torch.empty((*bshape, *((Wf,) if dim1 else (height, Wf))), dtype=torch.float64, device=par.device)
    ctx = get_context(par.device)
    if kind == "expquad":
        ctx.call("gpt_gp_periodic_exp_quad_cov_rfft2", height, width, _p(par), _numel(bshape), int(dim1), _p(out))
    else:
        ctx.call("gpt_gp_periodic_matern_cov_rfft2", float(nu), height, width, _p(par), _numel(bshape), int(dim1),
                 _p(out))
    return out


def gp_periodic_exp_quad_cov_rfft(n, sigma, length_scale, period):
    """fft1.stan:154-158."""
    return _spec("expquad", 0.0, 1, n, sigma, length_scale, period, True)


def gp_periodic_matern_cov_rfft(nu, n, sigma, length_scale, period):
    """fft1.stan:170-176."""
    return _spec("matern", nu, 1, n, sigma, length_scale, period, True)


def gp_periodic_exp_quad_cov_rfft2(height, width, sigma, length_scale, period):
    """fft2.stan:206-213."""
    return _spec("expquad", 0.0, height, width, sigma, length_scale, period, False)


def gp_periodic_matern_cov_rfft2(nu, height, width, sigma, length_scale, period):
    """fft2.stan:228-247."""
    return _spec("matern", nu, height, width, sigma, length_scale, period, False)


def _spec_vjp(kind, nu, height, width, sigma, length_scale, period, covbar, dim1):
    par, bshape = _params(dim1, sigma, length_scale, period)
    cb = _dev(covbar, par)
    cb, _, cb_bs = _batched(cb, 1 if dim1 else 2, bshape)
    npar = 3 if dim1 else 5
    out = torch.empty((*bshape, npar), dtype=torch.float64, device=par.device)
    ctx = get_context(par.device)
    if kind == "expquad":
        ctx.call("gpt_gp_periodic_exp_quad_cov_rfft2_vjp", height, width, _p(par), _p(cb), cb_bs, _numel(bshape),
                 int(dim1), _p(out))
    else:
        ctx.call("gpt_gp_periodic_matern_cov_rfft2_vjp", float(nu), height, width, _p(par), _p(cb), cb_bs,
                 _numel(bshape), int(dim1), _p(out))
    if dim1:
        return out[..., 0], out[..., 1], out[..., 2]
    return out[..., 0], out[..., 1:3], out[..., 3:5]


def gp_periodic_exp_quad_cov_rfft_vjp(n, sigma, length_scale, period, covbar):
    """``covbar -> (sigma_bar, length_scale_bar, period_bar)`` for fft1.stan:154-158."""
    return _spec_vjp("expquad", 0.0, 1, n, sigma, length_scale, period, covbar, True)


def gp_periodic_matern_cov_rfft_vjp(nu, n, sigma, length_scale, period, covbar):
    return _spec_vjp("matern", nu, 1, n, sigma, length_scale, period, covbar, True)


def gp_periodic_exp_quad_cov_rfft2_vjp(height, width, sigma, length_scale, period, covbar):
    """``covbar -> (sigma_bar, length_scale_bar[2], period_bar[2])`` for fft2.stan:206-213."""
    return _spec_vjp("expquad", 0.0, height, width, sigma, length_scale, period, covbar, False)


def gp_periodic_matern_cov_rfft2_vjp(nu, height, width, sigma, length_scale, period, covbar):
    return _spec_vjp("matern", nu, height, width, sigma, length_scale, period, covbar, False)


# ---- names of the reference's NumPy mirror (python/util/gptools/util/fft/fft1.py, fft2.py) -----------
def transform_rfft(y, loc, *, cov_rfft):
    return gp_rfft(y, loc, cov_rfft)


def transform_irfft(z, loc, *, cov_rfft):
    return gp_inv_rfft(z, loc, cov_rfft)


def evaluate_log_prob_rfft(y, loc, *, cov_rfft):
    return gp_rfft_lpdf(y, loc, cov_rfft)


def evaluate_rfft_log_abs_det_jac(size, *, cov_rfft):
    return gp_rfft_log_abs_det_jac(cov_rfft, size)


def transform_rfft2(y, loc, *, cov_rfft2):
    return gp_rfft2(y, loc, cov_rfft2)


def transform_irfft2(z, loc, *, cov_rfft2):
    return gp_inv_rfft2(z, loc, cov_rfft2)


def evaluate_log_prob_rfft2(y, loc, *, cov_rfft2):
    return gp_rfft2_lpdf(y, loc, cov_rfft2)


def evaluate_rfft2_log_abs_det_jac(width, *, cov_rfft2):
    return gp_rfft2_log_abs_det_jac(cov_rfft2, width)
