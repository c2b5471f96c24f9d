"""
Cross-chain batching: the chains of one sampler process hand their `gp_rfft(2)_lpdf`
evaluations to a shared :class:`Batcher`; every call blocks until all active chains have called and ONE fused launch
evaluates them all (C ABI ``gpt_batcher_*``, include/gptools_hip.h). Host (NumPy) arrays in and out: this is the
boundary a Stan chain sees. The reference has no counterpart: there every chain evaluates on its own CPU core.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .fft import NotPositiveDefinite


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Batcher:
    """``nslots`` chains, one lattice shape. ``width`` alone selects the 1-D functions (fft1.stan). ``timeout``
    (seconds, 0 = none): a call that waits longer for the other chains raises TimeoutError instead of blocking for ever
    behind a chain that ended without ``leave``."""

    def __init__(self, nslots: int, height: int | None, width: int, device: int = 0, col_major: bool = False,
                 timeout: float = 0.0):
        self.lib = _lib.load()
        self.dim1 = height is None
        self.shape = (width,) if self.dim1 else (height, width)
        self.cov_shape = (width // 2 + 1,) if self.dim1 else (height, width // 2 + 1)
        self.order = "F" if col_major else "C"
        self.nslots = nslots
        h = ctypes.c_void_p()
        rc = self.lib.gpt_batcher_create(device, nslots, _lib.GPT_COL_MAJOR if col_major else _lib.GPT_ROW_MAJOR,
                                         1 if self.dim1 else height, width, int(self.dim1), ctypes.byref(h))
        if rc != 0:
            raise (ValueError if rc == _lib.GPT_ERR_SHAPE else RuntimeError)(f"gpt_batcher_create failed ({rc})")
        self.handle = h
        if timeout:
            self.lib.gpt_batcher_set_timeout(h, float(timeout))

    def _arr(self, x, shape):
        x = np.asarray(x, dtype=np.float64)
        if x.shape != shape:
            raise ValueError(f"expected shape {shape}, got {x.shape}")
        return np.require(x, requirements=["F_CONTIGUOUS" if self.order == "F" else "C_CONTIGUOUS", "ALIGNED"])

    def lpdf(self, slot, y, loc, cov, grad=True, grad_cov=True):
        """Blocks until every active slot has called. Returns ``lpdf`` or ``(lpdf, d/dy, d/dcov)``."""
        y = self._arr(y, self.shape)
        loc = None if loc is None else self._arr(loc, self.shape)
        cov = self._arr(cov, self.cov_shape)
        lp = ctypes.c_double(0.0)
        gy = np.empty(self.shape, order=self.order) if grad else None
        gc = np.empty(self.cov_shape, order=self.order) if (grad and grad_cov) else None
        rc = self.lib.gpt_batcher_lpdf_host(self.handle, slot, _ptr(y), _ptr(loc), _ptr(cov), ctypes.byref(lp),
                                            _ptr(gy), _ptr(gc))
        if rc == _lib.GPT_ERR_NOT_PD:
            mn = self.lib.gpt_batcher_min_eig(self.handle, slot)
            raise NotPositiveDefinite(f"covariance matrix is not positive-definite (minimum eigenvalue is {mn})")
        if rc == _lib.GPT_ERR_SHAPE:
            raise ValueError(f"slot {slot} is out of range or has left the batcher")
        if rc == _lib.GPT_ERR_TIMEOUT:
            raise TimeoutError(self.lib.gpt_batcher_last_error(self.handle).decode())
        if rc != 0:
            raise RuntimeError(f"batched evaluation failed ({rc}): {self.lib.gpt_batcher_last_error(self.handle).decode()}")
        return (lp.value, gy, gc) if grad else lp.value

    def leave(self, slot):
        """The chain of `slot` stops calling: the others no longer wait for it."""
        self.lib.gpt_batcher_leave(self.handle, slot)

    @property
    def launches(self):
        return self.lib.gpt_batcher_launches(self.handle)

    def close(self):
        if self.handle:
            self.lib.gpt_batcher_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
