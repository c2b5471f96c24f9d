"""
gptools_amd -- the Fourier-method Gaussian-process path of onnela-lab/gptools as hand-written HIP kernels
for AMD MI355X (gfx950), behind the reference's own function names.

    from gptools_amd import fft
    lpdf, grad_y, grad_cov = fft.gp_rfft2_lpdf_vjp(y, loc, cov_rfft2)     # device tensors

The compute lives in ``libgptools_hip.so`` (sources in ``gptools_amd/csrc``, C ABI in
``include/gptools_hip.h``). There is no CPU fallback: importing :mod:`gptools_amd.fft` needs the built
library and calling it needs a GPU.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def __getattr__(name):
    if name == "fft":
        import importlib

        return importlib.import_module(".fft", __name__)
    raise AttributeError(name)
