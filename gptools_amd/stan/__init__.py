"""
Drop-in twin of the reference's ``gptools.stan`` wrapper (python/stan/gptools/stan/__init__.py:8-87):
``get_include()`` and ``compile_model(...)`` with the same signature, pointing cmdstan at the
declarations-only ``gptools/*.stan`` files of this package, ``stanc --allow-undefined`` and the C++ user
header ``include/gptools_hip.hpp`` that binds the Stan functions to ``libgptools_hip.so``.

cmdstanpy/cmdstan are not part of this build image; importing this module works without them and the
functions raise an informative error if they are called without cmdstanpy.
"""
import logging
import os
from typing import Any, Dict, Optional, Type, Union

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))


def get_include() -> str:
    """Directory that holds ``gptools/fft.stan`` etc. (same contract as the reference's get_include)."""
    return _HERE


def set_cmdstanpy_log_level(level: Optional[Union[str, int]] = None) -> None:
    """Set the level of every handler of cmdstanpy's logger (reference ``__init__.py:26-35``); ``None`` means
    ``WARNING``."""
    try:
        import cmdstanpy
    except ImportError as exc:  # pragma: no cover - cmdstanpy is absent from the build image
        raise ImportError("gptools_amd.stan.set_cmdstanpy_log_level needs cmdstanpy") from exc
    target = logging.WARNING if level is None else level
    for handler in cmdstanpy.utils.get_logger().handlers:
        handler.setLevel(target)


def get_user_header() -> str:
    return os.path.join(_ROOT, "include", "gptools_hip.hpp")


def get_library_dir() -> str:
    return os.path.dirname(_HERE)


def compile_model(
    model_name: Optional[str] = None,
    stan_file: Optional[str] = None,
    exe_file: Optional[str] = None,
    compile: Union[bool, str] = True,
    stanc_options: Optional[Dict[str, Any]] = None,
    cpp_options: Optional[Dict[str, Any]] = None,
    user_header: Optional[str] = None,
    cls: Optional[Type] = None,
    **kwargs,
):
    """Create a :class:`cmdstanpy.CmdStanModel` whose gptools Fourier functions run on the MI355X.

    Same arguments as the reference's ``compile_model``. Adds: this package's include path,
    ``allow-undefined``, the user header (unless the caller passes one) and the link flags of
    ``libgptools_hip.so``.
    """
    try:
        import cmdstanpy
    except ImportError as exc:  # pragma: no cover - cmdstanpy is absent from the build image
        raise ImportError("gptools_amd.stan.compile_model needs cmdstanpy and a cmdstan >= 2.36 install") from exc

    stanc_options = dict(stanc_options or {})
    stanc_options.setdefault("include-paths", []).append(get_include())
    stanc_options["allow-undefined"] = True
    cpp_options = dict(cpp_options or {})
    libdir = get_library_dir()
    inc = os.path.join(_ROOT, "include")
    cpp_options.setdefault("CXXFLAGS", f"-I{inc}")
    cpp_options.setdefault("LDFLAGS", f"-L{libdir} -Wl,-rpath,{libdir}")
    cpp_options.setdefault("LDLIBS", "-lgptools_hip")
    cls = cls or cmdstanpy.CmdStanModel
    return cls(
        model_name=model_name,
        stan_file=stan_file,
        exe_file=exe_file,
        compile=compile,
        stanc_options=stanc_options,
        cpp_options=cpp_options,
        user_header=user_header or get_user_header(),
        **kwargs,
    )


if os.environ.get("READTHEDOCS"):  # as the reference does at import time (python/stan/gptools/stan/__init__.py:93-94)
    try:
        set_cmdstanpy_log_level("ERROR")
    except ImportError:  # pragma: no cover
        pass
