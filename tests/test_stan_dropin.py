"""CPU-only checks of the Stan drop-in boundary: declarations-only include files, wrapper twin,
and the C++ user header naming every Stan function of the hot path."""
import os
import re

from conftest import ROOT

HOT_PATH_FUNCTIONS = [
    # util.stan:255-330
    "rfft", "expand_rfft", "inv_rfft", "rfft2", "inv_rfft2",
    # fft1.stan
    "gp_evaluate_rfft_scale", "gp_unpack_rfft", "gp_rfft", "gp_rfft_log_abs_det_jac", "gp_rfft_lpdf",
    "gp_pack_rfft", "gp_inv_rfft", "gp_periodic_exp_quad_cov_rfft", "gp_periodic_matern_cov_rfft",
    # fft2.stan
    "gp_evaluate_rfft2_scale", "gp_unpack_rfft2", "gp_pack_rfft2", "gp_rfft2", "gp_inv_rfft2",
    "gp_rfft2_log_abs_det_jac", "gp_rfft2_lpdf", "gp_periodic_exp_quad_cov_rfft2",
    "gp_periodic_matern_cov_rfft2",
]


def test_include_dir_resolves_the_reference_include_names():
    from gptools_amd import stan

    inc = stan.get_include()
    for name in ("fft.stan", "fft1.stan", "fft2.stan", "util.stan"):
        assert os.path.exists(os.path.join(inc, "gptools", name)), name
    text = "".join(open(os.path.join(inc, "gptools", n)).read() for n in ("fft1.stan", "fft2.stan", "util.stan"))
    for fn in HOT_PATH_FUNCTIONS:
        assert re.search(rf"\b{fn}\s*\(", text), f"{fn} is not declared"
    # declarations only: the hot-path functions have no bodies (the renamed shims that always reject do)
    bodies = re.findall(r"\b(\w+)\s*\([^;{]*\)\s*\{", text)
    assert set(bodies) <= {"gp_rfft_log_abs_det_jacobian", "gp_rfft2_log_abs_det_jacobian"}, bodies


def test_user_header_defines_every_declared_function():
    from gptools_amd import stan

    hpp = open(stan.get_user_header()).read()
    for fn in HOT_PATH_FUNCTIONS:
        assert re.search(rf"\b{fn}\s*\(", hpp) or re.search(rf"GPTOOLS_HIP_SCALE\({fn}\b", hpp), fn
    # it binds the C ABI, not anything CPU-side
    assert "gpt_gp_rfft2_lpdf_host" in hpp and "gp_oracle" not in hpp


def test_compile_model_needs_cmdstanpy_but_import_does_not():
    import pytest

    from gptools_amd import stan

    try:
        import cmdstanpy  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="cmdstanpy"):
            stan.compile_model(stan_file="does_not_matter.stan")


def test_observation_model_extension_is_declared_and_bound():
    """gptools/fft_obs.stan (an opt-in extension, not a reference file) and its definitions in the user header."""
    from gptools_amd import stan

    inc = stan.get_include()
    text = open(os.path.join(inc, "gptools", "fft_obs.stan")).read()
    hpp = open(stan.get_user_header()).read()
    for fn in ("gp_inv_rfft2_obs_lpdf", "gp_inv_rfft_obs_lpdf"):
        assert re.search(rf"^real {fn}\(", text, re.M), fn
        assert re.search(rf"inline auto {fn}\(", hpp), fn
    assert "gpt_gp_inv_rfft2_obs_lpdf_host" in hpp
    # the reference's own include names stay free of the extension: existing models compile unchanged
    assert "fft_obs" not in open(os.path.join(inc, "gptools", "fft.stan")).read()
