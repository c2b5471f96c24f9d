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
    text = re.sub(r"//[^\n]*", "", text)
    bodies = set(re.findall(r"^\s*[\w\[\], ]+?\s+(\w+)\s*\([^;{]*\)\s*\{", text, re.M))
    assert not bodies & set(HOT_PATH_FUNCTIONS), bodies & set(HOT_PATH_FUNCTIONS)
    bodies -= {"if", "for", "while"}  # (control flow inside the plain-Stan helpers)
    assert bodies <= {"gp_rfft_log_abs_det_jacobian", "gp_rfft2_log_abs_det_jacobian"} | set(n for n, _ in UTIL_HELPERS)


# the reference's util.stan outside the FFT layer (util.stan:9-250, 344-350): name, argument types of every overload
UTIL_HELPERS = [
    ("assert_equal", "int, int"), ("is_close", "real, real, real, real"),
    ("assert_close", "real, real, real, real"), ("assert_close", "real, real"), ("is_finite", "complex"),
    ("assert_close", "vector, vector, real, real"), ("assert_close", "vector, vector"),
    ("assert_close", "vector, real, real, real"), ("assert_close", "vector, real"), ("is_finite", "vector"),
    ("assert_finite", "vector"), ("is_finite", "matrix"), ("print_matrix", "complex_matrix"),
    ("assert_close", "matrix, matrix, real, real"), ("assert_close", "matrix, matrix"),
    ("assert_close", "matrix, real, real, real"), ("assert_close", "matrix, real"),
    ("assert_close", "complex_vector, complex_vector, real, real"), ("assert_close", "complex_vector, complex_vector"),
    ("assert_close", "complex_vector, complex, real, real"), ("assert_close", "complex_vector, complex"),
    ("gp_conditional_loc_scale", "vector, real, vector, matrix"),
]


def test_util_stan_keeps_the_non_hot_helpers():
    """`#include gptools/util.stan` alone must serve a model that calls the reference's debug helpers (VERDICT r2,
    missing 4): every overload of util.stan:9-250 and gp_conditional_loc_scale is defined, as ordinary Stan."""
    from gptools_amd import stan

    text = open(os.path.join(stan.get_include(), "gptools", "util.stan")).read()
    text = re.sub(r"//[^\n]*", "", text)
    defined = set()
    for fn, args in re.findall(r"^\s*[\w\[\], ]+?\s+(\w+)\s*\(([^;{)]*)\)\s*\{", text, re.M):
        defined.add((fn, ", ".join(a.split()[0] for a in args.split(","))))
    missing = [h for h in UTIL_HELPERS if h not in defined]
    assert not missing, missing
    for f, name in (("fft1.stan", "gp_rfft"), ("fft2.stan", "gp_rfft2")):  # the shims' full message (fft1.stan:84, fft2.stan:175)
        t = open(os.path.join(stan.get_include(), "gptools", f)).read()
        assert (f"`{name}_log_abs_det_jacobian` has been renamed to `{name}_log_abs_det_jac` to comply with upcoming "
                "changes (see https://github.com/stan-dev/stanc3/issues/1470 for details).") in t


def test_wrapper_twin_exports_the_reference_names():
    from gptools_amd import stan

    for name in ("get_include", "set_cmdstanpy_log_level", "compile_model"):  # python/stan/gptools/stan/__init__.py:8,26,38
        assert callable(getattr(stan, name))


def _stan_declarations():
    """name -> (return type, [(type, name), ...]) of every body-less declaration in the shipped .stan files."""
    from gptools_amd import stan

    inc = stan.get_include()
    decls = {}
    for name in ("fft1.stan", "fft2.stan", "util.stan", "fft_obs.stan", "fft_kernel.stan"):
        text = open(os.path.join(inc, "gptools", name)).read()
        text = re.sub(r"//[^\n]*", "", text)
        for ret, fn, args in re.findall(r"^\s*(\w+)\s+(\w+)\s*\(([^)]*)\)\s*;", text, re.M):
            decls[fn] = (ret, [tuple(a.split()) for a in args.split(",")])
    return decls


def _header_definition(hpp, fn):
    """(template head, return type text, parameter list text) of the definition of `fn` in the user header."""
    m = re.search(r"template <((?:[^<>]|<(?:[^<>]|<(?:[^<>]|<[^<>]*>)*>)*>)*)>\s*\n?([^;{}]*?)\b" + fn + r"\(([^{]*?)\)\s*\{", hpp)
    assert m, f"{fn} is not defined in the user header"
    return m.group(1), m.group(2), m.group(3)


# what stanc emits per Stan argument type: constraint macro of the header / how the parameter is spelled
REQ = {"vector": "GPT_REQ_VEC", "complex_vector": "GPT_REQ_CVEC", "matrix": "GPT_REQ_MAT", "complex_matrix": "GPT_REQ_CMAT",
       "real": "stan::is_stan_scalar"}


def test_user_header_defines_every_declared_function():
    """Every declaration of the shipped .stan files has a definition in the user header whose template head follows
    from the declaration the way stanc derives its forward declaration: one type parameter and one constraint per
    non-int argument, in order; `const int&` for ints; `bool propto__` first on `_lpdf`; trailing pstream__."""
    from gptools_amd import stan

    hpp = open(stan.get_user_header()).read()
    decls = _stan_declarations()
    for fn in HOT_PATH_FUNCTIONS + ["gp_inv_rfft2_obs_lpdf", "gp_inv_rfft_obs_lpdf", "gp_inv_rfft2_obs_matern_lpdf",
                                    "gp_inv_rfft2_obs_exp_quad_lpdf", "gp_rfft2_matern_lpdf", "gp_rfft2_exp_quad_lpdf",
                                    "gp_rfft_matern_lpdf", "gp_rfft_exp_quad_lpdf"]:
        assert fn in decls, f"{fn} is not declared in the .stan files"
        ret, args = decls[fn]
        head, rtype, params = _header_definition(hpp, fn)
        tparams = re.findall(r"typename (T\d+__)", head)
        want_t = [f"T{i}__" for i, (t, _) in enumerate(args) if t != "int"]
        assert tparams == want_t, (fn, tparams, want_t)
        assert head.lstrip().startswith("bool propto__") == fn.endswith("_lpdf"), fn
        constraints = re.findall(r"(GPT_REQ_\w+|stan::is_stan_scalar)[(<](T\d+__)", head)
        assert constraints == [(REQ[t], f"T{i}__") for i, (t, _) in enumerate(args) if t != "int"], (fn, constraints)
        plist = [p.strip() for p in params.split(",")]
        assert plist[-1] == "std::ostream* pstream__", fn
        for i, ((t, name), p) in enumerate(zip(args, plist)):
            assert p == (f"const int& {name}" if t == "int" else f"const T{i}__& {name}"), (fn, p)
        # return type: complex containers spell std::complex<promote_args_t<...>>, reals a bare promote_args_t
        assert ("std::complex" in rtype) == ret.startswith("complex"), (fn, rtype)
        assert ("Eigen::Matrix" in rtype) == (ret != "real"), (fn, rtype)
        assert "stan::promote_args_t" in rtype and "auto" not in rtype.split(), (fn, rtype)
    # the Stan-typed veneer holds no C-ABI call of its own: everything goes through the compiled, tested bind header
    bind = open(os.path.join(os.path.dirname(stan.get_user_header()), "gptools_hip_bind.hpp")).read()
    assert "#include <gptools_hip_bind.hpp>" in hpp and not re.search(r"\bgpt_\w+_host\(", hpp)
    assert "gpt_gp_rfft2_lpdf_host" in bind and "#include <stan" not in bind and "Eigen::" not in bind
    assert "gp_oracle" not in hpp + bind and "std::function" not in hpp.replace("no std::function", "")


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
    for fn in ("gp_inv_rfft2_obs_lpdf", "gp_inv_rfft_obs_lpdf", "gp_inv_rfft2_obs_matern_lpdf", "gp_inv_rfft2_obs_exp_quad_lpdf"):
        assert re.search(rf"^real {fn}\(", text, re.M), fn
        assert re.search(rf"\b{fn}\(", hpp), fn
    bind = open(os.path.join(os.path.dirname(stan.get_user_header()), "gptools_hip_bind.hpp")).read()
    assert "gpt_gp_inv_rfft2_obs_lpdf_host" in bind and "gpt_gp_inv_rfft2_obs_lpdf_kernel_host" in bind
    kt = open(os.path.join(inc, "gptools", "fft_kernel.stan")).read()
    for fn in ("gp_rfft2_matern_lpdf", "gp_rfft2_exp_quad_lpdf", "gp_rfft_matern_lpdf", "gp_rfft_exp_quad_lpdf"):
        assert re.search(rf"^real {fn}\(", kt, re.M), fn
        assert re.search(rf"\b{fn}\(", hpp), fn
    assert "gpt_gp_rfft2_lpdf_kernel_host" in bind
    assert "fft_kernel" not in open(os.path.join(inc, "gptools", "fft.stan")).read()
    # the reference's own include names stay free of the extension: existing models compile unchanged
    assert "fft_obs" not in open(os.path.join(inc, "gptools", "fft.stan")).read()
