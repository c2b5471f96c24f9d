"""ctypes binding of libgptools_hip.so (the C ABI of include/gptools_hip.h).

The library is the product: there is no CPU fallback. Importing this module without the built library,
or calling it without a GPU, raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPT_LIB: another build of the same library (tuning variants made by scripts/mkvar.sh)
LIB_PATH = os.environ.get("GPT_LIB") or os.path.join(_HERE, "libgptools_hip.so")

GPT_OK, GPT_ERR_NOT_PD, GPT_ERR_SHAPE, GPT_ERR_UNSUPPORTED, GPT_ERR_TIMEOUT = 0, 1, 2, 3, 4
GPT_ROW_MAJOR, GPT_COL_MAJOR = 0, 1

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_dbl = ctypes.c_double
c_ptr = ctypes.c_void_p

# name -> argument types after the leading gpt_ctx* (all return int)
_SIGNATURES = {
    "gpt_rfft2": [c_ptr, c_i64, c_i64, c_i64, c_ptr],
    "gpt_inv_rfft2": [c_ptr, c_i64, c_i64, c_i64, c_ptr],
    "gpt_expand_rfft": [c_ptr, c_i64, c_i64, c_ptr],
    "gpt_gp_evaluate_rfft2_scale": [c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr],
    "gpt_gp_unpack_rfft2": [c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr],
    "gpt_gp_pack_rfft2": [c_ptr, c_i64, c_i64, c_i64, c_int, c_ptr],
    "gpt_gp_rfft2": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr],
    "gpt_gp_inv_rfft2": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr],
    "gpt_gp_rfft2_log_abs_det_jac": [c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr],
    "gpt_gp_rfft2_log_abs_det_jac_vjp": [c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr],
    "gpt_gp_rfft2_lpdf": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr],
    "gpt_gp_rfft2_lpdf_vjp": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr,
                              c_ptr, c_ptr, c_ptr],
    "gpt_gp_rfft2_lpdf_vjp_noalloc": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr,
                                      c_ptr, c_ptr, c_ptr],
    "gpt_gp_inv_rfft2_vjp": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr],
    "gpt_gp_rfft2_vjp": [c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_int,
                         c_ptr, c_ptr],
    "gpt_gp_periodic_exp_quad_cov_rfft2": [c_i64, c_i64, c_ptr, c_i64, c_int, c_ptr],
    "gpt_gp_periodic_matern_cov_rfft2": [c_dbl, c_i64, c_i64, c_ptr, c_i64, c_int, c_ptr],
    "gpt_gp_periodic_exp_quad_cov_rfft2_vjp": [c_i64, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr],
    "gpt_gp_periodic_matern_cov_rfft2_vjp": [c_dbl, c_i64, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr],
    "gpt_gp_rfft2_lpdf_kernel_vjp": [c_int, c_dbl, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_dbl, c_i64, c_i64, c_i64, c_int,
                                     c_ptr, c_ptr, c_ptr, c_ptr],
    "gpt_gp_inv_rfft2_obs_lpdf_vjp": [c_int, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_ptr,
                                      c_int, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                      c_ptr],
    "gpt_gp_inv_rfft2_obs_lpdf_kernel_vjp": [c_int, c_int, c_dbl, c_ptr, c_i64, c_ptr, c_i64, c_ptr, c_dbl, c_ptr, c_i64,
                                             c_i64, c_i64, c_ptr, c_int, c_i64, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr,
                                             c_ptr, c_ptr, c_ptr],
    # host-pointer entry points
    "gpt_gp_rfft2_host": [c_int, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr],
    "gpt_gp_inv_rfft2_host": [c_int, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr],
    "gpt_gp_rfft2_log_abs_det_jac_host": [c_int, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr],
    "gpt_gp_rfft2_lpdf_host": [c_int, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr],
    "gpt_gp_inv_rfft2_obs_lpdf_host": [c_int, c_int, c_ptr, c_ptr, c_dbl, c_ptr, c_ptr, c_i64, c_i64, c_dbl, c_int, c_i64,
                                       c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "gpt_gp_inv_rfft2_obs_lpdf_kernel_host": [c_int, c_int, c_int, c_dbl, c_ptr, c_ptr, c_dbl, c_ptr, c_dbl, c_ptr, c_i64,
                                              c_i64, c_dbl, c_int, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "gpt_gp_rfft2_lpdf_kernel_host": [c_int, c_int, c_dbl, c_ptr, c_ptr, c_ptr, c_dbl, c_i64, c_i64, c_int, c_ptr, c_ptr, c_ptr],
    "gpt_gp_inv_rfft2_vjp_host": [c_int, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr],
    "gpt_gp_rfft2_vjp_host": [c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_int, c_ptr, c_ptr],
    "gpt_rfft2_host": [c_int, c_ptr, c_i64, c_i64, c_ptr],
    "gpt_inv_rfft2_host": [c_int, c_ptr, c_i64, c_i64, c_ptr],
    "gpt_expand_rfft_host": [c_ptr, c_i64, c_ptr],
    "gpt_gp_evaluate_rfft2_scale_host": [c_int, c_ptr, c_i64, c_i64, c_int, c_ptr],
    "gpt_gp_unpack_rfft2_host": [c_int, c_ptr, c_i64, c_i64, c_int, c_ptr],
    "gpt_gp_pack_rfft2_host": [c_int, c_ptr, c_i64, c_i64, c_int, c_ptr],
    "gpt_gp_periodic_exp_quad_cov_rfft2_host": [c_int, c_i64, c_i64, c_ptr, c_int, c_ptr],
    "gpt_gp_periodic_matern_cov_rfft2_host": [c_int, c_dbl, c_i64, c_i64, c_ptr, c_int, c_ptr],
    "gpt_gp_periodic_exp_quad_cov_rfft2_vjp_host": [c_int, c_i64, c_i64, c_ptr, c_ptr, c_int, c_ptr],
    "gpt_gp_periodic_matern_cov_rfft2_vjp_host": [c_int, c_dbl, c_i64, c_i64, c_ptr, c_ptr, c_int, c_ptr],
    "gpt_slab_rows_fwd": [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_i64],
    "gpt_slab_cols_lpdf_vjp": [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr],
    "gpt_slab_rows_inv": [c_ptr, c_i64, c_i64, c_i64, c_i64, c_dbl, c_ptr],
    "gpt_slab_rows_fwd_blocked": [c_ptr, c_ptr, c_i64, c_i64, c_i64, c_ptr, c_i64],
    "gpt_slab_cols_lpdf_vjp_owned": [c_ptr, c_i64, c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i64, c_i64, c_i64, c_ptr, c_ptr],
    "gpt_slab_rows_inv_blocked": [c_ptr, c_i64, c_i64, c_i64, c_i64, c_dbl, c_ptr],
}
# entry points without the leading context argument / with other return types
_PLAIN = {
    "gpt_create": ([c_int, c_ptr, ctypes.POINTER(c_ptr)], c_int),
    "gpt_destroy": ([c_ptr], c_int),
    "gpt_set_stream": ([c_ptr, c_ptr], c_int),
    "gpt_synchronize": ([c_ptr], c_int),
    "gpt_last_error": ([c_ptr], ctypes.c_char_p),
    "gpt_last_min_eig": ([c_ptr], c_dbl),
    "gpt_version": ([], ctypes.c_char_p),
    "gpt_profile_begin": ([c_ptr], c_int),
    "gpt_profile_end": ([c_ptr, ctypes.POINTER(c_dbl), ctypes.POINTER(c_i64)], c_int),
    "gpt_path_for_shape": ([c_i64, c_i64], c_int),
    "gpt_host_copy_slice": ([ctypes.c_size_t, c_int, c_int, ctypes.POINTER(ctypes.c_size_t),
                             ctypes.POINTER(ctypes.c_size_t)], None),
    # cross-chain batching
    "gpt_batcher_create": ([c_int, c_int, c_int, c_i64, c_i64, c_int, ctypes.POINTER(c_ptr)], c_int),
    "gpt_batcher_destroy": ([c_ptr], c_int),
    "gpt_batcher_lpdf_host": ([c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr], c_int),
    "gpt_batcher_leave": ([c_ptr, c_int], c_int),
    "gpt_batcher_set_timeout": ([c_ptr, c_dbl], c_int),
    "gpt_batcher_min_eig": ([c_ptr, c_int], c_dbl),
    "gpt_batcher_launches": ([c_ptr], ctypes.c_longlong),
    "gpt_batcher_last_error": ([c_ptr], ctypes.c_char_p),
}

EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + list(_PLAIN))

_lib = None


def load():
    """Load the shared library (once). Raises if it has not been built: the HIP path is the product."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C gptools_amd/csrc`. gptools_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = [c_ptr] + args
        fn.restype = c_int
    for name, (args, res) in _PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib
