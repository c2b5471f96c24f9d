"""CPU-only: the staging ring of the *_host entry points splits a copy over its worker threads without losing bytes
(round 2 lost the last n % nt bytes for nt >= 9 when floor(n / nt) was a multiple of 4096: ADVICE r2, medium)."""
import ctypes

import pytest


@pytest.fixture(scope="module")
def lib():
    from gptools_amd import _lib

    return _lib.load()


def slices(lib, n, nt):
    out = []
    for i in range(nt):
        lo, ln = ctypes.c_size_t(0), ctypes.c_size_t(0)
        lib.gpt_host_copy_slice(n, nt, i, ctypes.byref(lo), ctypes.byref(ln))
        out.append((lo.value, ln.value))
    return out


def check_tiling(lib, n, nt):
    pos = 0
    for lo, ln in slices(lib, n, nt):
        if ln:
            assert lo == pos, (n, nt)
            assert lo % 4096 == 0
            pos += ln
    assert pos == n, (n, nt, pos)


def test_slices_tile_the_copy_exactly(lib):
    for nt in range(1, 33):
        for k in (1, 2, 3, 7, 256):
            for extra in (0, 8, 16, 4088, 4096, 4104):
                check_tiling(lib, nt * 4096 * k + extra, nt)
    # the advisor's case: a 1-D series with W = 524288 has cov = 2097160 bytes = 16 * 131072 + 8
    check_tiling(lib, 2097160, 16)
    for n in (0, 1, 8, 4095, 4097, (8 << 20), (8 << 20) - 8):
        for nt in (1, 6, 9, 16, 32):
            check_tiling(lib, n, nt)
