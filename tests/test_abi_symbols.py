"""CPU-only: the C-ABI library loads and exports every symbol include/gptools_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gptools_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from gptools_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build the library first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    # the ctypes binding covers the same set
    assert sorted(_lib.EXPORTED_SYMBOLS) == names
    _lib.load()


def test_path_selection_needs_no_gpu():
    from gptools_amd import _lib

    lib = _lib.load()
    assert lib.gpt_version().decode().startswith("gptools_hip")
    assert lib.gpt_path_for_shape(35, 60) == 0  # docs/trees lattice: one workgroup per lattice
    assert lib.gpt_path_for_shape(1, 128) == 0
    assert lib.gpt_path_for_shape(100, 100) == 0  # fits one workgroup's LDS without the stride padding only
    assert lib.gpt_path_for_shape(102, 100) == 2
    assert lib.gpt_path_for_shape(0, 5) == -1
    assert lib.gpt_path_for_shape(65536, 65536) == -1  # n exceeds the reference's 32-bit int
    # any-length path: long series, long or prime axes (four-step / Bluestein line transforms)
    assert lib.gpt_path_for_shape(1, 8192) == 3 and lib.gpt_path_for_shape(1, 10007) == 3
    assert lib.gpt_path_for_shape(5000, 6) == 3 and lib.gpt_path_for_shape(16384, 16384) == 3
    assert lib.gpt_path_for_shape(10, 4093) == 3  # fits the generic path's lines, but 4093 is prime: Bluestein
    assert lib.gpt_path_for_shape(110, 110) == 2 and lib.gpt_path_for_shape(4096, 4096) == 1
    assert lib.gpt_path_for_shape(1, 1 << 24) == 3  # 4096 x 4096: the longest four-step line
    assert lib.gpt_path_for_shape(1, (1 << 23) + 1) == -1  # would need a Bluestein transform of 2^25 points


def test_product_path_does_not_import_the_oracle():
    """The shipped package must never route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "gptools_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "gp_oracle" not in src and "import oracle" not in src, f
