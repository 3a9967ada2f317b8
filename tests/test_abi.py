"""CPU: the C-ABI library loads and exports every symbol include/ultra_hip.h declares (no compute
calls: there is no GPU here), the product package never touches oracle/, and missing pieces fail loudly."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "ultra_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ultra_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ultra_amd import lib
    names = _declared_functions()
    assert len(names) >= 12
    assert os.path.exists(lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    so = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(so, n), n
    assert sorted(lib.EXPORTS) == names


def test_cfg_defaults_are_the_reference_constants():
    from ultra_amd import lib
    so = lib.load()
    c = lib.Cfg()
    so.ultra_cfg_default(ctypes.byref(c))
    # classify_read_with_mams.py:561 / :575 / :563
    assert (c.chain_switch_n, c.overlap_small, c.overlap_large) == (200, 20, 5)


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    from ultra_amd import lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lib.UltraError):
        lib.Context(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "ultra_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text, os.path.join(dirpath, f)
