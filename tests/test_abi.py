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


def test_python_set_order_restatement_matches_the_interpreter():
    """csrc/pyset.h restates CPython's tuple hash and set probing (the reference's sorted(set(hits), key=start) leaves
    equal starts in set order, classify_read_with_mams.py:180-253): checked against this interpreter's own sets,
    including every table size up to the device planner's capacity and repeated insertions."""
    import ctypes as C
    import random
    import numpy as np
    from ultra_amd import lib
    L = lib.load()
    rnd = random.Random(11)
    sizes = [1, 2, 4, 5, 6, 18, 19, 20, 76, 77, 78, 306, 307, 308, 1228, 1229, 1230, 2048] + [rnd.randint(1, 90) for _ in range(400)]
    for n in sizes:
        base = rnd.randint(0, 10 ** rnd.randint(2, 9))
        seen, lst = set(), []
        while len(lst) < n:
            a = base + rnd.randint(0, 4000)
            t = (rnd.randint(1, 40), a, a + rnd.randint(1, 300))
            if t not in seen:
                seen.add(t); lst.append(t)
        with_repeats = []
        for t in lst:                       # a key that is already in the set changes nothing when added again
            with_repeats.append(t)
            if rnd.random() < 0.3:
                with_repeats.append(rnd.choice(with_repeats))
        arr = np.ascontiguousarray(np.array(lst, np.int64).ravel())
        order = np.zeros(n, np.int32)
        assert L.ultra_pyset_order(arr.ctypes.data_as(C.POINTER(C.c_int64)), n, order.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        assert [lst[i] for i in order] == list(set(with_repeats)), n
