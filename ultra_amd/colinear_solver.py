"""Drop-in for the two MAM chainers of /root/reference/modules/colinear_solver.py, backed by the
GPU chaining kernels of libultra_hip.so (one wavefront per instance).

    read_coverage_mam_score(mams, overlap_threshold=20)     reference :276-389
    n_logn_read_coverage_mams(mams, overlap_threshold=5)    reference :394-493

Both take MAM tuples already sorted by y (fields x, y, c, d, val are read) and return
(tuple_of_chosen_mams, value, unique). The reference's third element (an O(n^2) "are all DP values
distinct" flag, :382-386 / :488-491) is ignored by its only caller
(classify_read_with_mams.py:563,575) and is returned as None here.
"""
import numpy as np

from . import lib as _lib


def _solve(mams, solver, thr_small, thr_large, ctx=None):
    ctx = ctx or _lib.default_context()
    n = len(mams)
    idx, ln, value = ctx.chain_mams_batch(np.array([0, n], np.int64), [m.x for m in mams], [m.y for m in mams],
                                          [m.c for m in mams], [m.d for m in mams], [float(m.val) for m in mams],
                                          solver=solver, overlap_small=thr_small, overlap_large=thr_large)
    return tuple(mams[j] for j in idx[:int(ln[0])]), float(value[0]), None


def read_coverage_mam_score(mams, overlap_threshold=20):
    return _solve(mams, 1, overlap_threshold, -1)


def n_logn_read_coverage_mams(mams, overlap_threshold=5):
    return _solve(mams, 2, -1, overlap_threshold)
