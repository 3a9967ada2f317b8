"""Drop-in for the chainers of /root/reference/modules/colinear_solver.py, backed by the GPU chaining
kernels of libultra_hip.so (one wavefront per instance).

NAM (seed) chaining, called per (read, chromosome, strand) at align.py:448-453:

    read_coverage(mems, max_intron)                         reference :169-272
    n_logn_read_coverage(mems)                              reference :85-167

take seed tuples (fields y, c, d are read; j == position) and return (solutions, value): EVERY co-optimal
chain as a list of mems, in ascending order of its last mem, and the covered-bases optimum.
read_coverage_batch does many instances in one launch.

MAM chaining:

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


# ---- NAM chaining -------------------------------------------------------

def all_solutions(mems, C, trace):
    """all_solutions_c_max_indicies + reconstruct_all_solutions (reference :30-31, :49-62) on the DP vectors
    the library returns: every index attaining max(C), each walked back through trace."""
    best = int(C.max())
    solutions = []
    for idx in np.flatnonzero(C == best):
        idx = int(idx)
        sol = []
        while idx > 0:
            sol.append(mems[idx - 1])
            idx = int(trace[idx])
        solutions.append(sol[::-1])
    return solutions, best


def read_coverage_batch(instances, max_introns=None, solver=0, ctx=None):
    """instances: list of mem lists; max_introns: one limit per instance (or one for all).  solver 0 applies the
    reference's switch (align.py:448: < 90 mems -> quadratic with the intron limit, else n log n without).
    Returns [(solutions, value)] in input order."""
    ctx = ctx or _lib.default_context()
    off = np.zeros(len(instances) + 1, np.int64)
    np.cumsum([len(m) for m in instances], out=off[1:])
    flat = [m for inst in instances for m in inst]
    C, trace, _ = ctx.chain_nams_batch(off, [m.y for m in flat], [m.c for m in flat], [m.d for m in flat],
                                       max_introns, solver=solver)
    out = []
    for i, mems in enumerate(instances):
        lo, hi = int(off[i]) + i, int(off[i + 1]) + i + 1
        out.append(all_solutions(mems, C[lo:hi], trace[lo:hi]))
    return out


def read_coverage(mems, max_intron):
    return read_coverage_batch([mems], max_intron, solver=1)[0]


def n_logn_read_coverage(mems):
    return read_coverage_batch([mems], None, solver=2)[0]
