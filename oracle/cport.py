"""ctypes front end of the C oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY. Nothing under ultra_amd/ may import this module; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg do.
"""
import ctypes as C
import ctypes as C_mod
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OPS = "=IDX"  # op code -> extended cigar letter (0 match, 1 insert, 2 delete, 3 mismatch)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.oracle_rmq_run.restype = C.c_int
        for name in ("oracle_hw_bruteforce", "oracle_hw_bitvector", "oracle_hw_forward_only", "oracle_hw_batch",
                     "oracle_chain_quadratic", "oracle_chain_nlogn", "oracle_chain_batch"):
            getattr(_LIB, name).restype = C.c_int
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def ops_to_cigar(ops):
    out, i = [], 0
    while i < len(ops):
        j = i
        while j < len(ops) and ops[j] == ops[i]:
            j += 1
        out.append("%d%s" % (j - i, OPS[ops[i]]))
        i = j
    return "".join(out)


def ins_range(query, target, expect):
    """(I_min, I_max): fewest / most inserted query bases over every optimal global alignment of query with target,
    whose distance must be `expect` (edlib_port.c: oracle_ins_range)."""
    q = query.encode() if isinstance(query, str) else bytes(query)
    t = target.encode() if isinstance(target, str) else bytes(target)
    a, b = C.c_int(), C.c_int()
    rc = lib().oracle_ins_range(q, len(q), t, len(t), int(expect), C.byref(a), C.byref(b))
    if rc != 0:
        raise RuntimeError("oracle_ins_range failed rc=%d" % rc)
    return a.value, b.value


def set_length_prefilter(on):
    """bench.py's cpu_baseline leg only: answer pairs whose query is more than k longer than the target without computing
    them (edlib_port.c: oracle_length_prefilter).  Tests never set it."""
    lib().oracle_set_length_prefilter(1 if on else 0)


def hw_align(query, target, k, method="bitvector", want_path=True):
    """Oracle for edlib.align(query, target, mode='HW', task='path', k=k).

    query/target: bytes or str. Returns dict(editDistance, locations, cigar, insertions)."""
    q = query.encode() if isinstance(query, str) else bytes(query)
    t = target.encode() if isinstance(target, str) else bytes(target)
    m, n = len(q), len(t)
    k = int(k)
    cap = max(n, 1)
    starts = np.zeros(cap, np.int32)
    ends = np.zeros(cap, np.int32)
    ops = np.zeros(m + n + 2, np.uint8)
    ed, nloc, ins, nops = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    L = lib()
    if method == "bruteforce":
        rc = L.oracle_hw_bruteforce(q, m, t, n, k, C.byref(ed), C.byref(nloc), _p(starts, C.c_int), _p(ends, C.c_int),
                                    cap, C.byref(ins), _p(ops, C.c_uint8), len(ops), C.byref(nops))
    else:
        rc = L.oracle_hw_bitvector(q, m, t, n, k, C.byref(ed), C.byref(nloc), _p(starts, C.c_int), _p(ends, C.c_int),
                                   cap, C.byref(ins), _p(ops, C.c_uint8), len(ops), C.byref(nops), 1 if want_path else 0)
    if rc != 0:
        raise RuntimeError("oracle hw_align failed rc=%d" % rc)
    if ed.value < 0:
        return {"editDistance": -1, "locations": [], "cigar": None, "insertions": 0, "alphabetLength": 0}
    locs = [(int(starts[i]), int(ends[i])) for i in range(nloc.value)]
    return {"editDistance": ed.value, "locations": locs, "cigar": ops_to_cigar(ops[:nops.value].tolist()),
            "insertions": ins.value, "alphabetLength": len(set(q + t))}


def hw_batch(arena, q_off, q_len, t_off, t_len, k, loc_off, want_path=True):
    """Single-thread batched oracle. arena: uint8 array. Returns (ed, n_loc, starts, ends, first_ins, overflowed)."""
    n = len(q_len)
    arena = np.ascontiguousarray(arena, np.uint8)
    q_off = np.ascontiguousarray(q_off, np.int64); t_off = np.ascontiguousarray(t_off, np.int64)
    q_len = np.ascontiguousarray(q_len, np.int32); t_len = np.ascontiguousarray(t_len, np.int32)
    k = np.ascontiguousarray(k, np.int32); loc_off = np.ascontiguousarray(loc_off, np.int64)
    tot = int(loc_off[-1])
    ed = np.zeros(n, np.int32); nl = np.zeros(n, np.int32); fi = np.zeros(n, np.int32)
    st = np.zeros(max(tot, 1), np.int32); en = np.zeros(max(tot, 1), np.int32)
    rc = lib().oracle_hw_batch(_p(arena, C.c_uint8), n, _p(q_off, C.c_int64), _p(q_len, C.c_int32),
                               _p(t_off, C.c_int64), _p(t_len, C.c_int32), _p(k, C.c_int32), _p(loc_off, C.c_int64),
                               _p(ed, C.c_int32), _p(nl, C.c_int32), _p(st, C.c_int32), _p(en, C.c_int32),
                               _p(fi, C.c_int32), 1 if want_path else 0)
    if rc < 0:
        raise RuntimeError("oracle hw_batch failed rc=%d" % rc)
    return ed, nl, st, en, fi, rc


def chain_quadratic(c, d, val, thr=20):
    n = len(c)
    c = np.ascontiguousarray(c, np.int32); d = np.ascontiguousarray(d, np.int32)
    val = np.ascontiguousarray(val, np.float64)
    out = np.zeros(max(n, 1), np.int32); v = C.c_double()
    ln = lib().oracle_chain_quadratic(n, _p(c, C.c_int32), _p(d, C.c_int32), _p(val, C.c_double), thr,
                                      _p(out, C.c_int32), C.byref(v))
    return out[:ln].tolist(), v.value


def chain_nlogn(x, y, c, d, val, thr=5):
    n = len(c)
    x = np.ascontiguousarray(x, np.int32); y = np.ascontiguousarray(y, np.int32)
    c = np.ascontiguousarray(c, np.int32); d = np.ascontiguousarray(d, np.int32)
    val = np.ascontiguousarray(val, np.float64)
    out = np.zeros(max(n, 1), np.int32); v = C.c_double()
    ln = lib().oracle_chain_nlogn(n, _p(x, C.c_int32), _p(y, C.c_int32), _p(c, C.c_int32), _p(d, C.c_int32),
                                  _p(val, C.c_double), thr, _p(out, C.c_int32), C.byref(v))
    if ln < 0:
        raise RuntimeError("oracle chain_nlogn: reference BUG path")
    return out[:ln].tolist(), v.value


def chain_batch(inst_off, x, y, c, d, val, switch_n=200, thr_small=20, thr_large=5):
    inst_off = np.ascontiguousarray(inst_off, np.int64)
    n_inst = len(inst_off) - 1
    x = np.ascontiguousarray(x, np.int32); y = np.ascontiguousarray(y, np.int32)
    c = np.ascontiguousarray(c, np.int32); d = np.ascontiguousarray(d, np.int32)
    val = np.ascontiguousarray(val, np.float64)
    idx = np.zeros(max(len(c), 1), np.int32); ln = np.zeros(max(n_inst, 1), np.int32)
    value = np.zeros(max(n_inst, 1), np.float64)
    rc = lib().oracle_chain_batch(n_inst, _p(inst_off, C.c_int64), _p(x, C.c_int32), _p(y, C.c_int32),
                                  _p(c, C.c_int32), _p(d, C.c_int32), _p(val, C.c_double),
                                  switch_n, thr_small, thr_large, _p(idx, C.c_int32), _p(ln, C.c_int32),
                                  _p(value, C.c_double))
    if rc != 0:
        raise RuntimeError("oracle chain_batch failed")
    return idx, ln[:n_inst], value[:n_inst]


def rmq_run(d, upd_j, upd_v, ql, qr):
    d = np.ascontiguousarray(d, np.int32)
    upd_j = np.ascontiguousarray(upd_j, np.int32); upd_v = np.ascontiguousarray(upd_v, np.float64)
    ql = np.ascontiguousarray(ql, np.int32); qr = np.ascontiguousarray(qr, np.int32)
    nq = len(ql)
    oc = np.zeros(max(nq, 1), np.float64); oj = np.zeros(max(nq, 1), np.int32); ok = np.zeros(max(nq, 1), np.int32)
    rc = lib().oracle_rmq_run(len(d), _p(d, C.c_int32), len(upd_j), _p(upd_j, C.c_int32), _p(upd_v, C.c_double),
                              nq, _p(ql, C.c_int32), _p(qr, C.c_int32), _p(oc, C.c_double), _p(oj, C.c_int32),
                              _p(ok, C.c_int32))
    if rc != 0:
        raise RuntimeError("oracle rmq_run failed")
    return oc[:nq], oj[:nq], ok[:nq]


# ---- NAM chaining -----------------------------------------------------
def all_optimal_chains(C, trace):
    """all_solutions_c_max_indicies + reconstruct_all_solutions (colinear_solver.py:30, :49-62) on the DP
    vectors: every index attaining the maximum, each traced back; returns (list of index lists, value)."""
    best = max(C)
    sols = []
    for idx in [i for i, v in enumerate(C) if v == best]:
        sol = []
        while idx > 0:
            sol.append(idx - 1)
            idx = trace[idx]
        sols.append(sol[::-1])
    return sols, best


def nam_quadratic(y, c, d, max_intron):
    n = len(c)
    y = np.ascontiguousarray(y, np.int32); c = np.ascontiguousarray(c, np.int32); d = np.ascontiguousarray(d, np.int32)
    C = np.zeros(n + 1, np.int64); tr = np.zeros(n + 1, np.int32)
    lib().oracle_nam_quadratic(n, _p(y, C_mod.c_int32), _p(c, C_mod.c_int32), _p(d, C_mod.c_int32), C_mod.c_int64(int(max_intron)),
                               _p(C, C_mod.c_int64), _p(tr, C_mod.c_int32))
    return C.tolist(), tr.tolist()


def nam_nlogn(c, d):
    n = len(c)
    c = np.ascontiguousarray(c, np.int32); d = np.ascontiguousarray(d, np.int32)
    C = np.zeros(n + 1, np.int64); tr = np.zeros(n + 1, np.int32)
    lib().oracle_nam_nlogn.restype = C_mod.c_int
    rc = lib().oracle_nam_nlogn(n, _p(c, C_mod.c_int32), _p(d, C_mod.c_int32), _p(C, C_mod.c_int64), _p(tr, C_mod.c_int32))
    if rc != 0:
        raise RuntimeError("oracle nam_nlogn: reference BUG path")
    return C.tolist(), tr.tolist()


# ---- base-level affine-gap alignment -------------------------------------
def sg_affine(query, target, match=2, mismatch=-2, gap_open=3, gap_extend=1, want_path=True):
    """Oracle for help_functions.parasail_alignment (sg_trace_scan): dict(score, end_query, end_ref, cigar, ops)."""
    q = query.encode() if isinstance(query, str) else bytes(query)
    t = target.encode() if isinstance(target, str) else bytes(target)
    m, n = len(q), len(t)
    ops = np.zeros(m + n + 2, np.uint8)
    score, eq, et, nops = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    L = lib()
    L.oracle_sg_affine.restype = C.c_int
    rc = L.oracle_sg_affine(q, m, t, n, match, mismatch, gap_open, gap_extend, C.byref(score), C.byref(eq), C.byref(et),
                            _p(ops, C.c_uint8) if want_path else None, len(ops), C.byref(nops))
    if rc != 0:
        raise RuntimeError("oracle sg_affine failed rc=%d" % rc)
    o = ops[:nops.value].tolist()
    return {"score": score.value, "end_query": eq.value, "end_ref": et.value, "ops": o, "cigar": ops_to_cigar(o) if want_path else None}
