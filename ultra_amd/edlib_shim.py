"""edlib-shaped front end over libultra_hip.so.

`align(query, target, mode="HW", task="path", k=-1)` mirrors the third-party call the reference
makes at /root/reference/modules/classify_read_with_mams.py:99 and returns the same dictionary
keys the reference consumes (:100-119): 'editDistance', 'locations', 'cigar' (+ 'alphabetLength').
Only the semi-global mode the hot path uses ("HW") is implemented; anything else raises.
One call = one GPU batch of one pair: use ultra_amd.lib.Context.myers_hw_batch for real batches.
"""
import numpy as np

from . import lib as _lib

_OPS = "=IDX"


def _as_bytes(s):
    return s.encode() if isinstance(s, str) else bytes(s)


def align(query, target, mode="NW", task="distance", k=-1, additionalEqualities=None, ctx=None):
    if mode != "HW":
        raise NotImplementedError("ultra_amd.edlib_shim.align: only mode='HW' is implemented (got %r)" % (mode,))
    if additionalEqualities:
        raise NotImplementedError("ultra_amd.edlib_shim.align: additionalEqualities is not supported")
    if task not in ("distance", "locations", "path"):
        raise ValueError("task must be 'distance', 'locations' or 'path'")
    q, t = _as_bytes(query), _as_bytes(target)
    alphabet = len(set(q + t))
    if len(q) == 0 or len(t) == 0:
        # edlib's empty-sequence special cases are unreachable from the reference's call sites
        raise ValueError("ultra_amd.edlib_shim.align: empty query or target")
    ctx = ctx or _lib.default_context()
    arena = np.frombuffer(q + t, dtype=np.uint8)
    r = ctx.myers_hw_batch(arena, [0], [len(q)], [len(q)], [len(t)], [int(k)], want_ops=(task == "path"))
    ed = int(r["ed"][0])
    if ed < 0:
        return {"editDistance": -1, "alphabetLength": alphabet, "locations": [], "cigar": None}
    locs = [(int(s), int(e)) for s, e in zip(r["loc_start"], r["loc_end"])]
    out = {"editDistance": ed, "alphabetLength": alphabet, "locations": locs, "cigar": None}
    if task == "distance":
        out["locations"] = [(None, e) for _, e in locs]
    if task == "path":
        n = int(r["n_ops"][0])
        ops = r["ops"][:n][::-1]  # the device writes the script from the last column backwards
        parts, i = [], 0
        while i < n:
            j = i
            while j < n and ops[j] == ops[i]:
                j += 1
            parts.append("%d%s" % (j - i, _OPS[ops[i]]))
            i = j
        out["cigar"] = "".join(parts)
    return out
