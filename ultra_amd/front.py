"""The steps in front of the MAM stage for a whole batch, natively (csrc/front.hip, `ultra_front_*` in include/ultra_hip.h):
hit lines -> NAMs -> instances + bounds -> ONE NAM-chaining launch -> location selection -> ONE device-planned MAM stage.

What /root/reference/modules/align.py:427-489 does per read in Python comes back as flat arrays; the reference's tuples
(`mam` namedtuples, gap lists) are built only for the locations a caller asks for (`FrontResult.mam(l)`), not eagerly
for every read of the batch.
"""
import ctypes as C

import numpy as np

from . import classify_read_with_mams as _stage
from . import lib as _lib


def _i64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class PackedBatch(object):
    """The input of ultra_front_run: raw reads and the hit lines of every read as byte ranges of one text block."""
    __slots__ = ("n", "reads", "read_off", "text", "fw_begin", "fw_end", "rc_begin", "rc_end")

    def __init__(self, reads, read_off, text, fw_begin, fw_end, rc_begin, rc_end):
        a64 = lambda v: np.ascontiguousarray(v, np.int64)
        self.reads = np.ascontiguousarray(reads, np.uint8)
        self.read_off = a64(read_off)
        self.text = np.ascontiguousarray(text, np.uint8)
        self.fw_begin, self.fw_end, self.rc_begin, self.rc_end = a64(fw_begin), a64(fw_end), a64(rc_begin), a64(rc_end)
        self.n = len(self.read_off) - 1
        if not (len(self.fw_begin) == len(self.fw_end) == len(self.rc_begin) == len(self.rc_end) == self.n):
            raise ValueError("PackedBatch: one byte range per read and strand")

    @classmethod
    def from_records(cls, records):
        """records: [(acc, seq, qual, hits, hits_rc)] as the reference's queue batches hold them (hits = raw hit lines)."""
        seqs = [r[1] for r in records]
        fw = ["".join(r[3]) for r in records]
        rc = ["".join(r[4]) for r in records]
        n = len(records)
        read_off = np.zeros(n + 1, np.int64)
        np.cumsum([len(s) for s in seqs], out=read_off[1:])
        fw_len = np.fromiter((len(s) for s in fw), np.int64, n)
        rc_len = np.fromiter((len(s) for s in rc), np.int64, n)
        fw_begin = np.zeros(n, np.int64)
        if n:
            np.cumsum(fw_len[:-1], out=fw_begin[1:])
        fw_total = int(fw_len.sum())
        rc_begin = np.full(n, fw_total, np.int64)
        if n:
            rc_begin[1:] += np.cumsum(rc_len[:-1])
        # one byte per character whatever the input holds: offsets are character counts
        text = np.frombuffer(("".join(fw) + "".join(rc)).encode("latin-1", "replace") or b"\n", np.uint8)
        reads = np.frombuffer("".join(seqs).encode("latin-1", "replace") or b"\0", np.uint8)
        return cls(reads, read_off, text, fw_begin, fw_begin + fw_len, rc_begin, rc_begin + rc_len)


class FrontResult(object):
    """Arrays of one ultra_front_run (copies: they outlive the next call) + factories for the reference's Python objects."""

    def __init__(self, view, n_reads, with_mam):
        def arr(ptr, n, dt):
            return np.ctypeslib.as_array(ptr, shape=(max(n, 1),)).view(dt)[:n].copy()
        n_loc = int(view.n_loc)
        self.n_reads, self.n_loc = n_reads, n_loc
        self.loc_off = arr(view.loc_off, n_reads + 1, np.int64)
        self.last_limit = arr(view.last_limit, n_reads, np.int64)
        self.loc_chr = arr(view.loc_chr, n_loc, np.int32)
        self.loc_rc = arr(view.loc_rc, n_loc, np.uint8)
        self.loc_score = arr(view.loc_score, n_loc, np.int64)
        self.read_mod_off = arr(view.read_mod_off, n_reads + 1, np.int64)
        self.read_mod = arr(view.read_mod, int(self.read_mod_off[-1]), np.uint8)
        self.sol_off = arr(view.sol_off, n_loc + 1, np.int64)
        ns = int(self.sol_off[-1])
        self.sol_part = arr(view.sol_part, 3 * ns, np.int64).reshape(-1, 3)
        self.sol_x, self.sol_y = arr(view.sol_x, ns, np.int64), arr(view.sol_y, ns, np.int64)
        self.sol_c, self.sol_d, self.sol_j = arr(view.sol_c, ns, np.int32), arr(view.sol_d, ns, np.int32), arr(view.sol_j, ns, np.int32)
        self.loc_read_off = arr(view.loc_read_off, n_loc + 1, np.int64)
        self.loc_reads = arr(view.loc_reads, int(self.loc_read_off[-1]), np.uint8)
        self.n_quadratic, self.n_nlogn = int(view.n_quadratic), int(view.n_nlogn)
        self.ms = [view.ms[i] for i in range(6)]
        self.res = None
        if with_mam:
            m = view.mam
            tc = 0
            chain_off = arr(m.chain_off, n_loc + 1, np.int64)
            if n_loc:
                tc = int(chain_off[-1])
            self.res = {"value": arr(m.value, n_loc, np.float64), "n_mam": arr(m.n_mam, n_loc, np.int32), "chain_off": chain_off,
                        "chain_val": arr(m.chain_val, tc, np.float64), "chain_kind": arr(m.chain_kind, tc, np.uint8)}
            for f in ("x", "y", "c", "d", "msl", "j", "chr"):
                self.res["chain_" + f] = arr(getattr(m, "chain_" + f), tc, np.int32)
        self._lists = None

    # ---- the reference's objects, on demand ----
    def read_mod_seq(self, r):
        """The read after remove_read_polyA_ends(seq, qual, threshold, 1) (align.py:431)."""
        return self.read_mod[int(self.read_mod_off[r]):int(self.read_mod_off[r + 1])].tobytes().decode("latin-1")

    def loc_read_seq(self, l):
        """The read as location l's MAM stage saw it: reverse-complemented when the location is on the reverse strand."""
        return self.loc_reads[int(self.loc_read_off[l]):int(self.loc_read_off[l + 1])].tobytes().decode("latin-1")

    def solution(self, l):
        """The NAM chain of location l as Mem tuples (align.py:22)."""
        from .align import mem
        a, b = int(self.sol_off[l]), int(self.sol_off[l + 1])
        return [mem(int(self.sol_x[k]), int(self.sol_y[k]), int(self.sol_c[k]), int(self.sol_d[k]), int(self.sol_y[k] - self.sol_x[k] + 1), int(self.sol_j[k]),
                    "%d^%d^%d" % tuple(self.sol_part[k])) for k in range(a, b)]

    def mam(self, l):
        """(non_covered_regions, value, mam_solution) of location l: what classify_read_with_mams.main returns (:594)."""
        if self._lists is None:         # plain lists once: numpy scalar indexing costs more than building the tuples
            r = self.res
            self._lists = tuple(r[k].tolist() for k in ("n_mam", "chain_off", "value", "chain_x", "chain_y", "chain_c", "chain_d", "chain_val",
                                                        "chain_msl", "chain_j", "chain_kind", "chain_chr")) + (np.diff(self.loc_read_off).tolist(),)
        n_mam, chain_off, value, cx, cy, cc, cd, cval, cmsl, cj, ck, cchr, read_len = self._lists
        if n_mam[l] == 0:
            return [], -1, tuple()
        picked = []
        for r in range(chain_off[l], chain_off[l + 1]):
            msl = cmsl[r]
            val = int(cval[r]) if msl == 0 else cval[r]
            picked.append(_stage.mam(cx[r], cy[r], cc[r], cd[r], val, cj[r], msl, _stage.mam_name(ck[r], cchr[r], cx[r], cy[r]), cchr[r]))
        gaps = []
        if picked:
            gaps.append(picked[0].c)
            for p, q in zip(picked[:-1], picked[1:]):
                gaps.append(q.c - p.d - 1)
            gaps.append(read_len[l] - picked[-1].d)
        return gaps, value[l], tuple(picked)


class Front(object):
    """ultra_front: one per worker.  max_intron_chr: {chr_id: longest annotated intron} (align.py:404)."""

    def __init__(self, ctx, dindex=None, native_index=None, max_intron_chr=None, n_threads=0):
        self.lib = _lib.load()
        self.ctx, self.dindex, self.native = ctx, dindex, native_index
        self._h = C.c_void_p()
        rc = self.lib.ultra_front_create(C.byref(self._h))
        if rc != 0:
            raise _lib.UltraError(rc, "ultra_front_create failed")
        n_chr = (max(max_intron_chr) + 1) if max_intron_chr else 0
        self.max_intron = np.full(max(n_chr, 1), -1, np.int64)
        for c, v in (max_intron_chr or {}).items():
            self.max_intron[int(c)] = int(v)
        self.n_chr = n_chr
        self.n_threads = int(n_threads)

    def close(self):
        if self._h:
            self.lib.ultra_front_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _input(self, pb, max_global_intron, dropoff, max_loc, polyA_threshold, min_acc):
        inp = _lib.FrontInput()
        inp.n_reads = pb.n
        inp.reads, inp.read_off, inp.seed_text = _u8(pb.reads), _i64(pb.read_off), _u8(pb.text)
        inp.fw_begin, inp.fw_end, inp.rc_begin, inp.rc_end = _i64(pb.fw_begin), _i64(pb.fw_end), _i64(pb.rc_begin), _i64(pb.rc_end)
        inp.n_chr, inp.max_intron_chr = self.n_chr, _i64(self.max_intron)
        inp.max_global_intron, inp.dropoff, inp.max_loc = int(max_global_intron), float(dropoff), int(max_loc)
        inp.polyA_threshold, inp.min_acc, inp.n_threads = int(polyA_threshold), float(min_acc), self.n_threads
        return inp

    def _check(self, rc):
        if rc != 0:
            raise _lib.UltraError(rc, (self.lib.ultra_front_last_error(self._h) or b"").decode())

    def prepare(self, pb, max_global_intron, polyA_threshold=20):
        """Phase 1 only, no GPU: per instance (read, chr, is_rc, bound, intron limit, [(y, c, d)]) in processing order, and
        the trimmed reads."""
        inp = self._input(pb, max_global_intron, 1.0, 1, polyA_threshold, 0.0)
        v = _lib.FrontView()
        self._check(self.lib.ultra_front_prepare(self._h, C.byref(inp), C.byref(v)))
        ni, nn = int(v.n_inst), int(v.n_nam)
        arr = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(max(n, 1),)).view(dt)[:n].copy()
        info = arr(v.inst_chr_rc_bound, 4 * ni, np.int64).reshape(-1, 4)
        off = arr(v.inst_nam_off, ni + 1, np.int64)
        lim = arr(v.inst_intron, ni, np.int64)
        y, c, d = arr(v.nam_y, nn, np.int32), arr(v.nam_c, nn, np.int32), arr(v.nam_d, nn, np.int32)
        mod_off = arr(v.read_mod_off, pb.n + 1, np.int64)
        mod = arr(v.read_mod, int(mod_off[-1]), np.uint8)
        insts = [(int(info[i, 3]), int(info[i, 0]), bool(info[i, 1]), int(info[i, 2]), int(lim[i]),
                  list(zip(y[off[i]:off[i + 1]].tolist(), c[off[i]:off[i + 1]].tolist(), d[off[i]:off[i + 1]].tolist()))) for i in range(ni)]
        reads = [mod[int(mod_off[r]):int(mod_off[r + 1])].tobytes().decode("latin-1") for r in range(pb.n)]
        return insts, reads

    def run(self, pb, max_global_intron, dropoff, max_loc, polyA_threshold, min_acc, with_mam_stage=True, copy=True):
        inp = self._input(pb, max_global_intron, dropoff, max_loc, polyA_threshold, min_acc)
        v = _lib.FrontView()
        self._check(self.lib.ultra_front_run(self._h, self.ctx._h, self.dindex.handle if self.dindex is not None else None,
                                             self.native.handle if self.native is not None else None, C.byref(inp),
                                             1 if with_mam_stage else 0, C.byref(v)))
        if not copy:            # timing runs: the arrays stay in the library
            return {"n_loc": int(v.n_loc), "ms": [v.ms[i] for i in range(6)]}
        return FrontResult(v, pb.n, with_mam_stage)
