"""Packed index and native candidate enumeration.

The reference's MAM stage reads five pickled dictionaries (modules/align.py:53-110) and walks them per read in Python
(modules/classify_read_with_mams.py:180-300, :441-523).  `NativeIndex` flattens those dictionaries once into the
arrays of `ultra_index_input` (include/ultra_hip.h); `NativeIndex.plan(instances)` then builds the whole batch's
`ultra_mam_input` inside libultra_hip.so (csrc/plan.hip) - the same candidates, pairs and sequence classes as
`classify_read_with_mams.prepare_batch`, without a Python object per candidate.
"""
import ctypes as C
from array import array

import numpy as np

from . import lib as _lib


def _triple(key):
    """(chr, start, stop) of a key made with array('L', [chr, start, stop]).tobytes() (create_augmented_gene.py:104)."""
    a = array('L')
    a.frombytes(key)
    return a[0], a[1], a[2]


def _keys(d):
    out = np.empty((len(d), 3), np.int64)
    for i, k in enumerate(d):
        out[i] = _triple(k)
    return out


class NativePlan(object):
    """A planned batch held by the library; `packed` exposes its arrays as numpy views (valid while the plan lives)."""

    def __init__(self, so, handle, reads, n_inst):
        self._so, self.handle, self.reads, self.n_inst = so, handle, reads, n_inst
        self._cols, self._names = None, {}
        inp = so.ultra_plan_input(handle).contents
        self.n_pairs, self.n_cand = int(inp.n_pairs), int(inp.n_cand)
        chr_p, ident_p, cells = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.c_double()
        so.ultra_plan_cands(handle, C.byref(chr_p), C.byref(ident_p), C.byref(cells))
        self.cells = int(cells.value)

        def view(ptr, n, dt):
            return np.ctypeslib.as_array(ptr, shape=(n,)).view(dt) if n > 0 else np.zeros(0, dt)
        self.cand_chr = view(chr_p, self.n_cand, np.int32)
        self.cand_ident_chr = view(ident_p, self.n_cand, np.int32)
        self.packed = {
            "arena": view(inp.arena, int(inp.arena_len), np.uint8),
            "q_off": view(inp.q_off, self.n_pairs, np.int64), "q_len": view(inp.q_len, self.n_pairs, np.int32),
            "t_off": view(inp.t_off, self.n_pairs, np.int64), "t_len": view(inp.t_len, self.n_pairs, np.int32),
            "k": view(inp.k, self.n_pairs, np.int32),
            "cand_kind": view(inp.cand_kind, self.n_cand, np.uint8),
            "cand_e_start": view(inp.cand_e_start, self.n_cand, np.int32), "cand_e_stop": view(inp.cand_e_stop, self.n_cand, np.int32),
            "cand_pair": view(inp.cand_pair, self.n_cand, np.int32),
            "cand_pair_contained": view(inp.cand_pair_contained, self.n_cand, np.int32),
            "cand_seq_len": view(inp.cand_seq_len, self.n_cand, np.int32), "cand_seq_id": view(inp.cand_seq_id, self.n_cand, np.int32),
            "inst_cand_off": view(inp.inst_cand_off, n_inst + 1, np.int64), "inst_read_len": view(inp.inst_read_len, n_inst, np.int32),
            "inst_first_part_stop": view(inp.inst_first_part_stop, n_inst, np.int64),
            "inst_last_part_start": view(inp.inst_last_part_start, n_inst, np.int64),
        }
        ql, tl = self.packed["q_len"].astype(np.int64), self.packed["t_len"].astype(np.int64)
        self.algorithmic_bytes = int(ql.sum() + tl.sum() + 40 * self.n_pairs)

    def mam_id(self, ci, label):
        """The reference's mam_id of candidate ci: str(segm_id) + label (classify_read_with_mams.py:445, :456)."""
        if self._cols is None:  # plain lists: numpy scalar indexing costs more than the name itself
            self._cols = (self.packed["cand_kind"].tolist(), self.packed["cand_e_start"].tolist(),
                          self.packed["cand_e_stop"].tolist(), self.cand_ident_chr.tolist())
        kinds, starts, stops, chrs = self._cols
        key = (kinds[ci], chrs[ci], starts[ci], stops[ci])
        name = self._names.get(key)
        if name is None:  # segments recur across the reads of a batch
            if key[0] in (1, 4, 5):
                name = "flank_{0}_{1}".format(key[2], key[3]) + label
            else:
                name = str(array('L', [key[1], key[2], key[3]]).tobytes()) + label
            self._names[key] = name
        return name

    def close(self):
        if self.handle is not None:
            self.packed = None
            self._so.ultra_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def flatten_instances(instances, cache=None):
    """(solution, read_seq) pairs -> the flat arrays of ultra_plan_request: (nam_off, nam_part[n, 3], nam_x, nam_y,
    reads, read_off)."""
    nam_off, part, xs, ys, reads = [0], [], [], [], []
    cache = {} if cache is None else cache
    for solution, read in instances:
        for nam in solution:
            pid = nam.exon_part_id
            t = cache.get(pid)
            if t is None:
                f = pid.split('^')
                t = cache[pid] = (int(f[0]), int(f[1]), int(f[2]))
            part.append(t)
            xs.append(nam.x); ys.append(nam.y)
        nam_off.append(len(xs))
        reads.append(read.encode() if isinstance(read, str) else bytes(read))
    read_off = np.zeros(len(reads) + 1, np.int64)
    np.cumsum([len(b) for b in reads], out=read_off[1:])
    return (np.asarray(nam_off, np.int64), np.asarray(part, np.int64).reshape(-1, 3), np.asarray(xs, np.int64),
            np.asarray(ys, np.int64), np.frombuffer(b"".join(reads) or b"\0", dtype=np.uint8), read_off)


class DeviceIndex(object):
    """The packed index resident in the HBM of one GPU (ultra_dindex): what ultra_mam_stage_nams plans batches on."""

    def __init__(self, native_index, device=0):
        self._so = native_index._so
        h = C.c_void_p()
        rc = self._so.ultra_dindex_create(int(device), native_index.handle, C.byref(h))
        if rc != 0:
            raise _lib.UltraError(rc, self._so.ultra_dindex_last_error().decode())
        self.handle, self.device = h, int(device)

    def close(self):
        if self.handle is not None:
            self._so.ultra_dindex_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeIndex(object):
    def __init__(self, ref_segment_sequences, ref_flank_sequences, parts_to_segments, segment_to_gene, gene_to_small_segments):
        self._so = _lib.load()
        gene_no = {}
        for g in gene_to_small_segments:
            gene_no.setdefault(g, len(gene_no))
        for genes in segment_to_gene.values():
            for g in genes:
                if g not in gene_no:
                    raise KeyError("gene %r of segment_to_gene has no entry in gene_to_small_segments" % (g,))
        part_key = _keys(parts_to_segments)
        part_off = np.zeros(len(parts_to_segments) + 1, np.int64)
        part_flat = []
        for i, v in enumerate(parts_to_segments.values()):
            part_flat.extend(int(t) for t in v)
            part_off[i + 1] = len(part_flat) // 3
        sg_key = _keys(segment_to_gene)
        sg_off = np.zeros(len(segment_to_gene) + 1, np.int64)
        sg_flat = []
        for i, genes in enumerate(segment_to_gene.values()):
            sg_flat.extend(gene_no[g] for g in genes)
            sg_off[i + 1] = len(sg_flat)
        gs_off = np.zeros(len(gene_no) + 1, np.int64)
        gs_flat = []
        for g, no in gene_no.items():  # insertion order == numbering
            gs_flat.extend(int(t) for t in gene_to_small_segments[g])
            gs_off[no + 1] = len(gs_flat) // 3
        chunks, size = [], 0
        tables = []
        for d in (ref_segment_sequences, ref_flank_sequences):
            off = np.zeros(len(d) + 1, np.int64)
            for i, seq in enumerate(d.values()):
                b = seq.encode() if isinstance(seq, str) else bytes(seq)
                chunks.append(b)
                size += len(b)
                off[i + 1] = size
            tables.append((_keys(d), off))
        # offsets of the second table continue after the first (one pool)
        pool = np.frombuffer(b"".join(chunks) or b"\0", dtype=np.uint8)
        keep = {
            "part_key": part_key.ravel(), "part_off": part_off, "part_flat": np.asarray(part_flat, np.int64),
            "sg_key": sg_key.ravel(), "sg_off": sg_off, "sg_flat": np.asarray(sg_flat, np.int32),
            "gs_off": gs_off, "gs_flat": np.asarray(gs_flat, np.int64),
            "ss_key": tables[0][0].ravel(), "ss_off": tables[0][1], "fs_key": tables[1][0].ravel(), "fs_off": tables[1][1],
            "pool": pool,
        }
        keep["fs_off"][0] = keep["ss_off"][-1]  # first flank starts where the segments end
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        inp = _lib.IndexInput()
        inp.n_parts = len(parts_to_segments); inp.part_key = i64(keep["part_key"]); inp.part_seg_off = i64(keep["part_off"]); inp.part_seg = i64(keep["part_flat"])
        inp.n_seg_gene = len(segment_to_gene); inp.seg_gene_key = i64(keep["sg_key"]); inp.seg_gene_off = i64(keep["sg_off"])
        inp.seg_gene = keep["sg_flat"].ctypes.data_as(C.POINTER(C.c_int32))
        inp.n_genes = len(gene_no); inp.gene_small_off = i64(keep["gs_off"]); inp.gene_small = i64(keep["gs_flat"])
        inp.n_seg_seq = len(ref_segment_sequences); inp.seg_seq_key = i64(keep["ss_key"]); inp.seg_seq_off = i64(keep["ss_off"])
        inp.n_flank_seq = len(ref_flank_sequences); inp.flank_seq_key = i64(keep["fs_key"]); inp.flank_seq_off = i64(keep["fs_off"])
        inp.pool = pool.ctypes.data_as(C.POINTER(C.c_uint8)); inp.pool_len = size
        h = C.c_void_p()
        rc = self._so.ultra_index_create(C.byref(inp), C.byref(h))
        if rc != 0:
            raise _lib.UltraError(rc, self._so.ultra_plan_last_error().decode())
        self.handle = h
        self.pool = pool   # the bytes the device index's pool offsets refer to (segments, then flanks)
        self.n_genes, self.pool_bytes = len(gene_no), size
        self._part_cache = {}

    def plan(self, instances):
        """instances: iterable of (solution, read_seq); solution = the NAM chain (objects with .x, .y, .exon_part_id)."""
        return self.plan_arrays(*flatten_instances(instances, self._part_cache))

    def plan_arrays(self, nam_off, nam_part, nam_x, nam_y, reads, read_off):
        """The same from flat arrays (the fields of ultra_plan_request): no per-NAM Python work at all."""
        a_off = np.ascontiguousarray(nam_off, np.int64); a_part = np.ascontiguousarray(nam_part, np.int64).ravel()
        a_x = np.ascontiguousarray(nam_x, np.int64); a_y = np.ascontiguousarray(nam_y, np.int64)
        a_reads = np.ascontiguousarray(reads, np.uint8); a_roff = np.ascontiguousarray(read_off, np.int64)
        n_inst = len(a_roff) - 1
        if len(a_off) != n_inst + 1 or len(a_part) != 3 * len(a_x) or len(a_x) != len(a_y) or (n_inst and a_off[-1] != len(a_x)):
            raise ValueError("plan_arrays: inconsistent array lengths")
        i64 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int64))
        rq = _lib.PlanRequest()
        rq.n_inst = n_inst; rq.inst_nam_off = i64(a_off); rq.nam_part = i64(a_part); rq.nam_x = i64(a_x); rq.nam_y = i64(a_y)
        rq.reads = a_reads.ctypes.data_as(C.POINTER(C.c_uint8)); rq.read_off = i64(a_roff)
        h = C.c_void_p()
        rc = self._so.ultra_plan_create(self.handle, C.byref(rq), C.byref(h))
        if rc != 0:
            raise _lib.UltraError(rc, self._so.ultra_plan_last_error().decode())
        return NativePlan(self._so, h, np.diff(a_roff).tolist(), n_inst)

    def close(self):
        if self.handle is not None:
            self._so.ultra_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
