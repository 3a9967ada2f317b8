"""ctypes binding of libultra_hip.so (include/ultra_hip.h).

The library is the product: there is no CPU fallback.  Loading fails loudly when the shared
object is missing, and creating a Context fails loudly when no gfx950 device is present.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libultra_hip.so")

ULTRA_OK = 0
ULTRA_E_CAPACITY = -3

_ERR_NAMES = {-1: "ULTRA_E_ARG", -2: "ULTRA_E_HIP", -3: "ULTRA_E_CAPACITY", -4: "ULTRA_E_UNSUPPORTED",
              -5: "ULTRA_E_NODEVICE", -6: "ULTRA_E_STATE"}

EXPORTS = [
    "ultra_cfg_default", "ultra_ctx_create", "ultra_ctx_destroy", "ultra_last_error", "ultra_device_count",
    "ultra_device_info", "ultra_host_alloc", "ultra_host_free", "ultra_debug_canaries", "ultra_myers_hw_batch", "ultra_myers_stage", "ultra_myers_run", "ultra_myers_fetch",
    "ultra_myers_last_timing", "ultra_myers_cell_counts", "ultra_myers_class_timing", "ultra_chain_mams_batch", "ultra_chain_nams_batch",
    "ultra_chain_mams_batch_noalloc", "ultra_chain_noalloc_capacity",
    "ultra_mam_batch", "ultra_mam_stage", "ultra_mam_run", "ultra_mam_run_begin", "ultra_mam_run_end", "ultra_mam_sizes", "ultra_mam_fetch", "ultra_mam_last_timing",
    "ultra_index_create", "ultra_index_destroy", "ultra_plan_create", "ultra_plan_input", "ultra_plan_cands",
    "ultra_plan_destroy", "ultra_plan_last_error", "ultra_pyset_order",
    "ultra_dindex_create", "ultra_dindex_destroy", "ultra_dindex_last_error", "ultra_mam_stage_nams", "ultra_mam_upload_nams",
    "ultra_mam_plan_resident", "ultra_mam_plan_resident_behind", "ultra_mam_counts",
    "ultra_mam_fetch_plan",
    "ultra_pipe_create", "ultra_pipe_destroy", "ultra_pipe_last_error", "ultra_pipe_submit", "ultra_pipe_pending",
    "ultra_pipe_wait", "ultra_pipe_release", "ultra_pipe_timing", "ultra_sg_affine_batch",
    "ultra_front_create", "ultra_front_destroy", "ultra_front_last_error", "ultra_front_prepare", "ultra_front_run",
]


class UltraError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "%s (%d): %s" % (_ERR_NAMES.get(code, "ULTRA_E_?"), code, msg))
        self.code = code


class Cfg(C.Structure):
    _fields_ = [("chain_switch_n", C.c_int32), ("overlap_small", C.c_int32), ("overlap_large", C.c_int32),
                ("loc_slots", C.c_int32), ("banded_traceback", C.c_int32), ("nam_switch_n", C.c_int32),
                ("launch_lanes", C.c_int32), ("banded_starts", C.c_int32)]


_I32P, _I64P, _U8P, _F64P = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.POINTER(C.c_double)


class MamInput(C.Structure):
    """ultra_mam_input (include/ultra_hip.h)."""
    _fields_ = [("arena", _U8P), ("arena_len", C.c_size_t), ("n_pairs", C.c_int32),
                ("q_off", _I64P), ("q_len", _I32P), ("t_off", _I64P), ("t_len", _I32P), ("k", _I32P),
                ("n_cand", C.c_int32), ("cand_kind", _U8P), ("cand_e_start", _I32P), ("cand_e_stop", _I32P),
                ("cand_pair", _I32P), ("cand_pair_contained", _I32P), ("cand_seq_len", _I32P), ("cand_seq_id", _I32P),
                ("n_inst", C.c_int32), ("inst_cand_off", _I64P), ("inst_read_len", _I32P),
                ("inst_first_part_stop", _I64P), ("inst_last_part_start", _I64P), ("cand_chr", _I32P)]


class MamOutput(C.Structure):
    """ultra_mam_output (include/ultra_hip.h)."""
    _fields_ = [("value", _F64P), ("n_mam", _I32P), ("chain_off", _I64P), ("chain_cap", C.c_int64),
                ("chain_x", _I32P), ("chain_y", _I32P), ("chain_c", _I32P), ("chain_d", _I32P), ("chain_val", _F64P),
                ("chain_msl", _I32P), ("chain_cand", _I32P), ("chain_j", _I32P),
                ("mam_cap", C.c_int64), ("mam_off", _I64P),
                ("mam_x", _I32P), ("mam_y", _I32P), ("mam_c", _I32P), ("mam_d", _I32P), ("mam_val", _F64P),
                ("mam_msl", _I32P), ("mam_cand", _I32P), ("chain_kind", _U8P), ("chain_chr", _I32P)]


class IndexInput(C.Structure):
    """ultra_index_input (include/ultra_hip.h)."""
    _fields_ = [("n_parts", C.c_int64), ("part_key", _I64P), ("part_seg_off", _I64P), ("part_seg", _I64P),
                ("n_seg_gene", C.c_int64), ("seg_gene_key", _I64P), ("seg_gene_off", _I64P), ("seg_gene", _I32P),
                ("n_genes", C.c_int64), ("gene_small_off", _I64P), ("gene_small", _I64P),
                ("n_seg_seq", C.c_int64), ("seg_seq_key", _I64P), ("seg_seq_off", _I64P),
                ("n_flank_seq", C.c_int64), ("flank_seq_key", _I64P), ("flank_seq_off", _I64P),
                ("pool", _U8P), ("pool_len", C.c_int64)]


class PlanRequest(C.Structure):
    """ultra_plan_request (include/ultra_hip.h)."""
    _fields_ = [("n_inst", C.c_int32), ("inst_nam_off", _I64P), ("nam_part", _I64P), ("nam_x", _I64P), ("nam_y", _I64P),
                ("reads", _U8P), ("read_off", _I64P)]


class FrontInput(C.Structure):
    """ultra_front_input (include/ultra_hip.h)."""
    _fields_ = [("n_reads", C.c_int32), ("reads", _U8P), ("read_off", _I64P), ("seed_text", _U8P),
                ("fw_begin", _I64P), ("fw_end", _I64P), ("rc_begin", _I64P), ("rc_end", _I64P),
                ("n_chr", C.c_int32), ("max_intron_chr", _I64P), ("max_global_intron", C.c_int64), ("dropoff", C.c_double),
                ("max_loc", C.c_int32), ("polyA_threshold", C.c_int32), ("min_acc", C.c_double), ("n_threads", C.c_int32)]


class FrontView(C.Structure):
    """ultra_front_view (include/ultra_hip.h)."""
    _fields_ = [("n_inst", C.c_int64), ("read_inst_off", _I64P),
                ("n_nam", C.c_int64), ("inst_nam_off", _I64P), ("inst_intron", _I64P), ("nam_y", _I32P), ("nam_c", _I32P), ("nam_d", _I32P),
                ("inst_chr_rc_bound", _I64P),
                ("read_mod", _U8P), ("read_mod_off", _I64P),
                ("n_loc", C.c_int64), ("loc_off", _I64P), ("last_limit", _I64P),
                ("loc_chr", _I32P), ("loc_rc", _U8P), ("loc_score", _I64P),
                ("sol_off", _I64P), ("sol_part", _I64P), ("sol_x", _I64P), ("sol_y", _I64P), ("sol_c", _I32P), ("sol_d", _I32P), ("sol_j", _I32P),
                ("loc_reads", _U8P), ("loc_read_off", _I64P),
                ("n_quadratic", C.c_int64), ("n_nlogn", C.c_int64),
                ("mam", MamOutput), ("ms", C.c_float * 6)]


_lib = None


def load():
    """Loads libultra_hip.so; raises OSError if it has not been built (see __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("libultra_hip.so not found at %s - build it with `make -C ultra_amd/csrc` "
                      "(there is no CPU fallback)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32p, i64p, u8p, f64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
    lib.ultra_cfg_default.argtypes = [C.POINTER(Cfg)]
    lib.ultra_cfg_default.restype = None
    lib.ultra_ctx_create.argtypes = [C.c_int, C.POINTER(Cfg), C.POINTER(vp)]
    lib.ultra_ctx_destroy.argtypes = [vp]
    lib.ultra_ctx_destroy.restype = None
    lib.ultra_last_error.argtypes = [vp]
    lib.ultra_last_error.restype = C.c_char_p
    lib.ultra_device_count.argtypes = []
    lib.ultra_device_info.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ultra_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.ultra_host_free.argtypes = [vp]
    lib.ultra_host_free.restype = None
    lib.ultra_debug_canaries.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]
    lib.ultra_myers_hw_batch.argtypes = [vp, u8p, C.c_size_t, C.c_int32, i64p, i32p, i64p, i32p, i32p,
                                         i32p, i32p, i64p, i32p, i32p, C.c_int64, i64p, i32p, u8p, i64p, i32p]
    lib.ultra_myers_stage.argtypes = [vp, u8p, C.c_size_t, C.c_int32, i64p, i32p, i64p, i32p, i32p]
    lib.ultra_myers_run.argtypes = [vp]
    lib.ultra_myers_fetch.argtypes = [vp, i32p, i32p, i64p, i32p, i32p, C.c_int64, i64p, i32p]
    lib.ultra_myers_last_timing.argtypes = [vp, f64p, C.POINTER(C.c_float)]
    lib.ultra_myers_cell_counts.argtypes = [vp, f64p, f64p, i64p]
    lib.ultra_myers_class_timing.argtypes = [vp, C.c_int32, i32p, i32p, f64p, C.POINTER(C.c_float), i32p]
    lib.ultra_chain_mams_batch.argtypes = [vp, C.c_int32, i64p, i32p, i32p, i32p, i32p, f64p, C.c_int32, C.c_int32,
                                           C.c_int32, i32p, i32p, f64p]
    lib.ultra_chain_mams_batch_noalloc.argtypes = lib.ultra_chain_mams_batch.argtypes
    lib.ultra_chain_noalloc_capacity.argtypes = []
    lib.ultra_chain_noalloc_capacity.restype = C.c_int64
    lib.ultra_chain_nams_batch.argtypes = [vp, C.c_int32, i64p, i32p, i32p, i32p, i64p, C.c_int32, i64p, i32p, i64p]
    lib.ultra_mam_batch.argtypes = [vp, C.POINTER(MamInput), C.c_double, C.POINTER(MamOutput)]
    lib.ultra_mam_stage.argtypes = [vp, C.POINTER(MamInput)]
    lib.ultra_mam_run.argtypes = [vp, C.c_double]
    lib.ultra_mam_run_begin.argtypes = [vp, C.c_double]
    lib.ultra_mam_run_end.argtypes = [vp]
    lib.ultra_mam_sizes.argtypes = [vp, i64p, i64p]
    lib.ultra_mam_fetch.argtypes = [vp, C.POINTER(MamOutput)]
    lib.ultra_mam_last_timing.argtypes = [vp, C.POINTER(C.c_float)]
    lib.ultra_index_create.argtypes = [C.POINTER(IndexInput), C.POINTER(vp)]
    lib.ultra_index_destroy.argtypes = [vp]
    lib.ultra_index_destroy.restype = None
    lib.ultra_plan_create.argtypes = [vp, C.POINTER(PlanRequest), C.POINTER(vp)]
    lib.ultra_plan_input.argtypes = [vp]
    lib.ultra_plan_input.restype = C.POINTER(MamInput)
    lib.ultra_plan_cands.argtypes = [vp, C.POINTER(i32p), C.POINTER(i32p), f64p]
    lib.ultra_plan_destroy.argtypes = [vp]
    lib.ultra_plan_destroy.restype = None
    lib.ultra_plan_last_error.argtypes = []
    lib.ultra_plan_last_error.restype = C.c_char_p
    lib.ultra_pyset_order.argtypes = [i64p, C.c_int32, i32p]
    lib.ultra_dindex_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.ultra_dindex_destroy.argtypes = [vp]
    lib.ultra_dindex_destroy.restype = None
    lib.ultra_dindex_last_error.argtypes = []
    lib.ultra_dindex_last_error.restype = C.c_char_p
    lib.ultra_mam_stage_nams.argtypes = [vp, vp, C.POINTER(PlanRequest)]
    lib.ultra_mam_upload_nams.argtypes = [vp, C.POINTER(PlanRequest)]
    lib.ultra_mam_plan_resident.argtypes = [vp, vp]
    lib.ultra_mam_plan_resident_behind.argtypes = [vp, vp, vp]
    lib.ultra_mam_counts.argtypes = [vp, i32p, i32p, i32p]
    lib.ultra_mam_fetch_plan.argtypes = [vp, i64p, i32p, i64p, i64p, u8p, i32p, i32p, i32p, i32p, i32p, i32p, i32p,
                                         i64p, i32p, i64p, i32p, i32p]
    lib.ultra_sg_affine_batch.argtypes = [vp, u8p, C.c_size_t, C.c_int32, i64p, i32p, i64p, i32p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, i32p, i32p, i32p, u8p, i64p, i32p]
    lib.ultra_pipe_create.argtypes = [C.c_int, C.POINTER(Cfg), vp, C.c_int32, C.POINTER(vp)]
    lib.ultra_pipe_destroy.argtypes = [vp]
    lib.ultra_pipe_destroy.restype = None
    lib.ultra_pipe_last_error.argtypes = [vp]
    lib.ultra_pipe_last_error.restype = C.c_char_p
    lib.ultra_pipe_submit.argtypes = [vp, C.POINTER(PlanRequest), C.c_double]
    lib.ultra_pipe_pending.argtypes = [vp]
    lib.ultra_pipe_wait.argtypes = [vp, i32p, i64p, C.POINTER(MamOutput)]
    lib.ultra_pipe_release.argtypes = [vp]
    lib.ultra_pipe_timing.argtypes = [vp, C.POINTER(C.c_float), f64p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.ultra_front_create.argtypes = [C.POINTER(vp)]
    lib.ultra_front_destroy.argtypes = [vp]
    lib.ultra_front_destroy.restype = None
    lib.ultra_front_last_error.argtypes = [vp]
    lib.ultra_front_last_error.restype = C.c_char_p
    lib.ultra_front_prepare.argtypes = [vp, C.POINTER(FrontInput), C.POINTER(FrontView)]
    lib.ultra_front_run.argtypes = [vp, vp, vp, vp, C.POINTER(FrontInput), C.c_int32, C.POINTER(FrontView)]
    _lib = lib
    return lib


def _ptr(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _arr(v, dt):
    return np.ascontiguousarray(v, dtype=dt)


class Context(object):
    """One GPU context (ultra_ctx). Not thread-safe; use one per GPU/process."""

    def __init__(self, device=0, cfg=None):
        self.lib = load()
        self._h = C.c_void_p()
        c = Cfg()
        self.lib.ultra_cfg_default(C.byref(c))
        if cfg:
            for k, v in cfg.items():
                setattr(c, k, v)
        self.cfg = c
        rc = self.lib.ultra_ctx_create(int(device), C.byref(c), C.byref(self._h))
        if rc != ULTRA_OK:
            raise UltraError(rc, "ultra_ctx_create(device=%d) failed: no usable gfx950 device" % device)

    def close(self):
        if self._h:
            self.lib.ultra_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != ULTRA_OK:
            raise UltraError(rc, (self.lib.ultra_last_error(self._h) or b"").decode())

    def device_info(self):
        name = C.create_string_buffer(64)
        ncu, clk = C.c_int(), C.c_int()
        self._check(self.lib.ultra_device_info(self._h, name, 64, C.byref(ncu), C.byref(clk)))
        return {"arch": name.value.decode(), "n_cu": ncu.value, "clock_khz": clk.value}

    # ---- Myers -------------------------------------------------------------------------------
    def myers_hw_batch(self, arena, q_off, q_len, t_off, t_len, k, want_ops=False):
        """Returns dict(ed, n_loc, loc_off, loc_start, loc_end, first_ins[, ops, ops_off, n_ops])."""
        arena = _arr(arena, np.uint8)
        q_off, t_off = _arr(q_off, np.int64), _arr(t_off, np.int64)
        q_len, t_len, k = _arr(q_len, np.int32), _arr(t_len, np.int32), _arr(k, np.int32)
        n = len(q_len)
        ed = np.zeros(n, np.int32); nl = np.zeros(n, np.int32); fi = np.zeros(n, np.int32)
        lo = np.zeros(n + 1, np.int64)
        ops = ops_off = n_ops = None
        if want_ops:
            cap = q_len.astype(np.int64) + np.minimum(t_len.astype(np.int64), 2 * q_len.astype(np.int64))
            ops_off = np.zeros(n + 1, np.int64); np.cumsum(cap, out=ops_off[1:])
            ops = np.zeros(max(int(ops_off[-1]), 1), np.uint8)
            n_ops = np.zeros(n, np.int32)
        loc_cap = max(4 * n, 16)
        need = C.c_int64(0)
        first = True
        while True:
            ls = np.zeros(loc_cap, np.int32); le = np.zeros(loc_cap, np.int32)
            if first:
                rc = self.lib.ultra_myers_hw_batch(self._h, _ptr(arena, C.c_uint8), arena.size, n,
                                                   _ptr(q_off, C.c_int64), _ptr(q_len, C.c_int32),
                                                   _ptr(t_off, C.c_int64), _ptr(t_len, C.c_int32), _ptr(k, C.c_int32),
                                                   _ptr(ed, C.c_int32), _ptr(nl, C.c_int32), _ptr(lo, C.c_int64),
                                                   _ptr(ls, C.c_int32), _ptr(le, C.c_int32), loc_cap, C.byref(need),
                                                   _ptr(fi, C.c_int32), _ptr(ops, C.c_uint8), _ptr(ops_off, C.c_int64),
                                                   _ptr(n_ops, C.c_int32))
                if rc == ULTRA_E_CAPACITY and need.value > loc_cap and not want_ops:
                    loc_cap = int(need.value); first = False
                    continue
                if rc == ULTRA_E_CAPACITY and need.value > loc_cap:
                    loc_cap = int(need.value)  # edit scripts live only inside the one-shot call: run it again
                    continue
            else:
                rc = self.lib.ultra_myers_fetch(self._h, _ptr(ed, C.c_int32), _ptr(nl, C.c_int32), _ptr(lo, C.c_int64),
                                                _ptr(ls, C.c_int32), _ptr(le, C.c_int32), loc_cap, C.byref(need),
                                                _ptr(fi, C.c_int32))
            self._check(rc)
            break
        out = {"ed": ed, "n_loc": nl, "loc_off": lo, "loc_start": ls[:int(lo[-1])], "loc_end": le[:int(lo[-1])],
               "first_ins": fi}
        if want_ops:
            out.update(ops=ops, ops_off=ops_off, n_ops=n_ops)
        return out

    def myers_stage(self, arena, q_off, q_len, t_off, t_len, k):
        arena = _arr(arena, np.uint8)
        q_off, t_off = _arr(q_off, np.int64), _arr(t_off, np.int64)
        q_len, t_len, k = _arr(q_len, np.int32), _arr(t_len, np.int32), _arr(k, np.int32)
        self._staged_n = len(q_len)
        self._check(self.lib.ultra_myers_stage(self._h, _ptr(arena, C.c_uint8), arena.size, len(q_len),
                                               _ptr(q_off, C.c_int64), _ptr(q_len, C.c_int32), _ptr(t_off, C.c_int64),
                                               _ptr(t_len, C.c_int32), _ptr(k, C.c_int32)))

    def myers_run(self):
        self._check(self.lib.ultra_myers_run(self._h))

    def myers_timing(self):
        cells = C.c_double()
        ms = (C.c_float * 5)()
        self._check(self.lib.ultra_myers_last_timing(self._h, C.byref(cells), ms))
        return cells.value, [ms[i] for i in range(5)]

    def myers_cell_counts(self):
        """Of the staged batch: (cells the forward pass computes, cells of the pairs answered at stage time because the
        query is more than k longer than the target, number of those pairs)."""
        a, b, n = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.ultra_myers_cell_counts(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def myers_class_timing(self):
        """{(G, W): (cells, forward_ms)} of the last run."""
        cap = 32
        G = (C.c_int32 * cap)(); W = (C.c_int32 * cap)(); cells = (C.c_double * cap)(); ms = (C.c_float * cap)()
        n = C.c_int32()
        self._check(self.lib.ultra_myers_class_timing(self._h, cap, G, W, cells, ms, C.byref(n)))
        return {(G[i], W[i]): (cells[i], ms[i]) for i in range(min(n.value, cap))}

    def myers_fetch(self):
        n = self._staged_n
        ed = np.zeros(n, np.int32); nl = np.zeros(n, np.int32); fi = np.zeros(n, np.int32)
        lo = np.zeros(n + 1, np.int64)
        loc_cap = max(4 * n, 16)
        need = C.c_int64(0)
        while True:
            ls = np.zeros(loc_cap, np.int32); le = np.zeros(loc_cap, np.int32)
            rc = self.lib.ultra_myers_fetch(self._h, _ptr(ed, C.c_int32), _ptr(nl, C.c_int32), _ptr(lo, C.c_int64),
                                            _ptr(ls, C.c_int32), _ptr(le, C.c_int32), loc_cap, C.byref(need),
                                            _ptr(fi, C.c_int32))
            if rc == ULTRA_E_CAPACITY and need.value > loc_cap:
                loc_cap = int(need.value)
                continue
            self._check(rc)
            break
        return {"ed": ed, "n_loc": nl, "loc_off": lo, "loc_start": ls[:int(lo[-1])], "loc_end": le[:int(lo[-1])],
                "first_ins": fi}

    # ---- fused MAM stage -----------------------------------------------------------------------
    def mam_stage(self, packed):
        """packed: dict of numpy arrays named like the fields of ultra_mam_input (host -> HBM)."""
        spec = [("arena", np.uint8, C.c_uint8), ("q_off", np.int64, C.c_int64), ("q_len", np.int32, C.c_int32),
                ("t_off", np.int64, C.c_int64), ("t_len", np.int32, C.c_int32), ("k", np.int32, C.c_int32),
                ("cand_kind", np.uint8, C.c_uint8), ("cand_e_start", np.int32, C.c_int32), ("cand_e_stop", np.int32, C.c_int32),
                ("cand_pair", np.int32, C.c_int32), ("cand_pair_contained", np.int32, C.c_int32),
                ("cand_seq_len", np.int32, C.c_int32), ("cand_seq_id", np.int32, C.c_int32),
                ("inst_cand_off", np.int64, C.c_int64), ("inst_read_len", np.int32, C.c_int32),
                ("inst_first_part_stop", np.int64, C.c_int64), ("inst_last_part_start", np.int64, C.c_int64)]
        keep, inp = {}, MamInput()
        for name, dt, ct in spec:
            arr = _arr(packed[name], dt)
            keep[name] = arr
            setattr(inp, name, _ptr(arr, ct))
        inp.arena_len = keep["arena"].size
        inp.n_pairs = len(keep["k"])
        inp.n_cand = len(keep["cand_kind"])
        inp.n_inst = len(keep["inst_read_len"])
        self._mam_n_inst = inp.n_inst
        self._staged_n = inp.n_pairs  # myers_fetch works on the stage's pairs too
        self._check(self.lib.ultra_mam_stage(self._h, C.byref(inp)))

    def mam_stage_plan(self, plan):
        """plan: ultra_amd.index.NativePlan - its ultra_mam_input goes to the device as it is (no numpy round trip)."""
        inp = self.lib.ultra_plan_input(plan.handle)
        self._mam_n_inst = int(inp.contents.n_inst)
        self._staged_n = int(inp.contents.n_pairs)
        self._check(self.lib.ultra_mam_stage(self._h, inp))

    def mam_stage_nams(self, dindex, nam_off, nam_part, nam_x, nam_y, reads, read_off, upload_only=False):
        """Device-planned batch (ultra_mam_stage_nams): the NAM chains and reads of the instances as flat arrays (the
        fields of ultra_plan_request); candidates and pairs are produced on the GPU over the resident index.
        upload_only: only the copy to the device (ultra_mam_upload_nams); mam_plan_resident() plans from it."""
        a_off = _arr(nam_off, np.int64); a_part = _arr(nam_part, np.int64).ravel()
        a_x = _arr(nam_x, np.int64); a_y = _arr(nam_y, np.int64)
        a_reads = _arr(reads, np.uint8); a_roff = _arr(read_off, np.int64)
        n_inst = len(a_roff) - 1
        if len(a_off) != n_inst + 1 or len(a_part) != 3 * len(a_x) or len(a_x) != len(a_y) or (n_inst and a_off[-1] != len(a_x)):
            raise ValueError("mam_stage_nams: inconsistent array lengths")
        rq = PlanRequest()
        rq.n_inst = n_inst; rq.inst_nam_off = _ptr(a_off, C.c_int64); rq.nam_part = _ptr(a_part, C.c_int64)
        rq.nam_x = _ptr(a_x, C.c_int64); rq.nam_y = _ptr(a_y, C.c_int64)
        rq.reads = _ptr(a_reads, C.c_uint8); rq.read_off = _ptr(a_roff, C.c_int64)
        self._mam_n_inst = n_inst
        if upload_only:
            self._check(self.lib.ultra_mam_upload_nams(self._h, C.byref(rq)))
            return
        self._check(self.lib.ultra_mam_stage_nams(self._h, dindex.handle, C.byref(rq)))
        self._staged_n = self.mam_counts()[2]

    def mam_plan_resident(self, dindex):
        """Plans and stages the uploaded request again, from HBM (ultra_mam_plan_resident)."""
        self._check(self.lib.ultra_mam_plan_resident(self._h, dindex.handle))

    def mam_plan_resident_behind(self, dindex, front):
        """mam_plan_resident with the planner's kernels behind the start pass of the run `front` (another Context of this
        GPU) has in flight (ultra_mam_plan_resident_behind)."""
        self._check(self.lib.ultra_mam_plan_resident_behind(self._h, dindex.handle, front._h if front is not None else None))
        self._staged_n = self.mam_counts()[2]

    def mam_counts(self):
        """(instances, candidates, pairs) of the staged batch."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self.lib.ultra_mam_counts(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def mam_fetch_plan(self):
        """The staged batch's candidate / pair arrays as the device holds them (pairs in sorted order)."""
        ni, nc, npairs = self.mam_counts()
        r = {"inst_cand_off": np.zeros(ni + 1, np.int64), "inst_read_len": np.zeros(max(ni, 1), np.int32),
             "inst_first_part_stop": np.zeros(max(ni, 1), np.int64), "inst_last_part_start": np.zeros(max(ni, 1), np.int64),
             "cand_kind": np.zeros(max(nc, 1), np.uint8)}
        for f in ("cand_e_start", "cand_e_stop", "cand_pair", "cand_pair_contained", "cand_seq_len", "cand_seq_id", "cand_chr"):
            r[f] = np.zeros(max(nc, 1), np.int32)
        r["q_off"] = np.zeros(max(npairs, 1), np.int64); r["t_off"] = np.zeros(max(npairs, 1), np.int64)
        for f in ("q_len", "t_len", "k"):
            r[f] = np.zeros(max(npairs, 1), np.int32)
        self._check(self.lib.ultra_mam_fetch_plan(
            self._h, _ptr(r["inst_cand_off"], C.c_int64), _ptr(r["inst_read_len"], C.c_int32), _ptr(r["inst_first_part_stop"], C.c_int64),
            _ptr(r["inst_last_part_start"], C.c_int64), _ptr(r["cand_kind"], C.c_uint8), _ptr(r["cand_e_start"], C.c_int32),
            _ptr(r["cand_e_stop"], C.c_int32), _ptr(r["cand_pair"], C.c_int32), _ptr(r["cand_pair_contained"], C.c_int32),
            _ptr(r["cand_seq_len"], C.c_int32), _ptr(r["cand_seq_id"], C.c_int32), _ptr(r["cand_chr"], C.c_int32),
            _ptr(r["q_off"], C.c_int64), _ptr(r["q_len"], C.c_int32), _ptr(r["t_off"], C.c_int64), _ptr(r["t_len"], C.c_int32),
            _ptr(r["k"], C.c_int32)))
        for f in list(r):
            n = ni if f.startswith("inst_") and f != "inst_cand_off" else nc if f.startswith("cand_") else npairs if f in ("q_off", "t_off", "q_len", "t_len", "k") else None
            if n is not None:
                r[f] = r[f][:n]
        return r

    def mam_run(self, min_acc):
        self._check(self.lib.ultra_mam_run(self._h, float(min_acc)))

    def mam_run_begin(self, min_acc):
        """Enqueues the whole run and returns with its tail in flight (ultra_mam_run_begin); mam_run_end() closes it."""
        self._check(self.lib.ultra_mam_run_begin(self._h, float(min_acc)))

    def mam_run_end(self):
        self._check(self.lib.ultra_mam_run_end(self._h))

    def mam_timing(self):
        ms = (C.c_float * 3)()
        self._check(self.lib.ultra_mam_last_timing(self._h, ms))
        return [ms[i] for i in range(3)]

    def mam_fetch(self, want_all=False, want_names=False):
        """Results of the last mam_run as numpy arrays (chains, and optionally the full MAM lists).  want_names: also
        chain_kind / chain_chr, from which the mam_id and ref_chr_id of every chain element follow."""
        ni = self._mam_n_inst
        nc, nm = C.c_int64(), C.c_int64()
        self._check(self.lib.ultra_mam_sizes(self._h, C.byref(nc), C.byref(nm)))
        tc, tm = max(int(nc.value), 1), max(int(nm.value), 1)
        r = {"value": np.zeros(max(ni, 1), np.float64), "n_mam": np.zeros(max(ni, 1), np.int32),
             "chain_off": np.zeros(ni + 1, np.int64)}
        for f in ("x", "y", "c", "d", "msl", "cand", "j"):
            r["chain_" + f] = np.zeros(tc, np.int32)
        r["chain_val"] = np.zeros(tc, np.float64)
        out = MamOutput()
        out.value, out.n_mam, out.chain_off = _ptr(r["value"], C.c_double), _ptr(r["n_mam"], C.c_int32), _ptr(r["chain_off"], C.c_int64)
        out.chain_cap = tc
        for f in ("x", "y", "c", "d", "msl", "cand", "j"):
            setattr(out, "chain_" + f, _ptr(r["chain_" + f], C.c_int32))
        out.chain_val = _ptr(r["chain_val"], C.c_double)
        if want_names:
            r["chain_kind"] = np.zeros(tc, np.uint8); r["chain_chr"] = np.zeros(tc, np.int32)
            out.chain_kind, out.chain_chr = _ptr(r["chain_kind"], C.c_uint8), _ptr(r["chain_chr"], C.c_int32)
        if want_all:
            r["mam_off"] = np.zeros(ni + 1, np.int64)
            out.mam_off, out.mam_cap = _ptr(r["mam_off"], C.c_int64), tm
            for f in ("x", "y", "c", "d", "msl", "cand"):
                r["mam_" + f] = np.zeros(tm, np.int32)
                setattr(out, "mam_" + f, _ptr(r["mam_" + f], C.c_int32))
            r["mam_val"] = np.zeros(tm, np.float64)
            out.mam_val = _ptr(r["mam_val"], C.c_double)
        self._check(self.lib.ultra_mam_fetch(self._h, C.byref(out)))
        r["value"], r["n_mam"] = r["value"][:ni], r["n_mam"][:ni]
        return r

    # ---- base-level alignment ------------------------------------------------------------------
    def sg_affine_batch(self, arena, q_off, q_len, t_off, t_len, match=2, mismatch=-2, gap_open=3, gap_extend=1, want_ops=True):
        """Semi-global affine-gap alignment of a batch of pairs (ultra_sg_affine_batch).  Returns dict(score, end_query,
        end_ref[, ops, ops_off, n_ops]); the edit script of pair p is ops[ops_off[p] : ops_off[p] + n_ops[p]], forward."""
        arena = _arr(arena, np.uint8)
        q_off, t_off = _arr(q_off, np.int64), _arr(t_off, np.int64)
        q_len, t_len = _arr(q_len, np.int32), _arr(t_len, np.int32)
        n = len(q_len)
        score = np.zeros(max(n, 1), np.int32); eq = np.zeros(max(n, 1), np.int32); et = np.zeros(max(n, 1), np.int32)
        ops = ops_off = n_ops = None
        if want_ops:
            ops_off = np.zeros(n + 1, np.int64)
            np.cumsum(q_len.astype(np.int64) + t_len.astype(np.int64), out=ops_off[1:])
            ops = np.zeros(max(int(ops_off[-1]), 1), np.uint8); n_ops = np.zeros(max(n, 1), np.int32)
        self._check(self.lib.ultra_sg_affine_batch(self._h, _ptr(arena, C.c_uint8), arena.size, n, _ptr(q_off, C.c_int64),
                                                   _ptr(q_len, C.c_int32), _ptr(t_off, C.c_int64), _ptr(t_len, C.c_int32),
                                                   int(match), int(mismatch), int(gap_open), int(gap_extend),
                                                   _ptr(score, C.c_int32), _ptr(eq, C.c_int32), _ptr(et, C.c_int32),
                                                   _ptr(ops, C.c_uint8), _ptr(ops_off, C.c_int64), _ptr(n_ops, C.c_int32)))
        out = {"score": score[:n], "end_query": eq[:n], "end_ref": et[:n]}
        if want_ops:
            out.update(ops=ops, ops_off=ops_off, n_ops=n_ops[:n])
        return out

    # ---- chaining ----------------------------------------------------------------------------
    def chain_mams_batch(self, inst_off, x, y, c, d, val, solver=0, overlap_small=-1, overlap_large=-1, noalloc=False):
        """noalloc: through ultra_chain_mams_batch_noalloc (fixed device scratch inside the library, no hipMalloc)."""
        inst_off = _arr(inst_off, np.int64)
        x, y, c, d = _arr(x, np.int32), _arr(y, np.int32), _arr(c, np.int32), _arr(d, np.int32)
        val = _arr(val, np.float64)
        ni = len(inst_off) - 1
        idx = np.zeros(max(len(c), 1), np.int32); ln = np.zeros(max(ni, 1), np.int32)
        value = np.zeros(max(ni, 1), np.float64)
        fn = self.lib.ultra_chain_mams_batch_noalloc if noalloc else self.lib.ultra_chain_mams_batch
        self._check(fn(self._h, ni, _ptr(inst_off, C.c_int64), _ptr(x, C.c_int32),
                      _ptr(y, C.c_int32), _ptr(c, C.c_int32), _ptr(d, C.c_int32),
                      _ptr(val, C.c_double), solver, overlap_small, overlap_large,
                      _ptr(idx, C.c_int32), _ptr(ln, C.c_int32), _ptr(value, C.c_double)))
        return idx, ln[:ni], value[:ni]

    def chain_nams_batch(self, inst_off, y, c, d, max_intron=None, solver=0):
        """NAM chaining of a batch; returns (C, trace, value): the DP vectors (n + 1 entries per instance,
        instance i at inst_off[i] + i) and max(C) per instance."""
        inst_off = _arr(inst_off, np.int64)
        c, d = _arr(c, np.int32), _arr(d, np.int32)
        y = None if y is None else _arr(y, np.int32)
        ni = len(inst_off) - 1
        mi = None if max_intron is None else _arr(np.broadcast_to(np.asarray(max_intron, np.int64), (ni,)), np.int64)
        Cv = np.zeros(len(c) + max(ni, 1), np.int64); tr = np.zeros(len(c) + max(ni, 1), np.int32)
        value = np.zeros(max(ni, 1), np.int64)
        self._check(self.lib.ultra_chain_nams_batch(self._h, ni, _ptr(inst_off, C.c_int64),
                                                    None if y is None else _ptr(y, C.c_int32), _ptr(c, C.c_int32),
                                                    _ptr(d, C.c_int32), None if mi is None else _ptr(mi, C.c_int64), solver,
                                                    _ptr(Cv, C.c_int64), _ptr(tr, C.c_int32), _ptr(value, C.c_int64)))
        return Cv[:len(c) + ni], tr[:len(c) + ni], value[:ni]


class PinnedArray(object):
    """A numpy array in page-locked host memory (ultra_host_alloc): `.a` is the array; free with close()."""

    def __init__(self, src=None, shape=None, dtype=None):
        so = load()
        if src is not None:
            src = np.ascontiguousarray(src)
            shape, dtype = src.shape, src.dtype
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        rc = so.ultra_host_alloc(max(n, 1), C.byref(p))
        if rc != 0 or not p.value:
            raise MemoryError("ultra_host_alloc(%d) failed" % n)
        self._so, self._p = so, p
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        self.a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        if src is not None:
            self.a[...] = src

    def close(self):
        if self._p is not None:
            self.a = None
            self._so.ultra_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipe(object):
    """ultra_pipe: `depth` batches in flight on one GPU (stage | run | fetch of consecutive batches overlap).

        pipe = Pipe(native_index, device=0, depth=3)
        pipe.submit(nam_off, nam_part, nam_x, nam_y, reads, read_off, min_acc)      # returns at once
        res = pipe.wait()          # oldest batch: dict of numpy views into pinned memory ...
        pipe.release()             # ... valid until here
    """

    def __init__(self, native_index, device=0, depth=3, cfg=None):
        self.lib = load()
        self._h = C.c_void_p()
        c = Cfg()
        self.lib.ultra_cfg_default(C.byref(c))
        for k, v in (cfg or {}).items():
            setattr(c, k, v)
        self._index = native_index      # the pipe reads the host index when it has to plan a batch on the host
        self._keep = []
        rc = self.lib.ultra_pipe_create(int(device), C.byref(c), native_index.handle, int(depth), C.byref(self._h))
        if rc != ULTRA_OK:
            raise UltraError(rc, "ultra_pipe_create(device=%d, depth=%d) failed" % (device, depth))
        self.depth = int(depth)

    def close(self):
        if self._h:
            self.lib.ultra_pipe_destroy(self._h)
            self._h = C.c_void_p()
            self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != ULTRA_OK:
            raise UltraError(rc, (self.lib.ultra_pipe_last_error(self._h) or b"").decode())

    def pending(self):
        return self.lib.ultra_pipe_pending(self._h)

    def submit(self, nam_off, nam_part, nam_x, nam_y, reads, read_off, min_acc):
        arrs = (_arr(nam_off, np.int64), _arr(nam_part, np.int64).ravel(), _arr(nam_x, np.int64), _arr(nam_y, np.int64),
                _arr(reads, np.uint8), _arr(read_off, np.int64))
        rq = PlanRequest()
        rq.n_inst = len(arrs[5]) - 1
        rq.inst_nam_off = _ptr(arrs[0], C.c_int64); rq.nam_part = _ptr(arrs[1], C.c_int64)
        rq.nam_x = _ptr(arrs[2], C.c_int64); rq.nam_y = _ptr(arrs[3], C.c_int64)
        rq.reads = _ptr(arrs[4], C.c_uint8); rq.read_off = _ptr(arrs[5], C.c_int64)
        self._check(self.lib.ultra_pipe_submit(self._h, C.byref(rq), float(min_acc)))
        self._keep.append(arrs)         # the library reads them until the batch has been waited for

    def wait(self):
        ni, nc = C.c_int32(), C.c_int64()
        v = MamOutput()
        rc = self.lib.ultra_pipe_wait(self._h, C.byref(ni), C.byref(nc), C.byref(v))
        if self._keep:
            self._keep.pop(0)
        self._check(rc)
        n, t = ni.value, nc.value

        def view(ptr, cnt, dt):
            return np.ctypeslib.as_array(ptr, shape=(cnt,)).view(dt) if cnt > 0 else np.zeros(0, dt)
        r = {"value": view(v.value, n, np.float64), "n_mam": view(v.n_mam, n, np.int32), "chain_off": view(v.chain_off, n + 1, np.int64),
             "chain_val": view(v.chain_val, t, np.float64), "chain_kind": view(v.chain_kind, t, np.uint8)}
        for f in ("x", "y", "c", "d", "msl", "cand", "j", "chr"):
            r["chain_" + f] = view(getattr(v, "chain_" + f), t, np.int32)
        return r

    def release(self):
        self._check(self.lib.ultra_pipe_release(self._h))

    def timing(self):
        ms = (C.c_float * 4)(); my = (C.c_float * 5)(); mm = (C.c_float * 3)(); cells = C.c_double()
        self._check(self.lib.ultra_pipe_timing(self._h, ms, C.byref(cells), my, mm))
        return {"stage_ms": ms[0], "wait_ms": ms[1], "run_ms": ms[2], "fetch_ms": ms[3], "cells": cells.value,
                "myers_ms": [my[i] for i in range(5)], "mam_ms": [mm[i] for i in range(3)]}


_default_ctx = None


def default_context():
    """Process-wide context on device LOCAL_RANK (or 0)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_ctx
