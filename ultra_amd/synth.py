"""Synthetic annotation + long-read generator for tests and bench.py.

GRCh38 / GENCODE / namfinder are not available offline, so every BASELINE.json config is run on
self-generated data of the same shape:
  * gene models: exons log-normal (median ~130 bp, a few percent forced into 5-30 bp), introns
    log-normal, 1-30 exons per gene, several transcripts per gene;
  * index dictionaries in the exact layout the reference's hot path reads
    (/root/reference/modules/create_augmented_gene.py:63-263, keys = array('L',[chr,start,stop]).tobytes(),
    values = flat array('L') triples);
  * reads: sub-intervals of transcripts with the reference simulator's error mix - of the erroneous
    positions 50 % deletions / 30 % substitutions / 20 % insertions, geometric extension p = 0.5
    (/root/reference/evaluation/simulate_reads.py:77-102);
  * a NAM chain per read (`solution`): the exact-match runs between read and genome that are at
    least `min_nam` long, as Mem tuples in the form align.py hands to the MAM stage
    (/root/reference/modules/align.py:22, :357-388).
Everything is deterministic in (seed, parameters).
"""
from array import array
from collections import namedtuple

import numpy as np

Mem = namedtuple('Mem', ['x', 'y', 'c', 'd', 'val', 'j', "exon_part_id"])

PRESETS = {
    # read models of the BASELINE.json configs.  The Python simulator below draws
    # lengths from N(len_mean, len_sd); the native simulator (make_reads_native) also knows the log-normal model
    # of C5: median len_median, sigma len_sigma, clipped to [len_min, len_max], 3'-biased truncation.
    "isoseq": dict(err=0.01, len_mean=2500, len_sd=800, len_min=300, min_acc=0.8),
    "ont_cdna": dict(err=0.07, len_mean=1000, len_sd=400, len_min=200, min_acc=0.6),
    "ont_drna": dict(err=0.12, len_mean=1100, len_sd=900, len_min=200, min_acc=0.6,
                     len_median=1100, len_sigma=0.75, len_max=15000, bias3p=0.7),
}

# annotation shapes: "small" = the toy annotation of the parity tests; "c3" = SURVEY.md section 8d C3, GRCh38 /
# GENCODE-like: ~60 k genes on 24 chromosomes, merged exon parts per gene (alternative splice sites cut a part
# into 2-3 segments), a share of overlapping genes that share exon parts (a hit on a shared segment pulls in the
# small segments of every gene it belongs to, classify_read_with_mams.py:264-271).
WORLDS = {
    "small": dict(n_genes=1500, n_chr=8),
    "c2": dict(n_genes=900, n_chr=1, exon_mean=12.0, max_exons=60, split_frac=0.45, overlap_frac=0.15),
    "c3": dict(n_genes=60000, n_chr=24, exon_mean=12.0, max_exons=60, split_frac=0.45, overlap_frac=0.15, tr_max=7),  # ~240 k transcripts
}


def key_of(chr_id, a, b):
    return array('L', [int(chr_id), int(a), int(b)]).tobytes()


class World:
    """Index dictionaries + bookkeeping needed to simulate reads."""

    def __init__(self):
        self.ref_segment_sequences = {}
        self.ref_flank_sequences = {}
        self.parts_to_segments = {}
        self.segment_to_gene = {}
        self.gene_to_small_segments = {}
        self.transcripts = []   # list of (chr_id, gene_id, [(part_start, part_stop), ...], up_flank, down_flank)
        self.part_seq = {}      # (chr, start, stop) -> str   (exon parts and flanks)

    def index(self):
        return (self.ref_segment_sequences, self.ref_flank_sequences, self.parts_to_segments,
                self.segment_to_gene, self.gene_to_small_segments)


_ACGT = np.frombuffer(b"ACGT", np.uint8)


def _rand_seq(rng, n):
    return _ACGT[rng.randint(0, 4, size=n)].tobytes().decode()


def make_world(seed=20241008, n_genes=200, n_chr=2, flank=1000, small_thr=200, max_exons=30,
               small_exon_frac=0.03, split_frac=0.3, exon_mean=9.0, overlap_frac=0.0, tr_max=4):
    """overlap_frac: share of genes that overlap the previous gene of their chromosome - they reuse a run of its
    exon parts (same coordinates, so the shared segments belong to both genes) and add exons of their own behind it."""
    rng = np.random.RandomState(seed)
    w = World()
    pos = [5000] * n_chr
    last_gene = [None] * n_chr      # (exons, cuts per exon) of the previous gene on each chromosome
    for g in range(n_genes):
        chr_id = g % n_chr
        gene_id = "gene%05d" % g
        n_ex = int(min(max_exons, max(1, rng.geometric(1.0 / exon_mean))))
        shared = []
        if overlap_frac > 0.0 and last_gene[chr_id] is not None and rng.rand() < overlap_frac:
            prev = last_gene[chr_id]
            k = int(rng.randint(1, len(prev) + 1))
            a0 = int(rng.randint(0, len(prev) - k + 1))
            shared = prev[a0:a0 + k]
            p = pos[chr_id] - 2 * flank + int(rng.randint(70, 3000))   # own exons follow the previous gene's last one
        else:
            p = pos[chr_id] + 2 * flank + int(rng.randint(0, 3000))
        exons = []
        for e in range(n_ex):
            if rng.rand() < small_exon_frac:
                ln = int(rng.randint(5, 31))
            else:
                ln = int(max(5, min(6000, rng.lognormal(np.log(130.0), 0.9))))
            exons.append((p, p + ln))
            p += ln + int(max(70, min(200000, rng.lognormal(np.log(1500.0), 1.2))))
        gene_start = shared[0][0] if shared else exons[0][0]
        gene_stop = exons[-1][1]
        pos[chr_id] = gene_stop + 2 * flank
        small = array('L')
        for (a, b) in shared:         # parts that exist already: this gene joins their segments
            flat = w.parts_to_segments[key_of(chr_id, a, b)]
            for i in range(0, len(flat), 3):
                sk = key_of(flat[i], flat[i + 1], flat[i + 2])
                w.segment_to_gene[sk].add(gene_id)
                if flat[i + 2] - flat[i + 1] <= small_thr:
                    small.extend([flat[i], flat[i + 1], flat[i + 2]])
        for (a, b) in exons:
            seq = _rand_seq(rng, b - a)
            w.part_seq[(chr_id, a, b)] = seq
            # canonical segments of the part: the whole exon, or 2-3 pieces at internal boundaries
            cuts = [a, b]
            if b - a >= 60 and rng.rand() < split_frac:
                k = int(rng.randint(1, 3))
                inner = sorted(set(int(v) for v in rng.randint(a + 25, b - 25 + 1, size=k)))
                inner = [v for i, v in enumerate(inner) if i == 0 or v - inner[i - 1] >= 25]
                cuts = [a] + inner + [b]
            flat = array('L')
            for s0, s1 in zip(cuts[:-1], cuts[1:]):
                flat.extend([chr_id, s0, s1])
                sk = key_of(chr_id, s0, s1)
                w.ref_segment_sequences[sk] = seq[s0 - a:s1 - a]
                w.segment_to_gene[sk] = {gene_id}
                if s1 - s0 <= small_thr:
                    small.extend([chr_id, s0, s1])
            w.parts_to_segments[key_of(chr_id, a, b)] = flat
        w.gene_to_small_segments[gene_id] = small
        exons = list(shared) + exons
        n_ex = len(exons)
        last_gene[chr_id] = exons
        up = (chr_id, max(0, gene_start - flank), gene_start - 1)
        down = (chr_id, gene_stop, gene_stop + flank)
        for f in (up, down):
            if f in w.part_seq:       # an overlapping gene that starts on the previous gene's first exon
                continue
            fs = _rand_seq(rng, f[2] - f[1])
            w.part_seq[f] = fs
            w.ref_flank_sequences[key_of(*f)] = fs
        n_tr = int(rng.randint(1, tr_max + 1))
        for t in range(n_tr):
            if t == 0 or n_ex <= 2:
                chosen = list(exons)
            else:
                keep = rng.rand(n_ex) < 0.8
                keep[0] = keep[-1] = True
                chosen = [e for e, kf in zip(exons, keep) if kf]
            w.transcripts.append((chr_id, gene_id, chosen, up, down))
    return w


def _mutate(rng, ref, err):
    """Returns (read, ref_pos_of_each_read_base or -1 for inserted/substituted bases)."""
    out, src = [], []
    i, n = 0, len(ref)
    while i < n:
        if rng.rand() < err:
            r = rng.rand()
            if r < 0.5:                      # deletion, geometric length
                dl = 1
                while rng.rand() >= 0.5:
                    dl += 1
                i += dl
                continue
            elif r < 0.8:                    # substitution (may re-draw the same base, as the reference's does)
                out.append("ACGT"[rng.randint(0, 4)]); src.append(-1)
            else:                            # insertion after the base
                out.append(ref[i]); src.append(i)
                out.append("ACGT"[rng.randint(0, 4)]); src.append(-1)
                while rng.rand() >= 0.5:
                    out.append("ACGT"[rng.randint(0, 4)]); src.append(-1)
        else:
            out.append(ref[i]); src.append(i)
        i += 1
    return "".join(out), src


def simulate_instance(w, rng, preset="ont_cdna", min_nam=15, flank_start_prob=0.15, flank_end_prob=0.1):
    """One (read, NAM chain) instance. Returns (read_seq, solution:list[Mem], chr_id)."""
    P = PRESETS[preset]
    chr_id, gene_id, exons, up, down = w.transcripts[int(rng.randint(0, len(w.transcripts)))]
    pieces = []  # (part tuple, local offset in part, length)
    if rng.rand() < flank_start_prob:
        ext = int(rng.randint(20, 400))
        flen = up[2] - up[1]
        ext = min(ext, flen)
        pieces.append((up, flen - ext, ext))
    for (a, b) in exons:
        pieces.append(((chr_id, a, b), 0, b - a))
    if rng.rand() < flank_end_prob:
        ext = min(int(rng.randint(20, 400)), down[2] - down[1])
        pieces.append((down, 0, ext))
    total = sum(p[2] for p in pieces)
    want = int(max(P["len_min"], rng.normal(P["len_mean"], P["len_sd"])))
    if want < total:
        # 3'-biased truncation for dRNA-like reads, uniform otherwise
        off = total - want if preset == "ont_drna" and rng.rand() < 0.7 else int(rng.randint(0, total - want + 1))
        lo, hi = off, off + want
        cut, acc = [], 0
        for part, o, ln in pieces:
            s, e = max(lo, acc), min(hi, acc + ln)
            if e > s:
                cut.append((part, o + (s - acc), e - s))
            acc += ln
        pieces = cut
    ref = "".join(w.part_seq[part][o:o + ln] for part, o, ln in pieces)
    read, src = _mutate(rng, ref, P["err"])
    if len(read) < 30:
        return simulate_instance(w, rng, preset, min_nam, flank_start_prob, flank_end_prob)
    # piece boundaries on the spliced reference
    bounds, acc = [], 0
    for part, o, ln in pieces:
        bounds.append((acc, acc + ln, part, o))
        acc += ln
    sol, run_start = [], None
    def flush(r0, r1):  # read interval [r0, r1] inclusive, all copied, consecutive on the reference
        g0 = src[r0]
        while r0 <= r1:
            for (b0, b1, part, o) in bounds:
                if b0 <= g0 < b1:
                    take = min(r1 - r0 + 1, b1 - g0)
                    if take >= min_nam:
                        x = part[1] + o + (g0 - b0)
                        sol.append((x, x + take - 1, r0, r0 + take - 1, take,
                                    "{0}^{1}^{2}".format(part[0], part[1], part[2])))
                    r0 += take; g0 += take
                    break
    for r in range(len(read) + 1):
        ok = r < len(read) and src[r] >= 0 and (run_start is None or src[r] == src[r - 1] + 1)
        if run_start is not None and not ok:
            flush(run_start, r - 1)
            run_start = None
        if run_start is None and r < len(read) and src[r] >= 0:
            run_start = r
    sol.sort(key=lambda v: v[1])
    solution = [Mem(x, y, c, d, val, j, pid) for j, (x, y, c, d, val, pid) in enumerate(sol)]
    if not solution:
        return simulate_instance(w, rng, preset, min_nam, flank_start_prob, flank_end_prob)
    return read, solution, chr_id


def make_instances(w, n, seed=1, preset="ont_cdna", **kw):
    rng = np.random.RandomState(seed)
    return [simulate_instance(w, rng, preset, **kw) for _ in range(n)]


# ---- native read simulator (csrc/synth.cpp -> libultra_synth.so): the same model at a million reads ---------------
import ctypes as _C
import os as _os

_SYNTH_LIB = None


class _SynthWorld(_C.Structure):
    _fields_ = [("n_tr", _C.c_int64), ("tr_off", _C.c_void_p), ("tr_piece", _C.c_void_p), ("tr_up", _C.c_void_p),
                ("tr_down", _C.c_void_p), ("piece_chr", _C.c_void_p), ("piece_start", _C.c_void_p),
                ("piece_stop", _C.c_void_p), ("piece_seq", _C.c_void_p), ("pool", _C.c_void_p)]


class _SynthParams(_C.Structure):
    _fields_ = [("seed", _C.c_uint64), ("n_reads", _C.c_int64), ("err", _C.c_double), ("len_model", _C.c_int32),
                ("len_a", _C.c_double), ("len_b", _C.c_double), ("len_min", _C.c_int32), ("len_max", _C.c_int32),
                ("bias3p", _C.c_double), ("flank_start_prob", _C.c_double), ("flank_end_prob", _C.c_double),
                ("min_nam", _C.c_int32), ("n_threads", _C.c_int32)]


class _PairParams(_C.Structure):
    _fields_ = [("seed", _C.c_uint64), ("n_pairs", _C.c_int64), ("q_model", _C.c_int32), ("q_a", _C.c_int32), ("q_b", _C.c_int32),
                ("q_min", _C.c_int32), ("q_max", _C.c_int32), ("q_sigma", _C.c_double), ("t_len", _C.c_int32),
                ("t_factor", _C.c_double), ("per_target", _C.c_int32), ("err", _C.c_double), ("planted", _C.c_double)]


# Myers micro-benchmarks of SURVEY.md section 8d: 50 % of the queries planted in their target at the config's error
# rate, 50 % unrelated decoys, k = (2 * min(m, n)) // 5, seeds 1-4.  M-flank's target is not given there: a 2.5 kb read.
MICRO = {
    "M-small": dict(seed=1, n_pairs=1 << 20, q_model=0, q_a=9, q_b=64, t_len=1000, per_target=64),
    "M-exon": dict(seed=2, n_pairs=1 << 20, q_model=1, q_a=130, q_sigma=0.9, q_min=9, q_max=2000, t_len=1000, per_target=64),
    "M-flank": dict(seed=3, n_pairs=1 << 16, q_model=2, q_a=1000, q_b=2000, t_len=2500, per_target=8),
    "M-contained": dict(seed=4, n_pairs=1 << 14, q_model=0, q_a=1000, q_b=10000, t_len=0, t_factor=1.25, per_target=1),
}


def make_micro_pairs(name, err=0.07, planted=0.5, n_pairs=None):
    """(arena, q_off, q_len, t_off, t_len, k) of one micro-benchmark, generated by libultra_synth.so."""
    spec = dict(MICRO[name])
    if n_pairs is not None:
        spec["n_pairs"] = int(n_pairs)
    prm = _PairParams()
    for f, _ in _PairParams._fields_:
        setattr(prm, f, spec.get(f, 0))
    prm.err, prm.planted = float(err), float(planted)
    lib = _synth_lib()
    h = lib.synth_pairs(_C.byref(prm))
    if not h:
        raise RuntimeError("synth_pairs failed")
    try:
        n, nb = _C.c_int64(), _C.c_int64()
        lib.synth_pairs_sizes(h, _C.byref(n), _C.byref(nb))
        arena = np.empty(max(nb.value, 1), np.uint8)
        q_off, t_off = np.empty(n.value, np.int64), np.empty(n.value, np.int64)
        q_len, t_len, k = np.empty(n.value, np.int32), np.empty(n.value, np.int32), np.empty(n.value, np.int32)
        lib.synth_pairs_copy(h, *[a.ctypes.data_as(_C.c_void_p) for a in (arena, q_off, q_len, t_off, t_len, k)])
    finally:
        lib.synth_pairs_free(h)
    return arena[:nb.value], q_off, q_len, t_off, t_len, k


def _synth_lib():
    global _SYNTH_LIB
    if _SYNTH_LIB is None:
        path = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "libultra_synth.so")
        if not _os.path.exists(path):
            raise OSError("libultra_synth.so not found - build it with `make -C ultra_amd/csrc`")
        lib = _C.CDLL(path)
        lib.synth_generate.restype = _C.c_void_p
        lib.synth_generate.argtypes = [_C.POINTER(_SynthWorld), _C.POINTER(_SynthParams)]
        lib.synth_sizes.argtypes = [_C.c_void_p] + [_C.POINTER(_C.c_int64)] * 3
        lib.synth_sizes.restype = None
        lib.synth_copy.argtypes = [_C.c_void_p] + [_C.c_void_p] * 9
        lib.synth_copy.restype = None
        lib.synth_free.argtypes = [_C.c_void_p]
        lib.synth_free.restype = None
        lib.synth_pairs.restype = _C.c_void_p
        lib.synth_pairs.argtypes = [_C.POINTER(_PairParams)]
        lib.synth_pairs_sizes.argtypes = [_C.c_void_p, _C.POINTER(_C.c_int64), _C.POINTER(_C.c_int64)]
        lib.synth_pairs_sizes.restype = None
        lib.synth_pairs_copy.argtypes = [_C.c_void_p] * 7
        lib.synth_pairs_copy.restype = None
        lib.synth_pairs_free.argtypes = [_C.c_void_p]
        lib.synth_pairs_free.restype = None
        lib.synth_seed_text.argtypes = [_C.c_int64] + [_C.c_void_p] * 6 + [_C.c_int64, _C.c_void_p]
        lib.synth_seed_text.restype = _C.c_int64
        _SYNTH_LIB = lib
    return _SYNTH_LIB


class FlatWorld(object):
    """Transcripts and part sequences of a World as flat arrays (what the native simulator reads)."""

    def __init__(self, w):
        ids, chunks, size = {}, [], 0
        chrs, starts, stops, seqoff = [], [], [], []
        for part, seq in w.part_seq.items():
            ids[part] = len(chrs)
            chrs.append(part[0]); starts.append(part[1]); stops.append(part[2]); seqoff.append(size)
            b = seq.encode()
            chunks.append(b); size += len(b)
        tr_off, tr_piece, tr_up, tr_down = [0], [], [], []
        for chr_id, _, exons, up, down in w.transcripts:
            tr_piece.extend(ids[(chr_id, a, b)] for a, b in exons)
            tr_off.append(len(tr_piece)); tr_up.append(ids[up]); tr_down.append(ids[down])
        a64 = lambda v: np.ascontiguousarray(v, np.int64)
        self.tr_off, self.tr_piece, self.tr_up, self.tr_down = a64(tr_off), a64(tr_piece), a64(tr_up), a64(tr_down)
        self.piece_chr, self.piece_start, self.piece_stop, self.piece_seq = a64(chrs), a64(starts), a64(stops), a64(seqoff)
        self.pool = np.frombuffer(b"".join(chunks) or b"\0", np.uint8)
        self.n_tr = len(w.transcripts)

    def c_struct(self):
        ptr = lambda a: a.ctypes.data_as(_C.c_void_p)
        return _SynthWorld(self.n_tr, ptr(self.tr_off), ptr(self.tr_piece), ptr(self.tr_up), ptr(self.tr_down), ptr(self.piece_chr),
                           ptr(self.piece_start), ptr(self.piece_stop), ptr(self.piece_seq), ptr(self.pool))


class ReadSet(object):
    """n reads with their NAM chains as flat arrays: exactly the fields of ultra_plan_request (include/ultra_hip.h)
    plus the read coordinates c, d of every NAM and the chromosome of every read."""
    __slots__ = ("reads", "read_off", "nam_off", "nam_part", "nam_x", "nam_y", "nam_c", "nam_d", "chr")

    def __len__(self):
        return len(self.read_off) - 1

    def slice(self, a, b):
        """Reads [a, b) as a ReadSet of their own (offsets rebased; arrays are views where possible)."""
        r = ReadSet()
        r0, r1, n0, n1 = int(self.read_off[a]), int(self.read_off[b]), int(self.nam_off[a]), int(self.nam_off[b])
        r.reads = self.reads[r0:r1]
        r.read_off = self.read_off[a:b + 1] - r0
        r.nam_off = self.nam_off[a:b + 1] - n0
        r.nam_part = self.nam_part[3 * n0:3 * n1]
        r.nam_x, r.nam_y, r.nam_c, r.nam_d = self.nam_x[n0:n1], self.nam_y[n0:n1], self.nam_c[n0:n1], self.nam_d[n0:n1]
        r.chr = self.chr[a:b]
        return r

    def instance(self, i):
        """(read_seq, solution, chr_id) of read i in the form simulate_instance returns (Python objects, for per-read checks)."""
        r0, r1, n0, n1 = int(self.read_off[i]), int(self.read_off[i + 1]), int(self.nam_off[i]), int(self.nam_off[i + 1])
        sol = []
        for j, u in enumerate(range(n0, n1)):
            pid = "{0}^{1}^{2}".format(int(self.nam_part[3 * u]), int(self.nam_part[3 * u + 1]), int(self.nam_part[3 * u + 2]))
            x, y, c, d = int(self.nam_x[u]), int(self.nam_y[u]), int(self.nam_c[u]), int(self.nam_d[u])
            sol.append(Mem(x, y, c, d, y - x + 1, j, pid))
        return self.reads[r0:r1].tobytes().decode(), sol, int(self.chr[i])


def seed_text(rs):
    """(text uint8, line_off int64[n + 1]): the hit lines a seeder would write for the NAM chains of ReadSet rs - read r's
    forward record is text[line_off[r]:line_off[r + 1]] (the format of seed_wrapper.read_seeds)."""
    n = len(rs)
    cap = int(rs.nam_off[-1]) * 100 + 16
    text = np.empty(cap, np.uint8)
    line_off = np.zeros(n + 1, np.int64)
    a = [np.ascontiguousarray(v, np.int64) for v in (rs.nam_off, rs.nam_part, rs.nam_x, rs.nam_y, rs.nam_c)]
    got = _synth_lib().synth_seed_text(n, *[v.ctypes.data_as(_C.c_void_p) for v in a], text.ctypes.data_as(_C.c_void_p), cap,
                                       line_off.ctypes.data_as(_C.c_void_p))
    if got < 0:
        raise RuntimeError("synth_seed_text: buffer too small")
    return text[:max(got, 1)], line_off


def make_reads_native(flat, n, seed=1, preset="ont_cdna", min_nam=15, flank_start_prob=0.15, flank_end_prob=0.1, threads=0):
    """n simulated reads + NAM chains as a ReadSet, generated by libultra_synth.so (deterministic in seed and n; a
    different random stream than make_instances).  flat: FlatWorld."""
    P = PRESETS[preset]
    lib = _synth_lib()
    prm = _SynthParams()
    prm.seed, prm.n_reads, prm.err = int(seed), int(n), float(P["err"])
    if "len_median" in P:
        prm.len_model, prm.len_a, prm.len_b = 1, float(P["len_median"]), float(P["len_sigma"])
        prm.len_max, prm.bias3p = int(P["len_max"]), float(P["bias3p"])
    else:
        prm.len_model, prm.len_a, prm.len_b, prm.len_max, prm.bias3p = 0, float(P["len_mean"]), float(P["len_sd"]), 0, 0.0
    prm.len_min = int(P["len_min"])
    prm.flank_start_prob, prm.flank_end_prob, prm.min_nam = float(flank_start_prob), float(flank_end_prob), int(min_nam)
    prm.n_threads = int(threads) if threads else max(1, min(16, len(_os.sched_getaffinity(0))))
    cw = flat.c_struct()
    h = lib.synth_generate(_C.byref(cw), _C.byref(prm))
    if not h:
        raise RuntimeError("synth_generate failed")
    try:
        nr, nb, nn = _C.c_int64(), _C.c_int64(), _C.c_int64()
        lib.synth_sizes(h, _C.byref(nr), _C.byref(nb), _C.byref(nn))
        rs = ReadSet()
        rs.reads = np.empty(max(nb.value, 1), np.uint8)
        rs.read_off = np.empty(nr.value + 1, np.int64); rs.nam_off = np.empty(nr.value + 1, np.int64)
        rs.nam_part = np.empty(3 * nn.value, np.int64)
        rs.nam_x, rs.nam_y = np.empty(nn.value, np.int64), np.empty(nn.value, np.int64)
        rs.nam_c, rs.nam_d = np.empty(nn.value, np.int64), np.empty(nn.value, np.int64)
        rs.chr = np.empty(nr.value, np.int32)
        lib.synth_copy(h, *[a.ctypes.data_as(_C.c_void_p) for a in (rs.reads, rs.read_off, rs.nam_off, rs.nam_part, rs.nam_x,
                                                                  rs.nam_y, rs.nam_c, rs.nam_d, rs.chr)])
        rs.reads = rs.reads[:nb.value]
    finally:
        lib.synth_free(h)
    return rs
