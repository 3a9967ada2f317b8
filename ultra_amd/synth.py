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
    # name: (error rate, read length mean, sd, min, lognormal tail?, min_acc)
    "isoseq": dict(err=0.01, len_mean=2500, len_sd=800, len_min=300, min_acc=0.8),
    "ont_cdna": dict(err=0.07, len_mean=1000, len_sd=400, len_min=200, min_acc=0.6),
    "ont_drna": dict(err=0.12, len_mean=1100, len_sd=900, len_min=200, min_acc=0.6),
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


def _rand_seq(rng, n):
    return "".join("ACGT"[i] for i in rng.randint(0, 4, size=n))


def make_world(seed=20241008, n_genes=200, n_chr=2, flank=1000, small_thr=200, max_exons=30,
               small_exon_frac=0.03, split_frac=0.3):
    rng = np.random.RandomState(seed)
    w = World()
    pos = [5000] * n_chr
    for g in range(n_genes):
        chr_id = g % n_chr
        gene_id = "gene%05d" % g
        n_ex = int(min(max_exons, max(1, rng.geometric(1.0 / 9.0))))
        p = pos[chr_id] + 2 * flank + int(rng.randint(0, 3000))
        gene_start = p
        exons = []
        for e in range(n_ex):
            if rng.rand() < small_exon_frac:
                ln = int(rng.randint(5, 31))
            else:
                ln = int(max(5, min(6000, rng.lognormal(np.log(130.0), 0.9))))
            exons.append((p, p + ln))
            p += ln + int(max(70, min(200000, rng.lognormal(np.log(1500.0), 1.2))))
        gene_stop = exons[-1][1]
        pos[chr_id] = gene_stop + 2 * flank
        small = array('L')
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
        up = (chr_id, max(0, gene_start - flank), gene_start - 1)
        down = (chr_id, gene_stop, gene_stop + flank)
        for f in (up, down):
            fs = _rand_seq(rng, f[2] - f[1])
            w.part_seq[f] = fs
            w.ref_flank_sequences[key_of(*f)] = fs
        n_tr = int(rng.randint(1, 5))
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
