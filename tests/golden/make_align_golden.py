"""Generates tests/golden/align_golden.json by RUNNING THE REFERENCE'S OWN seed parsing, instance splitting /
upper bound and NAM chaining in the build container:

    modules/seed_wrapper.py:139   read_seeds(seeds)                       (gz seed file -> hit lines)
    modules/align.py:357          get_mems_from_input(hits)
    modules/align.py:112          annotate_guaranteed_optimal_bound(mems, is_rc, max_intron_chr, max_global_intron)
    modules/colinear_solver.py    read_coverage / n_logn_read_coverage     (as switched at align.py:448)
    modules/help_functions.py:93  reverse_complement

    python tests/golden/make_align_golden.py           (needs /root/reference)

modules/align.py imports gffutils, pysam, edlib and parasail, which this image lacks and which none of the
functions above touches: they are bound to empty modules so that the file imports.  The location selection loop of
align.py:440-475 sits inside align_single() between queue and file handling and cannot be called on its own; the
generator drives the reference's functions through that loop's statements and records, besides the final selection,
every intermediate (mems, bounds, per-instance chainings) so a test can tell which step a difference comes from.

A fixture is data only: the seed file text that went in, the parameters, and what came out.
"""
import contextlib
import gzip
import io
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
for name in ("gffutils", "pysam", "edlib", "parasail"):
    sys.modules.setdefault(name, types.ModuleType(name))

from modules import align as ref_align            # noqa: E402
from modules import colinear_solver as ref_cs     # noqa: E402
from modules import help_functions as ref_help    # noqa: E402
from modules import seed_wrapper as ref_seeds     # noqa: E402
from ultra_amd import seed_wrapper, synth         # noqa: E402


def mem_rows(mems):
    return [[m.x, m.y, m.c, m.d, m.val, m.j, m.exon_part_id] for m in mems]


def build_seed_text(world, rng, n_reads):
    """True NAMs of simulated reads + decoy seeds on other parts / chromosomes / the reverse strand, in the
    namfinder format.  Every third read gets a dense block of short seeds so that an instance passes 90 NAMs."""
    parts = list(world.part_seq.keys())
    records = []
    for r in range(n_reads):
        read, solution, chr_id = synth.simulate_instance(world, rng, "ont_cdna", min_nam=12)
        fw = list(solution)
        rc = []

        def decoy(n, lo, hi):
            out = []
            for _ in range(n):
                part = parts[int(rng.randint(0, len(parts)))]
                plen = part[2] - part[1]
                ln = int(min(rng.randint(lo, hi), plen, len(read)))
                off = int(rng.randint(0, plen - ln + 1))
                c = int(rng.randint(0, len(read) - ln + 1))
                x = part[1] + off
                out.append(synth.Mem(x, x + ln - 1, c, c + ln - 1, ln, 0, "{0}^{1}^{2}".format(*part)))
            return out
        fw += decoy(int(rng.randint(0, 12)), 10, 40)
        rc += decoy(int(rng.randint(0, 8)), 10, 40)
        if r % 3 == 0:          # repeat-like pile of short seeds on the true parts
            for m in list(solution):
                for _ in range(int(rng.randint(2, 9))):
                    ln = int(rng.randint(8, max(9, min(30, m.val))))
                    s = int(rng.randint(0, m.val - ln + 1)) if m.val > ln else 0
                    shift = int(rng.randint(-3, 4))
                    c = min(max(0, m.c + s + shift), len(read) - ln)
                    fw.append(synth.Mem(m.x + s, m.x + s + ln - 1, c, c + ln - 1, ln, 0, m.exon_part_id))
        order = rng.permutation(len(fw))
        records.append(("read_{0}".format(r), [fw[i] for i in order], rc, read))
    return records


def main():
    world = synth.make_world(seed=4242, n_genes=60, n_chr=3)
    rng = np.random.RandomState(99)
    records = build_seed_text(world, rng, 48)
    # one read without forward hits, one without any hit: edge cases of read_seeds / the chaining loop
    records[5] = (records[5][0], [], records[5][2], records[5][3])
    records[7] = (records[7][0], [], [], records[7][3])
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "seeds.txt.gz")
    seed_wrapper.write_seeds(path, [(a, f, r) for a, f, r, _ in records])
    text = gzip.open(path, "rb").read().decode()
    chr_ids = sorted({p[0] for p in world.part_seq})
    max_intron_chr = {c: int(rng.randint(500, 4000)) for c in chr_ids}
    params = [dict(max_global_intron=1200000, dropoff=0.95, max_loc=5),
              dict(max_global_intron=3000, dropoff=0.6, max_loc=2)]
    out = {"seed_text": text, "max_intron_chr": {str(k): v for k, v in max_intron_chr.items()},
           "reads": [rec[3] for rec in records], "runs": []}
    with contextlib.redirect_stdout(io.StringIO()):
        parsed = list(ref_seeds.read_seeds(path))
    out["parsed"] = [[acc, hits, acc_rc, hits_rc] for acc, hits, acc_rc, hits_rc in parsed]
    out["revcomp"] = [ref_help.reverse_complement(rec[3][:60]) for rec in records[:8]] + \
                     [ref_help.reverse_complement("ACGTUNacgtunRYKMSWBVDHXrykmswbvdh")]
    n_tree = 0
    for p in params:
        run = {"params": p, "reads": []}
        for acc, hits, acc_rc, hits_rc in parsed:
            mems = ref_align.get_mems_from_input(hits)
            mems_rc = ref_align.get_mems_from_input(hits_rc)
            ub = ref_align.annotate_guaranteed_optimal_bound(mems, False, max_intron_chr, p["max_global_intron"])
            ub_rc = ref_align.annotate_guaranteed_optimal_bound(mems_rc, True, max_intron_chr, p["max_global_intron"])
            rec = {"mems": {str(k): mem_rows(v) for k, v in mems.items()},
                   "mems_rc": {str(k): mem_rows(v) for k, v in mems_rc.items()},
                   "bounds": [[k[0], k[1], v[0], v[1], mem_rows(v[2])] for k, v in list(ub.items()) + list(ub_rc.items())]}
            # ---- statements of align.py:440-475 ----
            all_chainings = []
            best_solution_value = 0
            for (chr_id, chr_instance_index), (upper_bound_cov, is_rc, all_mems_to_chromosome) in sorted(
                    list(ub.items()) + list(ub_rc.items()), key=lambda x: x[1][0], reverse=True):
                if upper_bound_cov < best_solution_value * p["dropoff"]:
                    break
                max_allowed_intron = min(max_intron_chr[chr_id] + 20000, p["max_global_intron"])
                if len(all_mems_to_chromosome) < 90:
                    solutions, mem_solution_value = ref_cs.read_coverage(all_mems_to_chromosome, max_allowed_intron)
                else:
                    solutions, mem_solution_value = ref_cs.n_logn_read_coverage(all_mems_to_chromosome)
                    n_tree += 1
                if mem_solution_value > best_solution_value:
                    best_solution_value = mem_solution_value
                for sol in solutions:
                    all_chainings.append((chr_id, sol, mem_solution_value, is_rc))
            selected = []
            if all_chainings:
                all_chainings = sorted(all_chainings, key=lambda x: (-x[2], (x[1][-1].y - x[1][0].x)))
                best_chaining_score = all_chainings[0][2]
                for i_nr_sol, (chr_id, mem_solution, chaining_score, is_rc) in enumerate(all_chainings):
                    if chaining_score / float(best_chaining_score) < p["dropoff"] or i_nr_sol >= p["max_loc"]:
                        break
                    selected.append([chr_id, mem_rows(mem_solution), chaining_score, is_rc])
            rec["selected"] = selected
            run["reads"].append(rec)
        out["runs"].append(run)
    assert n_tree > 5, n_tree
    dst = os.path.join(HERE, "align_golden.json")
    json.dump(out, open(dst, "w"), separators=(",", ":"))
    print("wrote", dst, os.path.getsize(dst), "bytes; n log n instances:", n_tree)


if __name__ == "__main__":
    main()
