#!/usr/bin/env python3
"""bench.py - throughput of the MAM stage (uLTRA's per-read alignment core) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads B] [--preset ont_cdna]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of B synthetic (read, NAM chain) instances
whose candidate (segment, read) pairs are already resident in HBM: Myers forward / start / traceback
kernels, MAM materialisation, chaining.  Workload = BASELINE.json configs[2] in its self-generated
synthetic form (ONT cDNA-like reads, ~7 % errors, ~1 kb): GRCh38/GENCODE do not
exist offline.  Reads are independent, so with N > 1 every rank runs its own shard (weak scaling, no
data-path collective; torch.distributed is used for the barrier and the max-over-ranks time only).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the Myers forward pass (one kernel template, launched once per register bucket, the launches of
                a step running concurrently): integer-VALU bound, achieved = 0.75 int32-op/cell * forward cells
                of the step / HIP-event time from the first launch's start to the last one's end;
                `largest_instantiation` = the same for the single biggest launch (shares the chip meanwhile)
  cpu_baseline  the CPU oracle (oracle/, "port") timed on a bounded sample on the host cores
"""
import argparse
import json
import os
import sys
import time

# Eight hardware queues instead of HIP's default four: the launches of one pass (five forward buckets, seven
# traceback kernels) then all run side by side.  Must be in the environment before the first HIP call.
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("ULTRA_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

OPS_PER_CELL = 0.75          # SURVEY.md section 8d convention: 48 int32 VALU ops per 64-cell block column
PEAK_CLOCK_GHZ = 2.4         # MI355X_MICROARCH.md: max clock
LANES_PER_CU_CLK = 128       # 4 SIMD-32 per CU (MI355X_MICROARCH.md "CU = 4 x SIMD-32")
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=262144, help="reads (instances) per step and per GPU")
    ap.add_argument("--preset", default="ont_cdna", choices=["ont_cdna", "isoseq", "ont_drna"])
    ap.add_argument("--genes", type=int, default=1500)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU-baseline sample budget (0 = skip)")
    ap.add_argument("--cpu-workers", type=int, default=0, help="0 = min(16, available cores)")
    ap.add_argument("--contexts", type=int, default=1, help="independent library contexts (host threads) per GPU; "
                    "each runs 1/N of the step's reads so that the passes of different sub-batches overlap")
    ap.add_argument("--stagger-ms", type=float, default=4.0, help="start offset between the contexts' loops (--contexts > 1)")
    ap.add_argument("--lanes", type=int, default=0, help="launch lanes (HIP streams) per context, 0 = library default (4)")
    ap.add_argument("--slots", type=int, default=0, help="end positions kept per pair in the first pass, 0 = library default")
    ap.add_argument("--planner", default="native", choices=["native", "python"],
                    help="host-side candidate enumeration: packed index + csrc/plan.hip, or the Python port")
    ap.add_argument("--check", type=int, default=64, help="instances of the last step verified against the oracle")
    return ap.parse_args()


# ---- CPU baseline (oracle; runs BEFORE anything touches the GPU so that fork is safe) -----------------
_CPU = {}


def _cpu_worker(rng):
    from oracle import classify_port
    idx, insts, min_acc = _CPU["idx"], _CPU["insts"], _CPU["min_acc"]
    t0 = time.perf_counter()
    for i in rng:
        read, sol, _ = insts[i]
        classify_port.classify(sol, *idx, read, None, min_acc)
    return time.perf_counter() - t0


def cpu_baseline(idx, insts, min_acc, seconds, workers, cells_per_read):
    import multiprocessing as mp
    from oracle import cport
    cport.lib()
    _CPU.update(idx=idx, insts=insts, min_acc=min_acc)
    # calibrate on a few instances, then size the sample to `seconds` of wall time
    t = _cpu_worker(range(min(32, len(insts))))
    per = t / min(32, len(insts))
    n = int(max(workers * 8, seconds / per * workers))   # wraps around the batch when it is smaller than the sample
    chunks = [[i % len(insts) for i in range(j, n, workers)] for j in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        pool.map(_cpu_worker, chunks)
    wall = time.perf_counter() - t0
    return {"value": n / wall, "unit": "reads/s", "cores": workers, "kind": "port",
            "sample": "%d reads of the same synthetic batch through oracle/classify_port.py + liboracle.so "
                      "(C Myers bit-vector + chainers), %d fork workers, %.1f s wall" % (n, workers, wall),
            "gcups": n * cells_per_read / wall * 1e-9}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    from ultra_amd import synth
    from ultra_amd import classify_read_with_mams as stage

    # ---- synthetic workload (same world on every rank, a different read shard per rank) ----------
    P = synth.PRESETS[args.preset]
    w = synth.make_world(seed=20241008, n_genes=args.genes, n_chr=8)
    idx = w.index()
    insts = synth.make_instances(w, args.reads, seed=1000 + rank, preset=args.preset)
    import torch  # noqa: F401  (before libultra_hip.so is loaded: both must end up on the same HIP runtime - torch's)
    t_plan = time.perf_counter()
    if args.planner == "native" and args.contexts <= 1:
        from ultra_amd import index as native_index
        nix = native_index.NativeIndex(*idx)
        t_plan = time.perf_counter()
        batch = nix.plan([(sol, read) for read, sol, _ in insts])
    else:
        batch = stage.prepare_batch([(sol, read) for read, sol, _ in insts], *idx)
    plan_s = time.perf_counter() - t_plan
    cells = batch.cells
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        workers = args.cpu_workers or max(1, min(16, len(os.sched_getaffinity(0))))
        cpu = cpu_baseline(idx, insts, P["min_acc"], args.cpu_seconds, workers, cells / float(len(insts)))

    # ---- GPU ---------------------------------------------------------------------------------------
    import torch
    import torch.distributed as dist
    from ultra_amd import lib
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    import threading
    NC = max(1, args.contexts)
    cfg = {}
    if args.lanes:
        cfg["launch_lanes"] = args.lanes
    if args.slots:
        cfg["loc_slots"] = args.slots
    ctxs = [lib.Context(local, cfg=cfg or None) for _ in range(NC)]
    ctx = ctxs[0]
    info = ctx.device_info()
    if NC == 1:
        batches = [batch]
    else:
        batches = [stage.pack_plans(batch.plans[i::NC]) for i in range(NC)]
    for c, b in zip(ctxs, batches):
        stage.stage_batch(c, b)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_all():
        if NC == 1:
            stage.run_staged(ctx, P["min_acc"])
            return
        errs = []

        def work(c):
            try:
                stage.run_staged(c, P["min_acc"])
            except Exception as e:   # surfaced below
                errs.append(e)
        th = [threading.Thread(target=work, args=(c,)) for c in ctxs]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    for _ in range(args.warmup):
        run_all()
    class_ms = {}
    fwd_ms = rev_ms = tb_ms = all_ms = fwdk_ms = 0.0
    build_ms = chain_ms = gather_ms = 0.0
    barrier()
    t0 = time.perf_counter()

    def collect(c):
        nonlocal fwd_ms, rev_ms, tb_ms, all_ms, fwdk_ms, build_ms, chain_ms, gather_ms
        _, ms = c.myers_timing()
        fwd_ms += ms[0]; rev_ms += ms[1]; tb_ms += ms[2]; all_ms += ms[3]; fwdk_ms += ms[4]
        mm = c.mam_timing()
        build_ms += mm[0]; chain_ms += mm[1]; gather_ms += mm[2]
        for key, (cc, t) in c.myers_class_timing().items():
            a = class_ms.setdefault(key, [0.0, 0.0, 0])
            a[0] += cc; a[1] += t; a[2] += 1

    if NC == 1:
        for _ in range(args.steps):
            run_all()
            collect(ctx)
    else:
        # Sub-batches in flight on one GPU: every context runs its K steps back to back from its own host thread,
        # started half a pass apart, so that the latency-bound passes of one (traceback, MAM build, chaining)
        # overlap with the throughput-bound ones (forward, starts) of the other.
        errs, lock = [], threading.Lock()

        def loop(i, c):
            try:
                time.sleep(i * args.stagger_ms * 1e-3)
                for _ in range(args.steps):
                    stage.run_staged(c, P["min_acc"])
                    with lock:
                        collect(c)
            except Exception as e:   # surfaced below
                errs.append(e)
        th = [threading.Thread(target=loop, args=(i, c)) for i, c in enumerate(ctxs)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
    barrier()
    elapsed = time.perf_counter() - t0
    # results back to the host (PCIe; outside the timed region, reported separately)
    t1 = time.perf_counter()
    ress = [c.mam_fetch() for c in ctxs]
    fetch_s = time.perf_counter() - t1
    t1 = time.perf_counter()
    outs = [stage.results_to_python(b, r) for b, r in zip(batches, ress)]
    topy_s = time.perf_counter() - t1
    out = [None] * len(insts)
    for i in range(NC):
        out[i::NC] = outs[i]
    res = {"chain_off": [sum(int(r["chain_off"][-1]) for r in ress)]}
    from ultra_amd import shard
    elapsed, reads_all, cells_all = shard.reduce_job(elapsed, len(insts), cells, device="cuda" if world > 1 else None)

    # ---- parity spot check of the timed output (oracle = checker only) ---------------------------
    checked = 0
    if rank == 0 and args.check > 0:
        import struct
        from oracle import classify_port
        for i in range(0, len(insts), max(1, len(insts) // args.check)):
            read, sol, _ = insts[i]
            want = classify_port.classify(sol, *idx, read, None, P["min_acc"])
            g = out[i]
            ok = (want[0] == g[0] and struct.pack("d", float(want[1])) == struct.pack("d", float(g[1]))
                  and tuple(tuple(m) for m in want[2]) == tuple(tuple(m) for m in g[2]))
            if not ok:
                raise SystemExit("bench.py: instance %d differs from the oracle - refusing to report a number" % i)
            checked += 1

    if rank == 0:
        K = args.steps
        peak_ops = info["n_cu"] * LANES_PER_CU_CLK * PEAK_CLOCK_GHZ * 1e9
        dom = max(class_ms.items(), key=lambda kv: kv[1][1]) if class_ms else None
        roof = None
        if dom:
            (G, Wd), (c, ms, launches) = dom
            ops_per_launch = OPS_PER_CELL * c / launches
            dur = ms / launches * 1e-3
            achieved = ops_per_launch / dur * 1e-12
            kname = "myers_mega<0, %d>" % G   # forward pass (MODE 0) of register bucket G
            # HBM bytes from the committed rocprofv3 --pmc passes of this same command (profiles/)
            traffic = fam_traffic = None
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc):
                try:
                    rec = json.load(open(pmc))
                    if rec.get("dominant_kernel") == kname:
                        traffic = rec.get("hbm_bytes_per_launch")
                    fam_traffic = rec.get("forward_family", {}).get("hbm_bytes_per_step")
                except Exception:
                    traffic = fam_traffic = None
            # The forward pass is ONE kernel template instantiated per register bucket; the (up to 5) instantiations
            # of a step are launched together on their own streams and share the CUs, so the duration of a single one says
            # how long it shared the chip, not how fast it is.  The roofline is therefore taken over the family:
            # all forward cells of the step / (first launch start -> last launch end, HIP events on those streams).
            # The largest single instantiation, whose avg duration is the one rocprofv3 --stats lists, is kept below.
            fam_ms = fwdk_ms / K / max(1, args.contexts)
            fam_ach = OPS_PER_CELL * (cells_all / world) / (fam_ms * 1e-3) * 1e-12 if fam_ms > 0 else 0.0
            roof = {"bound": "valu_int32", "kernel": "myers_mega<0, 0..4> (forward pass: concurrent launches, one per register bucket)",
                    "achieved": fam_ach, "peak": peak_ops * 1e-12, "unit": "Tops/s", "frac": fam_ach / (peak_ops * 1e-12),
                    "traffic": fam_traffic, "cells_per_launch": cells_all / world, "avg_launch_ms": fam_ms,
                    "ops_per_cell": OPS_PER_CELL,
                    # profiles/r01_valu_issue_rate.txt: a SIMD issues one plain 32-bit VALU instruction per 2.44 cycles of
                    # the nominal 2.4 GHz under this kind of load (the chip down-clocks), three-source ALU ops
                    # (v_alignbit / v_lshl_or / v_add3) and 64-bit adds per 4.2
                    "peak_sustained_measured": peak_ops * 1e-12 * 2.0 / 2.44,
                    "frac_of_sustained": fam_ach / (peak_ops * 1e-12 * 2.0 / 2.44),
                    "largest_instantiation": {"kernel": kname, "achieved": achieved, "frac": achieved / (peak_ops * 1e-12),
                                              "traffic": traffic, "cells_per_launch": c / launches, "avg_launch_ms": ms / launches,
                                              "note": "runs concurrently with the other buckets' launches"},
                    "hbm": {"algorithmic_bytes_per_step": batch.algorithmic_bytes, "achieved_GBs":
                            batch.algorithmic_bytes / (all_ms / K * 1e-3) * 1e-9 if all_ms > 0 else None,
                            "peak_GBs": HBM_PEAK_GBS}}
        line = {
            "metric": "reads aligned/sec (whole node), MAM stage: Myers HW matching + MAM chaining",
            "value": reads_all * K / elapsed, "unit": "reads/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 bit-vectors / int32 scores (Myers), f64 (chaining)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2] synthetic form: ONT-cDNA-like reads (%s, err %.2f, ~%d bp), "
                                   "%d reads/GPU/step, %d genes synthetic annotation" % (args.preset, P["err"], P["len_mean"], args.reads, args.genes),
                       "reads_per_gpu_per_step": args.reads, "contexts_per_gpu": args.contexts, "pairs_per_step_per_gpu": batch.n_pairs,
                       "cells_per_step_per_gpu": cells, "parallelism": "reads sharded, %d rank(s), no collective" % world},
            "myers_gcups_forward": cells_all * K / (fwd_ms * 1e-3) * 1e-9 / 1.0 if fwd_ms > 0 else None,
            "myers_gcups_all_passes": cells_all * K / (all_ms * 1e-3) * 1e-9 if all_ms > 0 else None,
            "stage_ms_per_step": {"myers_forward": fwd_ms / K, "myers_starts": rev_ms / K, "myers_traceback": tb_ms / K,
                                  "myers_total": all_ms / K, "mam_build": build_ms / K, "chain": chain_ms / K,
                                  "chain_gather": gather_ms / K},
            "outside_timed_region": {"host_plan_us_per_read": plan_s / len(insts) * 1e6, "host_planner": args.planner if args.contexts <= 1 else "python",
                                     "fetch_results_pcie_ms": fetch_s * 1e3, "python_tuples_ms": topy_s * 1e3,
                                     "reads_per_s_incl_fetch": reads_all / world / (elapsed / K + fetch_s),
                                     "chain_rows_fetched": int(res["chain_off"][-1])},
            "forward_buckets": {"bucket%d_wmax%d" % k: {"cells_per_step": v[0] / v[2], "ms_per_step": v[1] / v[2],
                                                        "tcups": v[0] / v[1] * 1e-9 if v[1] > 0 else None}
                                for k, v in sorted(class_ms.items())},
            "device": info, "parity_checked_instances": checked,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
