#!/usr/bin/env python3
"""bench.py - throughput of the MAM stage (uLTRA's per-read alignment core) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[2] in its self-generated synthetic form (GRCh38 / GENCODE do
not exist offline): a GENCODE-shaped annotation of ~60 000 genes on 24 chromosomes with overlapping genes
(ultra_amd/synth.py WORLDS["c3"]) and 1 000 000 DISTINCT simulated ONT-cDNA reads per GPU (7 % errors, N(1000, 400),
the reference simulator's error mix) as 4 batches of 262 144, each with its NAM chain.

A "step" is one pass of the WHOLE hot path over one batch whose inputs (the reads and their NAM chains, the packed
index) are already resident in HBM: candidate enumeration on the device (ultra_mam_plan_resident: row a4 of SURVEY.md
section 8 - classify_read_with_mams.py:180-300, :441-523), pair sort and work lists, then Myers forward / start /
traceback kernels, MAM materialisation and chaining (ultra_mam_run).  Every batch is uploaded into a context of its
own before the timed region (ultra_mam_upload_nams); the timed steps cycle through them, so no batch is replayed back
to back.  `kernels_only` is the same without the enumeration (round 2's `value`).  Reads are independent: with N > 1 the job is N x 4 batches dealt
round-robin over the ranks (ultra_amd/shard.py), weak scaling, no data-path collective; torch.distributed only
provides the barrier and the max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      the Myers forward pass (one kernel template, launched once per register bucket, the launches of a step
                running concurrently): integer-VALU bound, achieved = 0.75 int32-op/cell * forward cells of the step /
                HIP-event time from the first launch's start to the last one's end
  cpu_baseline  the CPU oracle (oracle/, "port") on a bounded sample of the same reads: all cores (pool capped at the
                box's CPU share of 16) as `value`, one thread beside it
  end_to_end    NOT `value`: reads/s from host arrays (NAM chains + reads) to fetched results through ultra_pipe
                (H2D, device plan, sort, kernels, D2H; `depth` batches in flight), plus what a synchronous caller sees
  micro         the four kernel-only Myers micro-benchmarks of SURVEY.md section 8d with per-class TCUPS / fraction
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ultra_amd import hwq  # noqa: E402

# One hardware queue per launch lane (the launches of a pass run side by side); must be in the environment before the
# first HIP call.  --hw-queues / ULTRA_HW_QUEUES=4 gives the line a caller gets that initialised HIP with its defaults.
for i, a in enumerate(sys.argv):
    if a == "--hw-queues" and i + 1 < len(sys.argv):
        os.environ["ULTRA_HW_QUEUES"] = sys.argv[i + 1]
    elif a.startswith("--hw-queues="):
        os.environ["ULTRA_HW_QUEUES"] = a.split("=", 1)[1]
HW_QUEUES = hwq.export_hw_queues()

import numpy as np  # noqa: E402

OPS_PER_CELL = 0.75          # SURVEY.md section 8d convention: 48 int32 VALU ops per 64-cell block column
PEAK_CLOCK_GHZ = 2.4         # MI355X_MICROARCH.md: max clock
LANES_PER_CU_CLK = 128       # 4 SIMD-32 per CU (MI355X_MICROARCH.md "CU = 4 x SIMD-32")
HBM_PEAK_GBS = 8000.0
CPU_SHARE = 16               # host cores that go with one GPU of the box (task statement)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--reads", type=int, default=1048576,
                    help="reads (instances) per batch.  The batch is the design's free parameter: launch tails and the host round trips of a "
                         "run are paid once per batch (measured, cDNA: 262144 reads per batch 10.4 M reads/s, 524288 11.0 M, 1048576 11.2 M)")
    ap.add_argument("--batches", type=int, default=1,
                    help="distinct batches per GPU (default: 1 x 1048576 = BASELINE configs[2]'s 1M reads; rounds 1-3 first ran them as 4 x 262144; "
                         "the pipelined extra leg needs 2 or more)")
    ap.add_argument("--preset", default="ont_cdna", choices=["ont_cdna", "isoseq", "ont_drna"])
    ap.add_argument("--world", default="c3", choices=["c3", "c2", "small"])
    ap.add_argument("--rehearse", action="store_true",
                    help="every rank on GPU 0, rendezvous over gloo: runs the multi-rank path (batches dealt over the ranks, "
                         "barriers, max / sum over the ranks) on a one-GPU box; the value it prints is not a measurement")
    ap.add_argument("--genes", type=int, default=0, help="override the annotation's gene count")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU-baseline budget per leg (0 = skip)")
    ap.add_argument("--e2e-rounds", type=int, default=4, help="passes over the batches through ultra_pipe (0 = skip)")
    ap.add_argument("--depth", type=int, default=2, help="batches in flight in the end-to-end leg")
    ap.add_argument("--overlap", type=int, default=1,
                    help="host threads of the timed loop (measured: no gain, the GPU is the bottleneck either way): 2 = the enumeration of the next batch rides beside the kernels of the current "
                         "one (two contexts, runs serialised), 1 = plan and run one after the other")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="timed region: 1 = one host thread, the planner of batch i + 1 behind the start pass of batch i's run (ultra_mam_run_begin / "
                         "_plan_resident_behind / _run_end; measured 1.5 %% faster, but the planner then runs inside the tail's stage timers); "
                         "0 = plan and run strictly one after the other, and the pipelined loop is an extra leg behind the timed region")
    ap.add_argument("--pipe-leg", type=int, default=1, help="the pipelined loop as an extra leg behind the timed region (0 = skip; the profile scripts do)")
    ap.add_argument("--front-rounds", type=int, default=2, help="passes over two batches through the native front end (0 = skip)")
    ap.add_argument("--micro", type=int, default=1, help="run the Myers micro-benchmarks (rank 0)")
    ap.add_argument("--micro-scale", type=float, default=1.0, help="fraction of the micro-benchmarks' pair counts")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this process (default: ultra_amd.hwq.DEFAULT)")
    ap.add_argument("--lanes", type=int, default=0, help="launch lanes (HIP streams) per context, 0 = one per hardware queue")
    ap.add_argument("--slots", type=int, default=0, help="end positions kept per pair in the first pass, 0 = library default")
    ap.add_argument("--banded-starts", type=int, default=-1, help="start pass: 1 banded, 0 full height, -1 library default")
    ap.add_argument("--check", type=int, default=48, help="instances verified against the oracle (rank 0)")
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


def cpu_baseline(idx, insts, min_acc, seconds, workers):
    import multiprocessing as mp
    from oracle import cport
    cport.lib()
    cport.set_length_prefilter(1)   # as edlib's band and the GPU path: no cells for queries more than k longer than the target
    _CPU.update(idx=idx, insts=insts, min_acc=min_acc)
    n_cal = min(24, len(insts))
    per = _cpu_worker(range(n_cal)) / n_cal                     # calibrate, then size both samples to `seconds`
    n1 = int(max(8, min(len(insts), seconds / per)))
    t1 = _cpu_worker(range(n1))
    n = int(max(workers * 8, seconds / per * workers))          # wraps around the sample when it is smaller
    chunks = [[i % len(insts) for i in range(j, n, workers)] for j in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(workers) as pool:
        pool.map(_cpu_worker, chunks)
    wall = time.perf_counter() - t0
    return {"value": n / wall, "unit": "reads/s", "cores": workers, "kind": "port",
            "sample": "%d reads of batch 0 through oracle/classify_port.py + liboracle.so (C Myers bit-vector without an "
                      "Ukkonen band, pairs whose query is more than k longer than the target answered without a DP, + "
                      "chainers), %d fork workers (pool capped at the box's CPU share of %d; %d cores "
                      "visible), %.1f s wall" % (n, workers, CPU_SHARE, len(os.sched_getaffinity(0)), wall),
            "one_thread": {"value": n1 / t1, "unit": "reads/s", "cores": 1, "sample": "%d reads, %.1f s" % (n1, t1)},
            "reference_cli": "uLTRA align not present on the box (edlib, namfinder, minimap2 are not installable offline)"}


def micro_benchmarks(lib, ctx, peak_tops, err, scale):
    """kernel-only Myers micro-benchmarks through the C-ABI (ultra_myers_stage / _run)."""
    from ultra_amd import synth
    out = {}
    for name in synth.MICRO:
        n = max(1024, int(synth.MICRO[name]["n_pairs"] * scale))
        pairs = synth.make_micro_pairs(name, err=err, planted=0.5, n_pairs=n)
        ctx.myers_stage(*pairs)
        ctx.myers_run()
        best = None
        for _ in range(3):
            ctx.myers_run()
            cells, ms = ctx.myers_timing()
            if best is None or ms[4] < best[1][4]:
                best = (cells, ms, ctx.myers_class_timing())
        cells, ms, cls = best
        fwd_tcups = cells / (ms[4] * 1e-3) * 1e-12 if ms[4] > 0 else None
        res = ctx.myers_fetch()
        out[name] = {
            "pairs": int(len(pairs[5])), "cells": cells, "hit_fraction": float((res["ed"] >= 0).mean()),
            "forward_ms": ms[4], "starts_ms": ms[1], "traceback_ms": ms[2], "all_passes_ms": ms[3],
            "forward_tcups": fwd_tcups, "forward_frac_of_int32_valu_peak": OPS_PER_CELL * fwd_tcups / peak_tops if fwd_tcups else None,
            "all_passes_tcups": cells / (ms[3] * 1e-3) * 1e-12 if ms[3] > 0 else None,
            "forward_buckets": {"bucket%d_wmax%d" % k: {"cells": v[0], "ms": v[1], "tcups": v[0] / v[1] * 1e-9 if v[1] > 0 else None}
                                for k, v in sorted(cls.items())},
        }
    return out


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import index as native_index
    from ultra_amd import shard, synth

    # ---- synthetic workload: the same annotation on every rank, the job's batches dealt over the ranks ----
    P = synth.PRESETS[args.preset]
    wkw = dict(synth.WORLDS[args.world])
    if args.genes:
        wkw["n_genes"] = args.genes
    t0 = time.perf_counter()
    w = synth.make_world(seed=20241008, **wkw)
    idx = w.index()
    flat = synth.FlatWorld(w)
    world_s = time.perf_counter() - t0
    my_batches = shard.deal_batches(world * args.batches, rank, world, batch=1)     # batch ids of this rank
    t0 = time.perf_counter()
    # the ranks of a node generate side by side: each takes its share of the cores that go with the node's GPUs
    gen_threads = max(1, min(CPU_SHARE, len(os.sched_getaffinity(0)) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))))
    sets = [synth.make_reads_native(flat, args.reads, seed=1000 + b, preset=args.preset, threads=gen_threads) for b in my_batches]
    reads_s = time.perf_counter() - t0
    import torch  # noqa: F401  (before libultra_hip.so is loaded: both must end up on the same HIP runtime - torch's)
    t0 = time.perf_counter()
    nix = native_index.NativeIndex(*idx)
    index_s = time.perf_counter() - t0

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        workers = max(1, min(CPU_SHARE, len(os.sched_getaffinity(0))))
        n_sample = min(len(sets[0]), 4096)
        sample = [sets[0].instance(i) for i in range(n_sample)]
        cpu = cpu_baseline(idx, sample, P["min_acc"], args.cpu_seconds, workers)

    # ---- GPU: bind the rank's device before anything else touches HIP ----
    import torch
    import torch.distributed as dist
    from ultra_amd import lib
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    if args.rehearse:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    red_dev = "cuda" if world > 1 and not args.rehearse else None
    cfg = {}
    if args.lanes:
        cfg["launch_lanes"] = args.lanes
    if args.slots:
        cfg["loc_slots"] = args.slots
    if args.banded_starts >= 0:
        cfg["banded_starts"] = args.banded_starts
    t0 = time.perf_counter()
    dix = native_index.DeviceIndex(nix, local)
    dindex_s = time.perf_counter() - t0
    ctxs = [lib.Context(local, cfg=cfg or None) for _ in sets]
    info = ctxs[0].device_info()
    stage_ms = []
    for c, rs in zip(ctxs, sets):
        t0 = time.perf_counter()
        c.mam_stage_nams(dix, rs.nam_off, rs.nam_part, rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
        stage_ms.append((time.perf_counter() - t0) * 1e3)
    counts = [c.mam_counts() for c in ctxs]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    nb = len(ctxs)
    for i in range(args.warmup):
        ctxs[i % nb].mam_plan_resident(dix)
        ctxs[i % nb].mam_run(P["min_acc"])
    # once more, staged a second time: what a batch costs to stage when the buffers exist already
    t0 = time.perf_counter()
    ctxs[0].mam_stage_nams(dix, sets[0].nam_off, sets[0].nam_part, sets[0].nam_x, sets[0].nam_y, sets[0].reads, sets[0].read_off)
    restage_ms = (time.perf_counter() - t0) * 1e3
    ctxs[0].mam_run(P["min_acc"])
    import threading
    run_lock = threading.Lock()      # a run fills the chip: one at a time; the other thread's enumeration rides beside it
    n_thr = max(1, min(args.overlap, nb))
    errors = []

    def new_acc():
        return ({}, {"fwd": 0.0, "rev": 0.0, "tb": 0.0, "all": 0.0, "fwdk": 0.0, "build": 0.0, "chain": 0.0, "gather": 0.0},
                {"cells_run": 0.0, "cells_skip": 0.0, "pairs_skip": 0.0, "plan_s": 0.0, "run_s": 0.0})

    def collect(c, class_ms, acc, tot):
        cells, ms = c.myers_timing()
        tot["cells_run"] += cells
        sk = c.myers_cell_counts()
        tot["cells_skip"] += sk[1]; tot["pairs_skip"] += sk[2]
        acc["fwd"] += ms[0]; acc["rev"] += ms[1]; acc["tb"] += ms[2]; acc["all"] += ms[3]; acc["fwdk"] += ms[4]
        mm = c.mam_timing()
        acc["build"] += mm[0]; acc["chain"] += mm[1]; acc["gather"] += mm[2]
        for key, (cc, t) in c.myers_class_timing().items():
            a = class_ms.setdefault(key, [0.0, 0.0, 0])
            a[0] += cc; a[1] += t; a[2] += 1

    class_ms, acc, tot = new_acc()
    host_ms = {"run_begin": 0.0, "plan_behind": 0.0, "run_end": 0.0}   # where the pipelined loop's host thread spends a step

    def steps_of(tid):
        try:
            for i in range(tid, args.steps, n_thr):     # step i works on context i % nb: never shared between the threads (nb % n_thr == 0)
                c = ctxs[i % nb]
                t1 = time.perf_counter()
                c.mam_plan_resident(dix)    # a4's enumeration, pair sort, work lists: from the request resident in HBM
                t2 = time.perf_counter()
                with run_lock:
                    t3 = time.perf_counter()
                    c.mam_run(P["min_acc"])
                    tot["run_s"] += time.perf_counter() - t3
                    tot["plan_s"] += t2 - t1
                    collect(c, class_ms, acc, tot)
        except Exception as e:      # noqa: BLE001 - reported by the main thread
            errors.append(e)

    def steps_pipelined(class_ms, acc, tot, host_ms):
        # One host thread, consecutive batches pipelined on the device: the run of batch i is enqueued as a whole
        # (ultra_mam_run_begin returns once the forward pass's counts are on the host), then the planner of batch i + 1 goes
        # out on its own context, ordered behind that run's start pass: it shares the chip with the run's latency-bound tail
        # (traceback, MAM build, chaining), not with the passes that fill it.  K plans and K runs inside the timed region;
        # the first plan has nothing to ride under.
        ctxs[0].mam_plan_resident(dix)
        for i in range(args.steps):
            c = ctxs[i % nb]
            t3 = time.perf_counter()
            c.mam_run_begin(P["min_acc"])
            t4 = time.perf_counter()
            if i + 1 < args.steps:
                ctxs[(i + 1) % nb].mam_plan_resident_behind(dix, c)
            t5 = time.perf_counter()
            c.mam_run_end()
            t6 = time.perf_counter()
            tot["run_s"] += t6 - t3
            host_ms["run_begin"] += (t4 - t3) * 1e3; host_ms["plan_behind"] += (t5 - t4) * 1e3; host_ms["run_end"] += (t6 - t5) * 1e3
            collect(c, class_ms, acc, tot)

    if nb % n_thr:
        n_thr = 1
    pipelined = bool(args.pipeline) and n_thr == 1 and nb >= 2
    if pipelined:       # the warm-up of this loop's shape: the contexts have met every buffer size already, the streams have not met each other
        for i in range(min(args.warmup, 2)):
            ctxs[i % nb].mam_plan_resident(dix)
            ctxs[i % nb].mam_run_begin(P["min_acc"])
            ctxs[(i + 1) % nb].mam_plan_resident_behind(dix, ctxs[i % nb])
            ctxs[i % nb].mam_run_end()
    barrier()
    t0 = time.perf_counter()
    if pipelined:
        steps_pipelined(class_ms, acc, tot, host_ms)
    elif n_thr == 1:
        steps_of(0)
    else:
        ths = [threading.Thread(target=steps_of, args=(t,)) for t in range(n_thr)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    barrier()
    elapsed = time.perf_counter() - t0
    if errors:
        raise errors[0]
    # ---- the same steps one after the other (plan, then run; nothing overlaps): the parts by themselves ----
    seq_class, seq_acc, seq_tot = new_acc()
    n_seq = nb if pipelined else 0
    t0 = time.perf_counter()
    for i in range(n_seq):
        c = ctxs[i % nb]
        t1 = time.perf_counter()
        c.mam_plan_resident(dix)
        t2 = time.perf_counter()
        c.mam_run(P["min_acc"])
        seq_tot["run_s"] += time.perf_counter() - t2
        seq_tot["plan_s"] += t2 - t1
        collect(c, seq_class, seq_acc, seq_tot)
    seq_elapsed = time.perf_counter() - t0
    pipe_leg = None
    if not pipelined:
        seq_acc, seq_tot, n_seq, seq_elapsed = acc, tot, args.steps, elapsed
        if n_thr == 1 and nb >= 2 and world == 1 and args.pipe_leg:
            # the pipelined loop as an extra leg (not `value`): its planner runs inside the tail's stage timers
            _, p_acc, p_tot = new_acc()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps_pipelined({}, p_acc, p_tot, host_ms)
            torch.cuda.synchronize()
            p_el = time.perf_counter() - t0
            pipe_leg = {"ms_per_step": p_el / args.steps * 1e3, "reads_per_s": args.reads * args.steps / p_el, "steps": args.steps,
                        "host_ms_per_step": {k: v / args.steps for k, v in host_ms.items()},
                        "stage_ms_per_step": {"myers_forward": p_acc["fwd"] / args.steps, "myers_starts": p_acc["rev"] / args.steps,
                                              "myers_traceback": p_acc["tb"] / args.steps, "mam_build": p_acc["build"] / args.steps,
                                              "chain": p_acc["chain"] / args.steps},
                        "note": "ultra_mam_run_begin(i) -> ultra_mam_plan_resident_behind(i + 1) -> ultra_mam_run_end(i) on one host thread: the "
                                "planner of the next batch shares the chip with the run's traceback, MAM build and chaining.  Both slow "
                                "down (the tail is not idle, it is short of wavefront slots): a little faster than plan-then-run, "
                                "not by the planner's 4 ms"}
    cells_run, cells_skip, pairs_skip, plan_s, run_s = (tot[k] for k in ("cells_run", "cells_skip", "pairs_skip", "plan_s", "run_s"))
    K = args.steps
    elapsed, reads_all, cells_all = shard.reduce_job(elapsed, args.reads * K, cells_run, device=red_dev)
    run_max, _, cells_skip_all = shard.reduce_job(run_s, 0, cells_skip, device=red_dev)

    # ---- results back to the host (PCIe; outside the timed region) + parity spot check against the oracle ----
    t1 = time.perf_counter()
    res0 = ctxs[0].mam_fetch(want_names=True)
    fetch_s = time.perf_counter() - t1
    # Algorithmic HBM bytes of a step: m + n sequence bytes + 40 B descriptor / result header per pair
    alg_bytes = None
    if rank == 0:
        plan0 = ctxs[0].mam_fetch_plan()
        alg_bytes = int(plan0["q_len"].astype(np.int64).sum() + plan0["t_len"].astype(np.int64).sum() + 40 * len(plan0["k"]))
        del plan0
    checked = 0
    if rank == 0 and args.check > 0:
        import struct
        from oracle import classify_port
        t1 = time.perf_counter()
        out0 = stage.results_from_names(res0, np.diff(sets[0].read_off).tolist())
        topy_s = time.perf_counter() - t1
        for i in range(0, len(sets[0]), max(1, len(sets[0]) // args.check)):
            read, sol, _ = sets[0].instance(i)
            want = classify_port.classify(sol, *idx, read, None, P["min_acc"])
            g = out0[i]
            ok = (want[0] == g[0] and struct.pack("d", float(want[1])) == struct.pack("d", float(g[1]))
                  and tuple(tuple(m) for m in want[2]) == tuple(tuple(m) for m in g[2]))
            if not ok:
                raise SystemExit("bench.py: instance %d differs from the oracle - refusing to report a number" % i)
            checked += 1
    else:
        topy_s = 0.0

    # ---- end to end: host arrays -> ultra_pipe -> fetched results (NOT `value`) ----
    e2e = None
    if args.e2e_rounds > 0:
        for c in ctxs[1:]:
            c.close()                       # the pipe's slots bring their own contexts
        ctxs = ctxs[:1]
        pipe = lib.Pipe(nix, local, depth=args.depth, cfg=cfg or None)
        fields = ("nam_off", "nam_part", "nam_x", "nam_y", "reads", "read_off")
        pageable = [tuple(getattr(rs, f) for f in fields) for rs in sets]
        pins = [[lib.PinnedArray(getattr(rs, f)) for f in fields] for rs in sets]   # a caller that parses into page-locked buffers
        pinned = [tuple(pa.a for pa in row) for row in pins]

        def run_e2e(batches, rounds):
            def submit(b):
                pipe.submit(*batches[b % nb], P["min_acc"])
            for b in range(2 * args.depth):     # every slot allocates its buffers on its first batches
                submit(b)
                if pipe.pending() >= args.depth:
                    pipe.wait(); pipe.release()
            while pipe.pending():
                pipe.wait(); pipe.release()
            total = rounds * nb
            sent = done = 0
            tims, chains = [], 0
            barrier()
            t0 = time.perf_counter()
            while done < total:
                while sent < total and sent - done < args.depth:
                    submit(sent); sent += 1
                r = pipe.wait()
                chains += int(r["chain_off"][-1])
                tims.append(pipe.timing())
                pipe.release(); done += 1
            barrier()
            secs = time.perf_counter() - t0
            secs, n_reads, _ = shard.reduce_job(secs, args.reads * total, 0.0, device=red_dev)
            avg = lambda k: sum(t[k] for t in tims) / len(tims)
            return n_reads / secs, total, chains, {"stage_h2d_device_plan_sort": avg("stage_ms"), "waiting_for_gpu": avg("wait_ms"),
                                                   "run": avg("run_ms"), "fetch_d2h": avg("fetch_ms")}
        rate_pg, total, chains, ms_pg = run_e2e(pageable, args.e2e_rounds)
        rate_pin, total, chains, ms_pin = run_e2e(pinned, args.e2e_rounds)
        pipe.close()
        for row in pins:
            for pa in row:
                pa.close()
        e2e = {"reads_per_s": rate_pin, "batches": total, "reads_per_batch": args.reads, "depth": args.depth,
               "host_threads": args.depth, "input": "NAM chains + reads as flat arrays in page-locked host memory (ultra_host_alloc)",
               "output": "chains with kind / chromosome in pinned host memory", "per_batch_ms": ms_pin,
               "pageable_input": {"reads_per_s": rate_pg, "per_batch_ms": ms_pg},
               "chain_rows_fetched": chains,
               "synchronous_caller_reads_per_s": args.reads / ((restage_ms + seq_tot["run_s"] / max(n_seq, 1) * 1e3 + fetch_s * 1e3) * 1e-3)}

    # ---- the per-read host steps in front of the MAM stage (align.py:427-489), natively: seed text -> SAM-ready arrays ----
    front_leg = None
    if rank == 0 and args.front_rounds > 0:
        from ultra_amd import front as front_mod
        fr = front_mod.Front(ctxs[0], dix, nix, {c: 200000 for c in range(wkw["n_chr"])})
        pbs = []
        for rs in sets[:2]:
            text, line_off = synth.seed_text(rs)
            none = np.zeros(len(rs), np.int64)
            pbs.append(front_mod.PackedBatch(rs.reads, rs.read_off, text, line_off[:-1], line_off[1:], none, none))
        prm = (1200000, 0.95, 5, 8, P["min_acc"])       # uLTRA's defaults: --max_intron, --dropoff, --max_loc, --reduce_read_ployA
        fr.run(pbs[0], *prm, copy=False)
        t0 = time.perf_counter()
        ms_acc, n_loc = [0.0] * 6, 0
        for _ in range(args.front_rounds):
            for pb in pbs:
                r = fr.run(pb, *prm, copy=False)
                n_loc += r["n_loc"]
                ms_acc = [a + b for a, b in zip(ms_acc, r["ms"])]
        secs = time.perf_counter() - t0
        n_calls = args.front_rounds * len(pbs)
        # two workers (each its own context and front object, half the host threads) on alternate batches: one's parsing and
        # selection run beside the other's device work
        two_rate = None
        try:
            import threading
            half = max(1, min(16, len(os.sched_getaffinity(0))) // 2)
            ctx2 = ctxs[1] if len(ctxs) > 1 else lib.Context(local, cfg=cfg or None)
            frs = [front_mod.Front(ctxs[0], dix, nix, {c: 200000 for c in range(wkw["n_chr"])}, n_threads=half),
                   front_mod.Front(ctx2, dix, nix, {c: 200000 for c in range(wkw["n_chr"])}, n_threads=half)]
            for f, pb in zip(frs, pbs):
                f.run(pb, *prm, copy=False)
            rounds2 = max(2, args.front_rounds)
            errs = []

            def worker(f, pb):
                try:
                    for _ in range(rounds2):
                        f.run(pb, *prm, copy=False)
                except Exception as e:      # noqa: BLE001
                    errs.append(e)
            ths = [threading.Thread(target=worker, args=(f, pb)) for f, pb in zip(frs, pbs)]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            two_s = time.perf_counter() - t0
            if errs:
                raise errs[0]
            two_rate = args.reads * rounds2 * len(frs) / two_s
            for f in frs:
                f.close()
        except Exception as e:              # noqa: BLE001
            two_rate = "failed: %r" % (e,)
        t0 = time.perf_counter()
        res = fr.run(pbs[0], *prm)                       # the same with every array copied out into numpy
        numpy_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        n_t = min(res.n_loc, 20000)
        for l in range(n_t):
            res.mam(l)
        tuple_us = (time.perf_counter() - t0) / max(n_t, 1) * 1e6
        # the Python shims this replaces, on a sample (they are the reference-shaped per-read loop)
        from ultra_amd import align as align_mod
        n_py = 2000
        hits = [(pbs[0].text[int(pbs[0].fw_begin[i]):int(pbs[0].fw_end[i])].tobytes().decode().splitlines(True), []) for i in range(n_py)]
        t0 = time.perf_counter()
        align_mod.chain_reads_batch(hits, {c: 200000 for c in range(wkw["n_chr"])}, 1200000, 0.95, 5, ctx=ctxs[0])
        py_us = (time.perf_counter() - t0) / n_py * 1e6
        front_leg = {"align_single_front_reads_per_s": args.reads * n_calls / secs,
                     "what": "ultra_front_run per batch of %d reads: hit lines (text, %.0f bytes per read) + raw reads -> instances and bounds "
                             "-> one NAM chaining launch -> location selection -> device-planned MAM stage -> results in host arrays; "
                             "align.py:427-489 for the whole batch, one process, %d host threads"
                             % (args.reads, pbs[0].text.size / float(args.reads), min(16, len(os.sched_getaffinity(0)))),
                     "ms_per_batch": {k: v / n_calls for k, v in zip(("parse_bounds", "nam_chaining", "selection", "stage", "run", "fetch"), ms_acc)},
                     "locations_per_read": n_loc / float(args.reads * n_calls),
                     "two_workers_reads_per_s": two_rate,
                     "with_numpy_copies_reads_per_s": args.reads / numpy_s,
                     "python_tuple_us_per_location_on_access": tuple_us,
                     "python_shims_us_per_read_phases_1_to_3": py_us}
        fr.close()

    micro = None
    if rank == 0 and args.micro:
        peak_tops = info["n_cu"] * LANES_PER_CU_CLK * PEAK_CLOCK_GHZ * 1e9 * 1e-12
        micro = micro_benchmarks(lib, ctxs[0], peak_tops, P["err"], args.micro_scale)

    if rank == 0:
        peak_ops = info["n_cu"] * LANES_PER_CU_CLK * PEAK_CLOCK_GHZ * 1e9
        n_pairs = float(np.mean([c[2] for c in counts]))
        n_cands = float(np.mean([c[1] for c in counts]))
        cells_step = cells_all / world / K                 # cells the forward pass computes
        cells_ref_step = cells_step + cells_skip_all / world / K   # sum of m * n over every edlib call of the reference
        if cpu:
            cpr = cells_step / args.reads
            cpu["gcups"] = cpu["value"] * cpr * 1e-9
            cpu["one_thread"]["gcups"] = cpu["one_thread"]["value"] * cpr * 1e-9
            cpu["gcups_note"] = "reads/s x the batch's computed cells per read (%.3g)" % cpr
        dom = max(class_ms.items(), key=lambda kv: kv[1][1]) if class_ms else None
        roof = None
        if dom:
            (G, Wd), (c, ms, launches) = dom
            achieved = OPS_PER_CELL * c / launches / (ms / launches * 1e-3) * 1e-12
            kname = "myers_mega<0, %d>" % G   # forward pass (MODE 0) of register bucket G
            # HBM bytes from the committed rocprofv3 --pmc passes of this command (profiles/); not measured in-run:
            # counters need the profiler.  The record names its command and date.
            traffic = fam_traffic = None
            traffic_src = None
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            # ... of the default command only: another preset or batch size has no counters on record
            if os.path.exists(pmc) and (args.preset, args.world, args.reads, args.genes) == ("ont_cdna", "c3", 1048576, 0):
                try:
                    rec = json.load(open(pmc))
                    if rec.get("dominant_kernel") == kname:
                        traffic = rec.get("hbm_bytes_per_launch")
                    fam_traffic = rec.get("forward_family", {}).get("hbm_bytes_per_step")
                    traffic_src = {"file": "profiles/pmc_latest.json", "recorded": rec.get("recorded"), "command": rec.get("command")}
                except Exception:
                    traffic = fam_traffic = None
            # The forward pass is ONE kernel template instantiated per register bucket; the (up to 5) instantiations
            # of a step are launched together on their own streams and share the CUs, so the duration of a single one says
            # how long it shared the chip, not how fast it is.  The roofline is therefore taken over the family:
            # all forward cells of the step / (first launch start -> last launch end, HIP events on those streams).
            fam_ms = acc["fwdk"] / K
            fam_ach = OPS_PER_CELL * cells_step / (fam_ms * 1e-3) * 1e-12 if fam_ms > 0 else 0.0
            whole_ach = OPS_PER_CELL * cells_step / (acc["all"] / K * 1e-3) * 1e-12 if acc["all"] > 0 else 0.0
            roof = {"bound": "valu_int32", "kernel": "myers_mega<0, 0..4> (forward pass: concurrent launches, one per register bucket)",
                    "achieved": fam_ach, "peak": peak_ops * 1e-12, "unit": "Tops/s", "frac": fam_ach / (peak_ops * 1e-12),
                    "traffic": fam_traffic, "traffic_source": traffic_src, "cells_per_launch": cells_step, "avg_launch_ms": fam_ms,
                    "ops_per_cell": OPS_PER_CELL,
                    # profiles/r01_valu_issue_rate.txt: a SIMD issues one plain 32-bit VALU instruction per 2.44 cycles of
                    # the nominal 2.4 GHz under this kind of load (the chip down-clocks)
                    "peak_sustained_measured": peak_ops * 1e-12 * 2.0 / 2.44,
                    "frac_of_sustained": fam_ach / (peak_ops * 1e-12 * 2.0 / 2.44),
                    "whole_myers_stage": {"achieved": whole_ach, "frac": whole_ach / (peak_ops * 1e-12),
                                          "note": "forward cells / all three passes (starts and traceback add no cells)"},
                    "largest_instantiation": {"kernel": kname, "achieved": achieved, "frac": achieved / (peak_ops * 1e-12),
                                              "traffic": traffic, "cells_per_launch": c / launches, "avg_launch_ms": ms / launches,
                                              "note": "runs concurrently with the other buckets' launches"},
                    "hbm": {"algorithmic_bytes_per_step": alg_bytes, "achieved_GBs":
                            alg_bytes / (acc["all"] / K * 1e-3) * 1e-9 if alg_bytes and acc["all"] > 0 else None,
                            "peak_GBs": HBM_PEAK_GBS}}
        line = {
            "metric": "reads aligned/sec (whole node), MAM stage: candidate enumeration + Myers HW matching + MAM chaining",
            "value": reads_all / elapsed, "unit": "reads/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 bit-vectors / int32 scores (Myers), f64 (chaining)", "data": "synthetic",
            "config": {"workload": {"ont_cdna": "BASELINE configs[2] synthetic form (SURVEY.md 8d C3)", "isoseq": "BASELINE configs[1] synthetic form (SURVEY.md 8d C2)",
                                    "ont_drna": "BASELINE configs[4] synthetic form (SURVEY.md 8d C5)"}[args.preset] + " on world '%s': %d-gene / %d-chromosome GENCODE-shaped annotation with "
                                   "overlapping genes (%d transcripts, %d segments), %d distinct %s reads per GPU (err %.2f, lengths drawn from the "
                                   "preset's model and cut to the transcript: mean %d bp) as %d batch(es) of %d, reads + NAM chains + index "
                                   "resident in HBM, one step = enumeration + alignment + chaining of one batch"
                                   % (args.world, wkw["n_genes"], wkw["n_chr"], len(w.transcripts), len(w.ref_segment_sequences), args.reads * nb, args.preset,
                                      P["err"], int(np.mean([rs.reads.size / max(len(rs), 1) for rs in sets])), nb, args.reads),
                       "reads_per_gpu_per_step": args.reads, "distinct_batches_per_gpu": nb, "pairs_per_step_per_gpu": n_pairs,
                       "candidates_per_step_per_gpu": n_cands, "cells_per_step_per_gpu": cells_step, "cells_per_read": cells_step / args.reads,
                       "cells_reference_per_step_per_gpu": cells_ref_step, "cells_computed_per_step_per_gpu": cells_step,
                       "pairs_answered_by_length_per_step": pairs_skip / K,
                       "cells_note": "cells_reference = sum of m * n over every edlib.align call the reference makes; cells_computed "
                                     "leaves out the pairs whose query is more than k longer than the target (distance >= m - n > k: "
                                     "edlib answers -1 without reaching the last row, the library at stage time); every GCUPS / "
                                     "roofline figure is taken over cells_computed",
                       "hw_queues": HW_QUEUES, "parallelism": "batches dealt over %d rank(s), no collective" % world,
                       **({"rehearsal": "all ranks share GPU 0, gloo rendezvous: the path is exercised, the value is not a measurement"} if args.rehearse else {})},
            "myers_gcups_forward": cells_all / (acc["fwd"] * 1e-3) * 1e-9 / world if acc["fwd"] > 0 else None,
            "myers_gcups_all_passes": cells_all / (acc["all"] * 1e-3) * 1e-9 / world if acc["all"] > 0 else None,
            "kernels_only": {"reads_per_s": args.reads * n_seq / seq_tot["run_s"] if seq_tot["run_s"] > 0 else None,
                             "ms_per_step": seq_tot["run_s"] / max(n_seq, 1) * 1e3,
                             "note": "ultra_mam_run alone (candidates and sorted pairs already staged), nothing beside it: what round 2 reported as value"},
            "plan_ms_per_step": seq_tot["plan_s"] / max(n_seq, 1) * 1e3,
            "sequential": {"ms_per_step": seq_elapsed / max(n_seq, 1) * 1e3, "reads_per_s": args.reads * n_seq / seq_elapsed if seq_elapsed > 0 else None,
                           "steps": n_seq,
                           "note": "plan, then run, nothing overlapping (after the timed region; round 3's first value); plan_ms_per_step, "
                                   "kernels_only and stage_ms_sequential are its parts",
                           "stage_ms_per_step": {"myers_forward": seq_acc["fwd"] / max(n_seq, 1), "myers_starts": seq_acc["rev"] / max(n_seq, 1),
                                                 "myers_traceback": seq_acc["tb"] / max(n_seq, 1), "myers_total": seq_acc["all"] / max(n_seq, 1),
                                                 "mam_build": seq_acc["build"] / max(n_seq, 1), "chain": seq_acc["chain"] / max(n_seq, 1),
                                                 "chain_gather": seq_acc["gather"] / max(n_seq, 1)}},
            "host_threads": n_thr,
            "pipelined": pipelined,
            "pipelined_host_ms_per_step": {k: v / K for k, v in host_ms.items()} if pipelined else None,
            "pipelined_leg": pipe_leg,
            "overlap_note": "timed region: one host thread; the run of batch i is enqueued as a whole (ultra_mam_run_begin), the planner of "
                            "batch i + 1 (its own context) is ordered behind that run's start pass and shares the chip with its traceback, "
                            "MAM build and chaining (ultra_mam_plan_resident_behind), then ultra_mam_run_end.  K plans + K runs in the timed "
                            "region; stage_ms_per_step are the timed region's own event timers (the tail's stages are longer than in "
                            "`sequential` because the planner runs beside them)" if pipelined else
                            "plan and run one after the other" if n_thr == 1 else
                            "2 host threads: the enumeration + pair sort of batch i + 1 (its own context) beside the kernels of batch i, uncoordinated",
            "stage_ms_per_step": {"myers_forward": acc["fwd"] / K, "myers_starts": acc["rev"] / K, "myers_traceback": acc["tb"] / K,
                                  "myers_total": acc["all"] / K, "mam_build": acc["build"] / K, "chain": acc["chain"] / K,
                                  "chain_gather": acc["gather"] / K},
            "outside_timed_region": {"stage_ms_first": stage_ms, "stage_ms": restage_ms,
                                     "stage_note": "ultra_mam_stage_nams: H2D of NAM chains + reads, device plan, pair sort, work lists",
                                     "fetch_results_pcie_ms": fetch_s * 1e3, "python_tuples_ms": topy_s * 1e3,
                                     "world_s": world_s, "reads_s": reads_s, "host_index_s": index_s, "device_index_s": dindex_s},
            "end_to_end": e2e,
            "front_end": front_leg,
            "forward_buckets": {"bucket%d_wmax%d" % k: {"cells_per_step": v[0] / v[2], "ms_per_step": v[1] / v[2],
                                                        "tcups": v[0] / v[1] * 1e-9 if v[1] > 0 else None}
                                for k, v in sorted(class_ms.items())},
            "micro": micro,
            "device": info, "parity_checked_instances": checked,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    for c in ctxs:
        c.close()
    dix.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
