"""CPU, world_size 2 over gloo: the multi-GPU path shards reads with no data-path collective; ranks only
agree on timing/totals.  Also checks that packing a shard is independent of what other ranks pack."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from ultra_amd import classify_read_with_mams as stage
    from ultra_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.make_world(seed=5, n_genes=20)
    insts = synth.make_instances(w, 60, seed=9)                 # the same job on every rank
    whole = stage.prepare_batch([(sol, read) for read, sol, _ in insts], *w.index())
    cells = []
    for i in range(len(insts)):
        b = stage.pack_plans([whole.plans[i]])
        cells.append(b.cells)
    mine = shard.deal_batches(len(insts), rank, world, batch=7, cells=cells)
    part = stage.pack_plans([whole.plans[i] for i in mine])
    elapsed, reads, tot_cells = shard.reduce_job(1.0 + rank, len(mine), part.cells)
    q.put((rank, mine, part.cells, part.n_inst, elapsed, reads, tot_cells, sum(cells)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_shard_the_job_without_exchanging_data():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, c0, n0, e0, reads0, cells0, all0), (r1, m1, c1, n1, e1, reads1, cells1, all1) = got
    assert sorted(m0 + m1) == list(range(60)) and not set(m0) & set(m1)     # disjoint cover
    assert n0 == len(m0) and n1 == len(m1)
    assert e0 == e1 == 2.0                                                    # max over ranks
    assert reads0 == reads1 == 60.0
    assert cells0 == cells1 == float(c0 + c1) == float(all0)                  # per-read work is shard-independent
    assert abs(c0 - c1) < 0.35 * (c0 + c1)                                    # cell-balanced deal


def _plan_digests(batch_ids, reads_per_batch):
    """What bench.py does per rank: the batches it was dealt, generated from their ids alone, planned with the native
    planner; one digest per batch over every array the MAM stage would be handed."""
    import hashlib
    from ultra_amd import index as native_index
    from ultra_amd import synth
    w = synth.make_world(seed=20241008, n_genes=120, n_chr=3)
    flat = synth.FlatWorld(w)
    nix = native_index.NativeIndex(*w.index())
    out = {}
    try:
        for b in batch_ids:
            rs = synth.make_reads_native(flat, reads_per_batch, seed=1000 + b, preset="ont_cdna", threads=2)
            pl = nix.plan_arrays(rs.nam_off, rs.nam_part.reshape(-1, 3), rs.nam_x, rs.nam_y, rs.reads, rs.read_off)
            h = hashlib.sha256()
            for name in sorted(pl.packed):
                h.update(name.encode())
                h.update(np.ascontiguousarray(pl.packed[name]).tobytes())
            out[b] = (h.hexdigest(), float(pl.cells), int(pl.n_inst))
            pl.close()
    finally:
        nix.close()
    return out


def _bench_worker(rank, world, port, q, batches_per_rank, reads_per_batch):
    import torch.distributed as dist
    from ultra_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.deal_batches(world * batches_per_rank, rank, world, batch=1)      # bench.py's deal
    got = _plan_digests(mine, reads_per_batch)
    elapsed, reads, cells = shard.reduce_job(0.5, len(mine) * reads_per_batch, sum(v[1] for v in got.values()))
    q.put((rank, got, reads, cells))
    dist.barrier()
    dist.destroy_process_group()


def test_union_of_rank_plans_equals_the_single_rank_plan():
    """bench.py --gpus N deals batch ids over the ranks and every rank generates and plans its batches from the ids
    alone.  The batches two ranks plan are, byte for byte, the batches one rank plans for the whole job."""
    batches_per_rank, reads_per_batch, world = 2, 400, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q, batches_per_rank, reads_per_batch)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    single = _plan_digests(range(world * batches_per_rank), reads_per_batch)
    union = {}
    for rank, plans, reads, cells in got:
        assert not set(plans) & set(union)                                    # disjoint
        union.update(plans)
        assert reads == float(world * batches_per_rank * reads_per_batch)
        assert cells == sum(v[1] for v in single.values())
    assert union == single
    assert len({v[0] for v in single.values()}) == len(single)                # distinct batches, not a replay


def test_shard_helpers():
    from ultra_amd import shard
    for n in (0, 1, 7, 1000, 1001):
        for world in (1, 2, 3, 8):
            rs = [shard.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            dealt = sorted(j for r in range(world) for j in shard.deal_batches(n, r, world, batch=10))
            assert dealt == list(range(n))
    assert shard.reduce_job(1.5, 10, 20) == (1.5, 10.0, 20.0)
