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


def test_shard_helpers():
    from ultra_amd import shard
    for n in (0, 1, 7, 1000, 1001):
        for world in (1, 2, 3, 8):
            rs = [shard.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            dealt = sorted(j for r in range(world) for j in shard.deal_batches(n, r, world, batch=10))
            assert dealt == list(range(n))
    assert shard.reduce_job(1.5, 10, 20) == (1.5, 10.0, 20.0)
