"""bench.py's multi-rank path on a one-GPU box: two ranks started the way the driver starts them
(`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), both on GPU 0 with a gloo rendezvous
(`--rehearse`).  What it covers: batches dealt over the ranks (shard.deal_batches), the barriers around the timed
steps, max-over-ranks time and summed reads (shard.reduce_job), one JSON line from rank 0.  What it cannot cover
here: RCCL itself and a second device (the driver's 8-GPU run does).  Collected late (file name) so that a launcher
problem on a box cannot hide the parity tests behind `pytest -x`; the allocation-free entry point's test stays last."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_share_one_gpu():
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--rehearse",
           "--world", "small", "--reads", "8192", "--batches", "2", "--e2e-rounds", "1", "--micro", "0",
           "--cpu-seconds", "0", "--check", "16"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak"
    # whole-job value: both ranks' reads over the slowest rank's time
    assert abs(d["value"] - 2 * 8192 * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    assert d["parity_checked_instances"] >= 16
    assert "rehearsal" in d["config"]
