"""Hardware queues of the process.

The launches of one pass (five forward buckets, up to nine traceback kernels) are dealt over HIP streams; streams
beyond the process's hardware queues share a queue and serialise.  HIP reads GPU_MAX_HW_QUEUES (default 4) once,
at its first call, so a harness that wants more has to export it before anything touches the GPU.  A caller that
initialised HIP earlier simply runs with what it has: ultra_ctx_create takes one launch lane per queue it finds.
"""
import os

HIP_DEFAULT = 4
POOL_MAX = 32     # the GPU pool refuses more
# Default 13 = the 12 streams of a context (8 launch lanes + 4 for the traceback's full-height launches) + the null stream:
# every context's streams then sit on 12 queues of their own, the same 12 for every context of the process.  Measured in
# round 3 (scripts/dev/hwq_sweep.sh, gpurun_out/hwq_sweep*.txt, twice each): 12 / 13 queues give 25.7 - 25.9 ms per cDNA
# step, 19.0 - 19.3 Iso-Seq, 35.5 - 35.6 dRNA; 16 (round 2's choice, from the days of 8 streams per context) 26.3 / 20.5 /
# 36.0 - with more queues than a context has streams, which streams of which contexts share a queue is HIP's choice, and
# in the Iso-Seq traceback three of ten launches then started when a neighbour had ended; 18 - 23 land between the two.
# The end-to-end pipeline (two contexts busy at once) prefers more: 9.1 M reads/s at 12 - 13, 9.4 at 16, 9.7 at 20.
# 24 and more are more than the hardware schedules at once: every step takes 1.7 x as long.
DEFAULT = 13


def _parse(raw):
    try:
        n = int(raw)
    except (TypeError, ValueError):
        return None
    return min(n, POOL_MAX) if n > 0 else None


def hw_queues(default=DEFAULT):
    """ULTRA_HW_QUEUES parsed and clamped to 1..32 (anything unparsable falls back to HIP's own default), else `default`.
    A launcher that initialises HIP before this process can (rocprofv3 does) must export BOTH GPU_MAX_HW_QUEUES and
    ULTRA_HW_QUEUES with the same value (scripts/profile_round.sh does): the first is what HIP reads, the second keeps
    this module - and through it the library's lane count and the bench line - on the count that is in force."""
    raw = os.environ.get("ULTRA_HW_QUEUES")
    if raw is None:
        return default
    return _parse(raw) or HIP_DEFAULT


GPU_BUDGET = 22   # queues of all processes on one GPU together: from 24 on every launch takes 1.7 x as long


def export_hw_queues(default=DEFAULT, processes_per_gpu=1):
    """Sets GPU_MAX_HW_QUEUES for this process (call before the first HIP call) and returns the value in force.
    processes_per_gpu: how many such processes share the card (pc.main's consumers_per_gpu): they share GPU_BUDGET."""
    n = hw_queues(default)
    if processes_per_gpu > 1 and os.environ.get("ULTRA_HW_QUEUES") is None:
        n = max(HIP_DEFAULT, min(n, GPU_BUDGET // processes_per_gpu))
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return n
