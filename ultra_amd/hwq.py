"""Hardware queues of the process.

The launches of one pass (five forward buckets, up to nine traceback kernels) are dealt over HIP streams; streams
beyond the process's hardware queues share a queue and serialise.  HIP reads GPU_MAX_HW_QUEUES (default 4) once,
at its first call, so a harness that wants more has to export it before anything touches the GPU.  A caller that
initialised HIP earlier simply runs with what it has: ultra_ctx_create takes one launch lane per queue it finds.
"""
import os

HIP_DEFAULT = 4
POOL_MAX = 32     # the GPU pool refuses more


def hw_queues(default=8):
    """ULTRA_HW_QUEUES parsed and clamped to 1..32; anything unparsable falls back to HIP's own default."""
    raw = os.environ.get("ULTRA_HW_QUEUES")
    if raw is None:
        return default
    try:
        n = int(raw)
    except ValueError:
        return HIP_DEFAULT
    if n <= 0:
        return HIP_DEFAULT
    return min(n, POOL_MAX)


def export_hw_queues(default=8):
    """Sets GPU_MAX_HW_QUEUES for this process (call before the first HIP call) and returns the value."""
    n = hw_queues(default)
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return n
