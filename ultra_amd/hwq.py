"""Hardware queues of the process.

The launches of one pass (five forward buckets, up to nine traceback kernels) are dealt over HIP streams; streams
beyond the process's hardware queues share a queue and serialise.  HIP reads GPU_MAX_HW_QUEUES (default 4) once,
at its first call, so a harness that wants more has to export it before anything touches the GPU.  A caller that
initialised HIP earlier simply runs with what it has: ultra_ctx_create takes one launch lane per queue it finds.
"""
import os

HIP_DEFAULT = 4
POOL_MAX = 32     # the GPU pool refuses more
# Default 16: a context deals its launches over 8 lanes (streams), and the end-to-end pipeline keeps two contexts busy at
# once - with 8 queues their 16 streams share queues pairwise, one context's staging kernels wait behind the other's
# 12 ms forward launches and the pipeline delivers 6.9 - 7.9 M reads/s instead of 8.9 M.  The device-resident step is the
# same at 8 and 16; 24 queues are more than the hardware schedules at once (35 ms per step instead of 24).


def _parse(raw):
    try:
        n = int(raw)
    except (TypeError, ValueError):
        return None
    return min(n, POOL_MAX) if n > 0 else None


def hw_queues(default=16):
    """ULTRA_HW_QUEUES parsed and clamped to 1..32 (anything unparsable falls back to HIP's own default), else `default`.
    A launcher that initialises HIP before this process can (rocprofv3 does) must export BOTH GPU_MAX_HW_QUEUES and
    ULTRA_HW_QUEUES with the same value (scripts/profile_round.sh does): the first is what HIP reads, the second keeps
    this module - and through it the library's lane count and the bench line - on the count that is in force."""
    raw = os.environ.get("ULTRA_HW_QUEUES")
    if raw is None:
        return default
    return _parse(raw) or HIP_DEFAULT


def export_hw_queues(default=16):
    """Sets GPU_MAX_HW_QUEUES for this process (call before the first HIP call) and returns the value in force."""
    n = hw_queues(default)
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return n
