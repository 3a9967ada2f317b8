"""The committed bench line (profiles/r02_bench_default.json, a copy of what `python bench.py` printed on the MI355X) keeps
the contract the driver and the judge read: one JSON object with the headline fields, a roofline object whose
fraction is achieved / peak, a CPU baseline with its sample stated, and no product number taken from the oracle."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as fh:
        rows = [l for l in fh.read().splitlines() if l.startswith("{")]
    assert len(rows) == 1
    return json.loads(rows[0])


def test_default_bench_line_keeps_the_contract():
    d = _line("r02_bench_default.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["unit"] == "reads/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    reads = d["config"]["reads_per_gpu_per_step"]
    assert abs(d["value"] - reads / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma", "valu_int32") and r["unit"] and r["traffic"] is not None
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic operations of a launch / its measured duration
    assert abs(r["achieved"] - r["cells_per_launch"] * r["ops_per_cell"] / (r["avg_launch_ms"] * 1e-3) * 1e-12) < 1e-6 * r["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["sample"] and c["unit"] == "reads/s"
    assert d["parity_checked_instances"] > 0


def test_rehearsed_two_rank_line_says_what_it_is():
    d = _line("r02_bench_2_ranks_rehearsal.json")
    assert d["n_gpus"] == 2 and "rehearsal" in d["config"]
