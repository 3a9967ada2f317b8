"""Development aid: the parts of gpurun_out/bench_x.log's line that the experiments of round 3 looked at."""
import json
j = json.loads(open("gpurun_out/bench_x.log").read().strip().splitlines()[-1])
print(round(j["ms_per_step"], 2), round(j["kernels_only"]["ms_per_step"], 2), {k[:9]: round(v, 2) for k, v in j["stage_ms_per_step"].items()})
if j.get("pipelined_leg"):
    print("pipelined", round(j["pipelined_leg"]["ms_per_step"], 2), {k: round(v, 2) for k, v in j["pipelined_leg"]["host_ms_per_step"].items()})
if j.get("end_to_end"):
    print("e2e", round(j["end_to_end"]["reads_per_s"]))
if j.get("front_end"):
    f = j["front_end"]
    print("front", round(f["align_single_front_reads_per_s"]), round(f["two_workers_reads_per_s"]), {k: round(v, 1) for k, v in f["ms_per_batch"].items()})
