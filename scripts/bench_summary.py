import json,sys
d=json.load(open(sys.argv[1]))
print("reads/s %.0f  ms/step %.2f"%(d["value"], d["ms_per_step"]))
print({k: round(v,3) for k,v in d["stage_ms_per_step"].items()})
print("fwd GCUPS %.0f all %.0f"%(d["myers_gcups_forward"], d["myers_gcups_all_passes"]), d["roofline"]["kernel"], "frac %.4f"%d["roofline"]["frac"])
if d.get("cpu_baseline"): print("cpu", round(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"])
for k, v in d.get("forward_buckets", {}).items(): print("  ", k, "cells %.3g ms %.3f TCUPS %.1f" % (v["cells_per_step"], v["ms_per_step"], v["tcups"] or 0))
