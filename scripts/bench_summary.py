"""Print the interesting numbers of a bench.py JSON line (test infrastructure).  usage: bench_summary.py file"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = lambda v: round(v, 3) if isinstance(v, float) else v
print("headline", {k: r(d[k]) for k in ("value", "ms_per_step", "whole_path_hbm_frac")}, {k: r(v) for k, v in d["kernels_ms"].items()},
      "roofline", d["roofline"]["bound"], r(d["roofline"]["frac"]), "traffic x", r(d["roofline"].get("traffic_over_algorithmic")))
if "cpu_baseline" in d:
    print("cpu", r(d["cpu_baseline"]["value"]), "C port", r(d["cpu_baseline_c"]["value"]), "on", d["cpu_baseline_c"]["cores"], "cores")
keys = ("songs", "forward_ms", "backtrace_ms", "ms_per_step", "Mframes_per_s", "Mframes_per_s_best", "whole_path_hbm_frac", "forward_hbm_frac",
        "overlapped_ms_per_step", "workspace_GB", "equals_normal_decode", "vs_normal_decode", "bit_exact_vs_oracle_sample", "vs_uniform_B2048_one_stream", "error")
for blk in ("sweep", "configs4"):
    for k, v in d.get(blk, {}).items():
        if isinstance(v, dict):
            print(blk, k, {kk: r(v[kk]) for kk in keys if kk in v})
for k, v in d.get("pipeline", {}).items():
    if isinstance(v, dict):
        print("pipeline", k, {kk: r(vv) for kk, vv in v.items() if kk != "builder"}, "builder", r(v["builder"]["Mframes_per_s"]), r(v["builder"]["roofline"]["frac"]))
