#!/bin/bash
# the default bench run (new default: 2000 steps behind 200) with its wall time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b26; mkdir -p $out
t0=$SECONDS
timeout -k 10 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "rc $? wall $((SECONDS - t0)) s"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r04_b26/bench_default.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["steps"], d["warmup"], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms | roofline", round(r["frac"],3), round(r["mean_launch_us"],2), round(r["mean_body_us"],2), r["launches_timed"], "| h2d", d.get("ms_per_step_with_h2d"), "| store", d.get("ms_per_step_with_device_store"), "| hbm", d["roofline_hbm_resident"]["lookup_rows"]["frac"], "| c4", d["configs4"]["value"], "| cpu", d["cpu_baseline"]["value"])
P
