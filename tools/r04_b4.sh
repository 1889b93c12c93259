#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b4; mkdir -p $out
timeout -k 10 600 python tests/debug/graph_vs_eager_bisect.py > $out/bisect.txt 2>&1; echo "bisect rc $?"; cat $out/bisect.txt | tail -30
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -8 $out/pytest_gpu.txt
grep -h "step vs" $out/pytest_gpu.txt > $out/step_reports.txt
A="--no-extra-legs --no-cpu-baseline --no-h2d"
for v in fused nt separate fused_b nt_b separate_b; do
  case $v in fused*) X="";; nt*) X="--lookup-nt";; separate*) X="--separate-lookup";; esac
  timeout -k 10 300 python bench.py $A $X > $out/bench_$v.json 2> $out/bench_$v.err; echo "bench $v rc $?"
done
python - <<'P'
import json
for v in ("fused","nt","separate","fused_b","nt_b","separate_b"):
    try:
        d=json.loads(open(f"gpurun_out/r04_b4/bench_{v}.json").read().strip().splitlines()[-1]); r=d["roofline"]
        print(v, round(d["ms_per_step"],5), "ms", round(d["device_ms_per_step_median"],5), "| roofline", round(r.get("frac") or 0,3), r.get("mean_launch_us"), r.get("mean_body_us"), (r.get("lookup_phase") or {}).get("mean_us"))
    except Exception as e: print(v, "ERR", e)
P
rocprofv3 -L 2>/dev/null | grep -i -E "COEXEC|VALU_MFMA|MFMA_BUSY" | head -10 > $out/counters.txt; cat $out/counters.txt
