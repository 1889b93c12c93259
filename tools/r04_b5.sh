#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b5; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -6 $out/pytest_gpu.txt
grep -h "step vs" $out/pytest_gpu.txt > $out/step_reports.txt
A="--no-extra-legs --no-cpu-baseline --no-h2d"
for v in rows ids fused rows_b ids_b fused_b; do
  case $v in rows*) X="";; ids*) X="--lookup-from-ids";; fused*) X="--fused-handover";; esac
  timeout -k 10 300 python bench.py $A $X > $out/bench_$v.json 2> $out/bench_$v.err; echo "bench $v rc $?"
done
timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
python - <<'P'
import json
for v in ("rows","ids","fused","rows_b","ids_b","fused_b",""):
    f = f"gpurun_out/r04_b5/bench_{v}.json" if v else "gpurun_out/r04_b5/bench.json"
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(v or "default", round(d["ms_per_step"],5), "ms", round(d["device_ms_per_step_median"],5), "| roofline", round(r.get("frac") or 0,3), r.get("mean_launch_us"), r.get("mean_body_us"), r.get("dispatch_overhead_us"), "| launches", d["config"].get("launches_per_step"))
        for k in ("roofline_hbm_resident","configs4"):
            if k in d: print("   ", k, json.dumps(d[k])[:2200])
    except Exception as e: print(v, "ERR", e)
P
