#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b8; mkdir -p $out
timeout -k 10 300 python tools/r04_tail_stamps.py > $out/tail_stamps.txt 2>&1; echo "stamps rc $?"; grep -E "tail_fwd|phase|workgroup" $out/tail_stamps.txt
A="--no-extra-legs --no-cpu-baseline --no-h2d"
for v in a b; do
  timeout -k 10 300 python bench.py $A > $out/bench_$v.json 2> $out/bench_$v.err; echo "bench $v rc $?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py $A --steps 50 > $out/bench_rocprof.json 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt || true
python - <<'P'
import json
for v in ("a","b","rocprof"):
    try:
        d=json.loads(open(f"gpurun_out/r04_b8/bench_{v}.json").read().strip().splitlines()[-1])
        print(v, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms", round(d["device_ms_per_step_median"],5))
    except Exception as e: print(v, "ERR", e)
P
head -14 $out/kstats.txt
