#!/bin/bash
# whole GPU suite + two default-shape bench runs + kernel stats on the library as committed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b23; mkdir -p $out
A="--no-extra-legs --no-cpu-baseline --no-h2d"
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $out/pytest_gpu.txt
if grep -q "Memory access fault" $out/*.txt; then echo FAULT; exit 1; fi
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $out/pytest_gpu.txt | head -20; exit 1; }
for i in 1 2; do timeout -k 10 300 python bench.py $A > $out/bench_$i.json 2> $out/bench_$i.err || exit 1; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py $A --steps 50 > /dev/null 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt; head -14 $out/kstats.txt
python - <<'P'
import json
for i in (1, 2):
    d=json.loads(open(f"gpurun_out/r04_b23/bench_{i}.json").read().strip().splitlines()[-1])
    print(i, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms loss", d["final_loss"])
P
