#!/bin/bash
# configs[4] (B = 65536, D = 256, fp8 score operands): kernel stats of the step on the final library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b33; mkdir -p $out
A="--no-extra-legs --no-cpu-baseline --no-h2d --no-lookup-profile --batch 65536 --final-dim 256 --score-dtype fp8 --pool 2"
timeout -k 10 400 python bench.py $A --steps 40 --warmup 10 > $out/bench_c4.json 2> $out/bench_c4.err; echo "rc $?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py $A --steps 20 --warmup 5 > /dev/null 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt; head -40 $out/kstats.txt
python - <<'P'
import json
d=json.loads(open("gpurun_out/r04_b33/bench_c4.json").read().strip().splitlines()[-1])
print(round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), "ms", d["config"].get("launches_per_step"))
P
