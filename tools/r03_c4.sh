#!/bin/bash
# configs[4] (B = 65536, D = 256, fp8 score, fused sparse Adam): bench line + kernel stats
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_c4; mkdir -p $out
timeout -k 10 400 python bench.py --batch 65536 --final-dim 256 --score-dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $out/bench.json 2> $out/bench.err
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py --batch 65536 --final-dim 256 --score-dtype fp8 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d > $out/bench_rocprof.json 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt
cat $out/kstats.txt
