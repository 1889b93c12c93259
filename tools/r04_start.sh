#!/bin/bash
# round-4 starting point: default bench line, kernel stats of the configs[1] step, the lookup on 100 M + 10 M rows
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_start; mkdir -p $out
timeout -k 10 500 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench done"; tail -c 600 $out/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py --no-cpu-baseline --no-h2d --steps 50 > $out/bench_rocprof.json 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt || true
echo "stats done"
A="--rows-notice 100000000 --rows-company 10000000 --no-cpu-baseline --no-h2d"
timeout -k 10 500 python bench.py $A --steps 50 --warmup 10 > $out/bench100m.json 2> $out/bench100m.err
echo "100m done"
