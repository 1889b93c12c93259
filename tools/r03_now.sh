#!/bin/bash
# configs[1]: bench line + kernel stats of the current build
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_now; mkdir -p $out
timeout -k 10 300 python bench.py --no-cpu-baseline --no-h2d --steps 100 --warmup 20 > $out/bench.json 2> $out/bench.err
python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('configs[1]', d['value'], d['ms_per_step'], d['config'].get('launches_per_step'), d['roofline']['frac'])"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py --no-cpu-baseline --no-h2d --steps 50 > $out/bench_rocprof.json 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt
cat $out/kstats.txt
