#!/bin/bash
# sharded step at world 1 (RCCL, --force-dist): bench line + kernel stats, configs[1] tables
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_shard; mkdir -p $out
timeout -k 10 300 python bench.py --force-dist --no-cpu-baseline --no-h2d --steps 100 --warmup 20 > $out/bench.json 2> $out/bench.err || (tail -20 $out/bench.err; exit 1)
python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print('sharded world 1', d['value'], d['ms_per_step'], d['config'].get('launches_per_step'))"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py --force-dist --no-cpu-baseline --no-h2d --steps 50 > $out/bench_rocprof.json 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt
cat $out/kstats.txt
