#!/bin/bash
# usage: ab_prof.sh <tag> [bench args...]  -- rocprofv3 kernel stats + the bench line of one configuration
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o p -- python bench.py --no-cpu-baseline --no-h2d --steps 50 "$@" > gpurun_out/$tag.log 2>&1 || exit 1
python tools/kstats.py gpurun_out/$tag/p_kernel_stats.csv
timeout -k 10 300 python bench.py --no-cpu-baseline --no-h2d "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
