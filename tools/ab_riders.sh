#!/bin/bash
# A/B on ONE box: riders hosted by the tail launches vs launches of their own, alternating, device-median step time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for f in "" "--no-riders"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-h2d --steps 200 --warmup 30 $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('riders' if '$f'=='' else 'own launches', d['ms_per_step'], d.get('device_ms_median'), d.get('launches_per_step'))"
  done
done
