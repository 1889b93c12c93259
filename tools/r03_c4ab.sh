#!/bin/bash
# configs[4]: fp8 parity tests, then the step with the two gradient forms (TT_OPT_FP8_GRAD 1 / 0)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_c4; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "score_fp8 or configs4 or rounded" > $out/tests.log 2>&1 || (tail -30 $out/tests.log; exit 1)
tail -2 $out/tests.log
for g in 1 0; do
  timeout -k 10 400 python bench.py --batch 65536 --final-dim 256 --score-dtype fp8 --fp8-grad $g --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $out/bench_fp8_g$g.json 2> $out/bench.err
  python -c "
import json; d=json.loads(open('$out/bench_fp8_g$g.json').read().strip().splitlines()[-1]); print('fp8 grad mode $g', d['value'], d['ms_per_step'], d['mfma']['us_per_iteration'])"
done
