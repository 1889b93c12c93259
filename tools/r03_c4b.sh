#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_c4; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "score or configs4 or rounded" > $out/tests.log 2>&1 || (tail -30 $out/tests.log; exit 1)
tail -2 $out/tests.log
timeout -k 10 400 python bench.py --batch 65536 --final-dim 256 --score-dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $out/bench_fp8.json 2> $out/bench.err
python -c "
import json; d=json.loads(open('$out/bench_fp8.json').read().strip().splitlines()[-1]); print('fp8', d['value'], d['ms_per_step'], d['mfma']['us_per_iteration'])"
timeout -k 10 400 python bench.py --batch 65536 --final-dim 256 --score-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d > $out/bench_bf16.json 2> $out/bench.err
python -c "
import json; d=json.loads(open('$out/bench_bf16.json').read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'], d['mfma']['us_per_iteration'])"
