#!/bin/bash
# round-end measurements on one GPU box: the default bench line, the same under rocprofv3 (kernel stats), train.py's tower widths,
# configs[4] with fp8 score operands.  Outputs under gpurun_out/final/ (copied into profiles/ by hand).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
timeout -k 10 500 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err
echo "bench n1 done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py --no-cpu-baseline --no-h2d > $out/bench_n1_under_rocprof.json 2> $out/prof.err
echo "rocprof done"
timeout -k 10 300 python bench.py --hidden 512,256 --final-dim 128 --no-cpu-baseline > $out/bench_train_py_dims.json 2> $out/train.err
echo "train dims done"
timeout -k 10 400 python bench.py --batch 65536 --final-dim 256 --score-dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > $out/bench_configs4_fp8.json 2> $out/c4.err
echo "configs4 done"
