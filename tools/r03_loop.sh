#!/bin/bash
# the real training loop on the fast path: bench line with the device-store leg, then an epoch of scripts/train.py on configs[0]'s
# sizes (10 k x 10 k entities, 100 k pairs, batch 256) eager and --fast, and --fast at batch 8192 on 1 M pairs
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_loop; mkdir -p $out
timeout -k 10 400 python bench.py --no-cpu-baseline --steps 100 --warmup 20 > $out/bench.json 2> $out/bench.err
echo "bench done"
timeout -k 10 300 python scripts/train.py --entities 10000 --pairs 100000 --batch-size 256 --output-dir $out/m_eager > $out/train_eager.log 2>&1
echo "eager done"; grep "throughput\|done:" $out/train_eager.log
timeout -k 10 300 python scripts/train.py --entities 10000 --pairs 100000 --batch-size 256 --output-dir $out/m_fast --fast > $out/train_fast.log 2>&1
echo "fast done"; grep "throughput\|done:" $out/train_fast.log
timeout -k 10 300 python scripts/train.py --entities 100000 --pairs 2000000 --batch-size 8192 --output-dir $out/m_fast8k --fast > $out/train_fast8k.log 2>&1
echo "fast 8k done"; grep "throughput\|done:" $out/train_fast8k.log
rm -rf $out/m_eager $out/m_fast $out/m_fast8k
