#!/bin/bash
# segmented capture with two real ranks: the test (empty segments kept alive), then the bench form traced call by call
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b11; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "segmented_capture_with_real_peers" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -5 $out/pytest.txt
grep -v "frame #" gpurun_out/dist_world2_segmented.log 2>/dev/null | grep "rank\|Error\|error\|assert" | cut -c1-300 | tail -20
TT_SYNC_DEBUG=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --rows-notice 2000000 --rows-company 1000000 --no-cpu-baseline --dist-segmented > $out/bench_traced.json 2> $out/bench_traced.err; echo "traced 2-rank segmented rc $?"
grep -v "^frame #" $out/bench_traced.err | grep -v "done$" | cut -c1-300 | tail -30
echo "---- last traced calls"; grep "\[tt\]" $out/bench_traced.err | tail -12 | cut -c1-200
