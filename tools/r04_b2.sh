#!/bin/bash
# round 4, batch 2: lab v2, the changed tests, the real loop at B = 8192 over three epochs
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b2; mkdir -p $out
timeout -k 10 300 tools/probe/lookup_lab 1000000 1000000 8192 64 0 > $out/lab_1m.txt 2>&1; echo "lab 1m rc $?"
timeout -k 10 300 tools/probe/lookup_lab 100000000 10000000 8192 64 0 > $out/lab_100m.txt 2>&1; echo "lab 100m rc $?"
timeout -k 10 900 python -m pytest tests/test_gpu_next_rows.py -x -q > $out/pytest_next_rows.txt 2>&1; echo "next_rows rc $?"; tail -5 $out/pytest_next_rows.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "bf16_step_vs_rounded_oracle or graphed_step_equals_eager or graph_ingest or adam_trajectory or full_size_step_is_reproducible" > $out/pytest_parity_subset.txt 2>&1; echo "parity subset rc $?"; tail -5 $out/pytest_parity_subset.txt
grep -h "step vs" $out/pytest_parity_subset.txt > $out/step_reports.txt
cd /tmp && timeout -k 10 600 python $GRAFT_REPO_ROOT/scripts/train.py --entities 100000 --pairs 2000000 --batch-size 8192 --fast --epochs 3 --output-dir /tmp/tt_models --results-csv /tmp/tt_results.csv > $GRAFT_REPO_ROOT/$out/train_fast_b8192.txt 2>&1; echo "train rc $?"
grep -E "throughput|Train -|Val " $GRAFT_REPO_ROOT/$out/train_fast_b8192.txt
