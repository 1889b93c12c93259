#!/bin/bash
# U training steps per graph launch (unrolled.UnrolledTrainStep): bit-identity tests, then the bench at U = 1, 2, 4, 8 on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b17; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_next_rows.py -m gpu -q -x -k "unrolled or step_from_store_equals or graphed_step_equals or graph_ingest" > $out/pytest.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -25 $out/pytest.txt | cut -c1-300
if grep -q "Memory access fault" $out/*.txt; then echo FAULT; exit 1; fi
[ $rc -ne 0 ] && exit 1
A="--no-extra-legs --no-cpu-baseline --no-h2d"
for u in 1 2 4 8 1 4; do
  timeout -k 10 300 python bench.py $A --unroll $u > $out/u${u}_$RANDOM.json 2> $out/u$u.err || { echo "bench unroll $u failed"; tail -5 $out/u$u.err; exit 1; }
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_b17/u*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f.split("/")[-1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms | dev median", round(d["device_ms_per_step_median"],5), "| host", round(d["host_enqueue_ms_per_step"],4), "| store", round((d.get("ms_per_step_with_device_store") or 0),5), "|", d["config"]["launch"], d["config"].get("launches_per_step"), "loss", round(d["final_loss"],4))
P
