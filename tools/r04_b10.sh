#!/bin/bash
# graph-node argument probe; segmented capture with two real ranks (test + bench rehearsal)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b10; mkdir -p $out
timeout -k 10 120 tools/probe/graph_setparams 200 > $out/graph_setparams.txt 2>&1; echo "probe rc $?"; cat $out/graph_setparams.txt
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "segmented_capture_with_real_peers or two_processes_equal" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -15 $out/pytest.txt
cat gpurun_out/dist_world2_segmented.log 2>/dev/null | tail -60
timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 --rows-notice 2000000 --rows-company 1000000 --no-cpu-baseline --dist-segmented > $out/bench_2rank_segmented.json 2> $out/bench_2rank_segmented.err; echo "2-rank segmented rc $?"
tail -5 $out/bench_2rank_segmented.err | cut -c1-400
python - <<'P'
import json
try:
    d=json.loads(open("gpurun_out/r04_b10/bench_2rank_segmented.json").read().strip().splitlines()[-1])
    print(round(d["value"]/1e6,3), "M", d["ms_per_step"], d["config"].get("launch"), "|", d["config"]["parallelism"][-160:])
except Exception as e: print("ERR", e)
P
