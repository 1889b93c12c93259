#!/bin/bash
# does the timed region's length matter?  (clock ramp: 100 steps are 23 ms of GPU time)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b25; mkdir -p $out
A="--no-extra-legs --no-cpu-baseline --no-h2d"
for sw in "100 20" "1000 20" "1000 200" "4000 200" "100 20" "4000 200" "100 1000"; do set -- $sw
  timeout -k 10 300 python bench.py $A --steps $1 --warmup $2 > $out/s$1_w$2_$RANDOM.json 2> $out/err.txt || exit 1
done
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_b25/s*.json"), key=lambda x: x):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["steps"], d["warmup"], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms | dev median", round(d["device_ms_per_step_median"],5))
P
