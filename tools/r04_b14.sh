#!/bin/bash
# row check moved to the hand-over launch: A/B against the unchecked build on one box; whole GPU suite
# (tools/probe/libtwotower_nocheck.so: see tools/r04_b13.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b14; mkdir -p $out
A="--no-extra-legs --no-cpu-baseline --no-h2d"
lib=jodalrob-twotower_amd/libtwotower_hip.so
cp $lib /tmp/lib_check.so
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.txt
if grep -q "Memory access fault" $out/*.txt; then echo FAULT; exit 1; fi
for i in 1 2 3; do
  cp /tmp/lib_check.so $lib; timeout -k 10 300 python bench.py $A > $out/check_$i.json 2> $out/check_$i.err || exit 1
  cp tools/probe/libtwotower_nocheck.so $lib; timeout -k 10 300 python bench.py $A > $out/nocheck_$i.json 2> $out/nocheck_$i.err || exit 1
done
cp /tmp/lib_check.so $lib
python - <<'P'
import json
for f in [f"{k}_{i}" for i in (1,2,3) for k in ("check","nocheck")]:
    d=json.loads(open(f"gpurun_out/r04_b14/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms | frac", round(r["frac"],3), "launch", round(r["mean_launch_us"],2), "body", round(r["mean_body_us"],2))
P
