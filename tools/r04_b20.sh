#!/bin/bash
# seg_reduce_chunk_slab: new role order / fat slab blocks / one block per projection-bias item against the previous library, same box
# (tools/probe/libtwotower_prev.so = the library of the commit before: a measurement copy, not kept in the tree)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b20; mkdir -p $out
A="--no-extra-legs --no-cpu-baseline --no-h2d"
lib=jodalrob-twotower_amd/libtwotower_hip.so
cp $lib /tmp/lib_new.so
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $out/pytest_gpu.txt
if grep -q "Memory access fault" $out/*.txt; then echo FAULT; exit 1; fi
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $out/pytest_gpu.txt | head -20; exit 1; }
for i in 1 2 3; do
  cp /tmp/lib_new.so $lib; timeout -k 10 300 python bench.py $A > $out/new_$i.json 2> $out/new_$i.err || exit 1
  cp tools/probe/libtwotower_prev.so $lib; timeout -k 10 300 python bench.py $A > $out/prev_$i.json 2> $out/prev_$i.err || exit 1
done
cp /tmp/lib_new.so $lib
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py $A --steps 50 > /dev/null 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv | head -14
python - <<'P'
import json
for f in [f"{k}_{i}" for i in (1,2,3) for k in ("new","prev")]:
    d=json.loads(open(f"gpurun_out/r04_b20/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms loss", d["final_loss"])
P
