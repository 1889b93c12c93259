#!/bin/bash
# what the lookup's row check costs (same box, alternating libraries); whole GPU suite on the new library; kernel stats of the sharded world-1 step
# (tools/probe/libtwotower_nocheck.so = the library linked with tt_embed.hip compiled -DTT_NO_ROW_CHECK: a measurement build made for this
#  batch and not kept in the tree; rebuild it as jodalrob-twotower_amd/build.py does, one object replaced)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b13; mkdir -p $out
A="--no-extra-legs --no-cpu-baseline --no-h2d"
lib=jodalrob-twotower_amd/libtwotower_hip.so
cp $lib /tmp/lib_check.so
for i in 1 2 3; do
  cp /tmp/lib_check.so $lib; timeout -k 10 300 python bench.py $A > $out/check_$i.json 2> $out/check_$i.err || exit 1
  cp tools/probe/libtwotower_nocheck.so $lib; timeout -k 10 300 python bench.py $A > $out/nocheck_$i.json 2> $out/nocheck_$i.err || exit 1
done
cp /tmp/lib_check.so $lib
python - <<'P'
import json
for f in [f"{k}_{i}" for i in (1,2,3) for k in ("check","nocheck")]:
    d=json.loads(open(f"gpurun_out/r04_b13/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms | frac", round(r["frac"],3), "launch", round(r["mean_launch_us"],2), "body", round(r["mean_body_us"],2))
P
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.txt
if grep -q "Memory access fault" $out/*.txt $out/*.err; then echo FAULT; exit 1; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_sh -o k -- python bench.py $A --force-dist --steps 50 > $out/bench_sharded_under_rocprof.json 2> $out/prof_sh.err; echo "rocprof sharded rc $?"
python tools/kstats.py $out/prof_sh/k_kernel_stats.csv > $out/kstats_sharded.txt || true
head -40 $out/kstats_sharded.txt
