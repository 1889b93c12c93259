#!/bin/bash
# gemm_back: order of the three roles in the flat grid (012 = shipping), rocprof kernel time + step time on one box
# (tools/probe/libtwotower_orderXYZ.so = the library with tt_gemm_back_batched's role loop reordered to X, Y, Z: measurement builds made for
#  this batch from a three-line patch that was not kept -- profiles/NOTES.md, 'gemm_back: the order of its roles')
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b18; mkdir -p $out
A="--no-extra-legs --no-cpu-baseline --no-h2d"
lib=jodalrob-twotower_amd/libtwotower_hip.so
cp $lib /tmp/lib_012.so
for o in 012 102 120 210 201 012; do
  if [ $o = 012 ]; then cp /tmp/lib_012.so $lib; else cp tools/probe/libtwotower_order$o.so $lib; fi
  timeout -k 10 300 python bench.py $A > $out/o${o}_$RANDOM.json 2> $out/o$o.err || { echo "bench $o failed"; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$o -o k -- python bench.py $A --steps 50 > /dev/null 2> $out/prof_$o.err
  echo "order $o: $(grep gemm_back $out/prof_$o/k_kernel_stats.csv | head -1 | cut -d, -f1-4)"
done
cp /tmp/lib_012.so $lib
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_b18/o*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms loss", d["final_loss"])
P
