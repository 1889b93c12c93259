#!/bin/bash
# seg_reduce_chunk_slab: role layouts 2 (shipping candidate), 3, 4 and the previous library; rocprofv3 kernel time + step time, one box
# (tools/probe/libtwotower_layoutN.so / libtwotower_prev.so: measurement builds, -DTT_SEG_LAYOUT=N / the commit before; not kept in the tree)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b22; mkdir -p $out
A="--no-extra-legs --no-cpu-baseline --no-h2d"
lib=jodalrob-twotower_amd/libtwotower_hip.so
cp $lib /tmp/lib_l2.so
for v in ${VARIANTS:-l2 layout3 layout4 prev l2 prev}; do
  if [ $v = l2 ]; then cp /tmp/lib_l2.so $lib; else cp tools/probe/libtwotower_$v.so $lib; fi
  timeout -k 10 300 python bench.py $A > $out/${v}_$RANDOM.json 2> $out/$v.err || { echo "bench $v failed"; tail -3 $out/$v.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -o k -- python bench.py $A --steps 50 > /dev/null 2> $out/prof_$v.err
  echo "$v: $(grep seg_reduce_chunk_slab $out/prof_$v/k_kernel_stats.csv | head -1 | awk -F, '{print $(NF-5), $(NF-4), $(NF-3)}') | $(grep -c . $out/prof_$v/k_kernel_stats.csv) kernels"
  python tools/kstats.py $out/prof_$v/k_kernel_stats.csv 2>/dev/null | grep "seg_reduce"
done
cp /tmp/lib_l2.so $lib
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_b22/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms loss", d["final_loss"])
P
