#!/bin/bash
# adam_fused: the three row stores as non-temporal stores (measurement build libtwotower_adamnt.so, -DTT_ADAM_NT) at 1 M and 100 M rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b36; mkdir -p $out
A="--no-extra-legs --no-cpu-baseline --no-h2d"
lib=jodalrob-twotower_amd/libtwotower_hip.so
cp $lib /tmp/lib_main.so
for v in main adamnt main adamnt; do
  if [ $v = main ]; then cp /tmp/lib_main.so $lib; else cp tools/probe/libtwotower_$v.so $lib; fi
  for rows in "1000000 1000000" "100000000 10000000"; do set -- $rows
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${v}_$1 -o k -- python bench.py $A --rows-notice $1 --rows-company $2 --steps 200 --warmup 50 > $out/bench_${v}_$1.json 2> $out/err.txt
    echo "$v rows $1: adam $(grep adam_fused $out/prof_${v}_$1/k_kernel_stats.csv | head -1 | awk -F, '{print $(NF-5)}') ns | $(python -c "import json;d=json.loads(open('$out/bench_${v}_$1.json').read().strip().splitlines()[-1]);print(round(d['ms_per_step'],5),'ms loss',d['final_loss'])")"
  done
done
cp /tmp/lib_main.so $lib
