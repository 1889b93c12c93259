#!/bin/bash
# the lookup on configs[2]-sized tables (100 M + 10 M rows, unsharded on one GPU): bench roofline + FETCH_SIZE / WRITE_SIZE passes
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_lookup100m; mkdir -p $out
A="--rows-notice 100000000 --rows-company 10000000 --no-cpu-baseline --no-h2d"
timeout -k 10 500 python bench.py $A --steps 50 --warmup 10 > $out/bench.json 2> $out/bench.err
python -c "
import json; d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(d['roofline'])"
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $out/fetch -o p -- python bench.py $A --steps 10 --warmup 3 --no-lookup-profile > $out/f.json 2> $out/f.err
echo fetch
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $out/write -o p -- python bench.py $A --steps 10 --warmup 3 --no-lookup-profile > $out/w.json 2> $out/w.err
echo write
python - <<'P'
import csv, collections
for name in ("fetch","write"):
    rows=list(csv.DictReader(open(f"gpurun_out/r03_lookup100m/{name}/p_counter_collection.csv")))
    agg=collections.defaultdict(list)
    for r in rows:
        if "lookup_wave" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(name, k, len(v), sum(v)/len(v))
P
