#!/bin/bash
# round 4: tests incl. the segmented capture, the default bench line, kernel stats, lookup traffic counters, reference-semantics lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_prof; mkdir -p $out
timeout -k 10 600 python tests/_dist_world1_worker.py > $out/world1_worker.txt 2>&1; echo "world1 worker rc $?"; grep -E "segmented|DIST_WORLD1_OK|Error|assert" $out/world1_worker.txt | tail -8
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest_gpu.txt
timeout -k 10 600 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc $?"
A="--no-extra-legs --no-cpu-baseline --no-h2d"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py $A --steps 50 > $out/bench_n1_under_rocprof.json 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt || true
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $out/fetch -o p -- python bench.py $A --steps 10 --warmup 3 --no-lookup-profile > $out/f.json 2> $out/f.err; echo "fetch rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $out/write -o p -- python bench.py $A --steps 10 --warmup 3 --no-lookup-profile > $out/w.json 2> $out/w.err; echo "write rc $?"
python - <<'P'
import csv, collections, json
res = {}
for name in ("fetch", "write"):
    rows = list(csv.DictReader(open(f"gpurun_out/r04_prof/{name}/p_counter_collection.csv")))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"]
        key = "lookup_wave_kernel" if "lookup_wave" in k else ("batch_ingest_kernel" if "batch_ingest_kernel" in k else None)
        if key: agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            res.setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
json.dump(res, open("gpurun_out/r04_prof/lookup_pmc_raw.json", "w"), indent=1)
print(json.dumps(res))
P
timeout -k 10 300 python bench.py $A --optimizer fused_dense > $out/bench_fused_dense.json 2> $out/bench_fused_dense.err; echo "fused_dense rc $?"
timeout -k 10 300 python bench.py $A --score-dtype fp32 > $out/bench_score_fp32.json 2> $out/bench_score_fp32.err; echo "score fp32 rc $?"
timeout -k 10 300 python bench.py $A --force-dist > $out/bench_sharded_world1.json 2> $out/bench_sharded_world1.err; echo "sharded rc $?"
timeout -k 10 300 python bench.py $A --force-dist --dist-segmented > $out/bench_sharded_world1_segmented.json 2> $out/bench_sharded_world1_segmented.err; echo "sharded segmented rc $?"
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name|^\s*Name" | grep -i -E "SQ_.*(MFMA|VALU)" > $out/sq_counters_available.txt
python - <<'P'
import json
for f in ("bench_n1","bench_n1_under_rocprof","bench_fused_dense","bench_score_fp32","bench_sharded_world1","bench_sharded_world1_segmented"):
    try:
        d=json.loads(open(f"gpurun_out/r04_prof/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
        print(f, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms | roofline", round(r.get("frac") or 0,3), r.get("mean_launch_us"), "| launch", d["config"].get("launch"), d["config"].get("launches_per_step"))
    except Exception as e: print(f, "ERR", e)
P
head -16 $out/kstats.txt
