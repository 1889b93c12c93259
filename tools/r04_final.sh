#!/bin/bash
# round-4 end: whole GPU suite, default bench line, kernel stats, multi-GPU bench path rehearsed with two ranks on the one GPU
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_final; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest_gpu.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc $?"; tail -2 $out/smoke.txt
timeout -k 10 600 python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc $?"
A="--no-extra-legs --no-cpu-baseline --no-h2d"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py $A --steps 50 > $out/bench_n1_under_rocprof.json 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt || true
timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 2 --rows-notice 2000000 --rows-company 1000000 --no-cpu-baseline > $out/bench_2rank_rehearsal.json 2> $out/bench_2rank.err; echo "2-rank rc $?"
timeout -k 10 600 python bench.py --gpus 2 --steps 10 --warmup 5 --rows-notice 2000000 --rows-company 1000000 --no-cpu-baseline --dist-segmented > $out/bench_2rank_rehearsal_segmented.json 2> $out/bench_2rank_segmented.err; echo "2-rank segmented rc $?"
if grep -q "Memory access fault" $out/*.err $out/*.txt; then echo "FAULT"; exit 1; fi
timeout -k 10 300 python bench.py $A --rows-notice 100000000 --rows-company 10000000 --steps 50 --warmup 10 > $out/bench_one_gpu_100m_rows.json 2> $out/b100m.err; echo "100m rc $?"
python - <<'P'
import json
for f in ("bench_n1","bench_n1_under_rocprof","bench_2rank_rehearsal","bench_2rank_rehearsal_segmented","bench_one_gpu_100m_rows"):
    try:
        d=json.loads(open(f"gpurun_out/r04_final/{f}.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
        print(f, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms | roofline", round(r.get("frac") or 0,3), r.get("mean_launch_us"), "|", d["config"].get("launch"), d["config"].get("launches_per_step"), d.get("metric","")[:60])
        for k in ("roofline_hbm_resident","configs4"):
            if k in d: print("   ", k, d[k].get("frac"), d[k].get("value"), d[k].get("ms_per_step"))
        for k in d:
            if k.startswith("value_global_batch"): print("   ", k, d[k])
    except Exception as e: print(f, "ERR", e)
P
head -15 $out/kstats.txt
