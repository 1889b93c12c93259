#!/bin/bash
# row-range device error; segmented 2-rank rehearsal (weak + strong + one_gpu legs); default bench A/B of the row check's cost
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b12; mkdir -p $out
fault() { grep -l "Memory access fault" $out/*.err $out/*.txt 2>/dev/null; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_next_rows.py -m gpu -q -k "outside_the_table or lookup or ingest or segmented or chained" > $out/pytest.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest.txt
if [ -n "$(fault)" ]; then echo "FAULT in tests"; exit 1; fi
timeout -k 10 500 python bench.py --gpus 2 --steps 20 --warmup 5 --rows-notice 2000000 --rows-company 1000000 --no-cpu-baseline --dist-segmented > $out/bench_2rank_segmented.json 2> $out/bench_2rank_segmented.err; echo "2-rank segmented rc $?"
grep -v "^frame #" $out/bench_2rank_segmented.err | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | cut -c1-300 | tail -8
if [ -n "$(fault)" ]; then echo "FAULT in rehearsal"; exit 1; fi
A="--no-extra-legs --no-cpu-baseline --no-h2d"
timeout -k 10 300 python bench.py $A > $out/bench_a.json 2> $out/bench_a.err; echo "bench a rc $?"
timeout -k 10 300 python bench.py $A > $out/bench_b.json 2> $out/bench_b.err; echo "bench b rc $?"
python - <<'P'
import json
for f in ("bench_2rank_segmented","bench_a","bench_b"):
    try:
        d=json.loads(open(f"gpurun_out/r04_b12/{f}.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
        print(f, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms | roofline", round(r.get("frac") or 0,3), r.get("mean_launch_us"), r.get("mean_body_us"), "|", d["config"].get("launch"))
        for k in d:
            if k.startswith("value_global_batch") or k.startswith("ms_per_step_global"): print("   ", k, d[k])
        if "strong_scaling" in d: print("    strong:", d["strong_scaling"]["config"].get("launch"), d["strong_scaling"]["ms_per_step"])
        if "one_gpu_same_tables" in d: print("    one_gpu:", d["one_gpu_same_tables"].get("ms_per_step"), d["one_gpu_same_tables"].get("error"))
    except Exception as e: print(f, "ERR", e)
P
