#!/bin/bash
# round 4, batch 3: the whole GPU suite on the fused hand-over + lookup path, the default bench line (with the two new legs), the
# A/B against round 3's separate launches, kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b3; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -4 $out/pytest_gpu.txt
grep -h "step vs" $out/pytest_gpu.txt > $out/step_reports.txt
timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"
timeout -k 10 300 python bench.py --separate-lookup --no-extra-legs --no-cpu-baseline --no-h2d > $out/bench_separate.json 2> $out/bench_separate.err; echo "bench separate rc $?"
timeout -k 10 300 python bench.py --no-extra-legs --no-cpu-baseline --no-h2d > $out/bench_fused2.json 2> $out/bench_fused2.err; echo "bench fused2 rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py --no-cpu-baseline --no-h2d --no-extra-legs --steps 50 > $out/bench_rocprof.json 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt || true
python - <<'P'
import json
for f in ("bench.json","bench_separate.json","bench_fused2.json","bench_rocprof.json"):
    try:
        d=json.loads(open(f"gpurun_out/r04_b3/{f}").read().strip().splitlines()[-1])
        r=d["roofline"]
        print(f, round(d["value"]/1e6,3), "M pairs/s", round(d["ms_per_step"],5), "ms | roofline", r.get("frac"), r.get("mean_launch_us"), r.get("mean_body_us"), (r.get("lookup_phase") or {}).get("mean_us"), "| launches", d["config"].get("launches_per_step"))
        for k in ("roofline_hbm_resident","configs4"):
            if k in d: print("   ", k, json.dumps(d[k])[:1500])
    except Exception as e:
        print(f, "ERR", e)
P
head -20 $out/kstats.txt
