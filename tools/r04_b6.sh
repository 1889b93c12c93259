#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b6; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -s > $out/pytest_gpu.txt 2>&1; echo "pytest rc $?"; tail -6 $out/pytest_gpu.txt
grep -h "trajectory A/B" $out/pytest_gpu.txt > $out/fp8_traj.txt; cat $out/fp8_traj.txt
A="--no-extra-legs --no-cpu-baseline --no-h2d"
for v in a b c; do
  timeout -k 10 300 python bench.py $A > $out/bench_$v.json 2> $out/bench_$v.err; echo "bench $v rc $?"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py $A --steps 50 > $out/bench_rocprof.json 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt || true
timeout -k 10 400 python bench.py --batch 65536 --final-dim 256 --score-dtype fp8 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-extra-legs > $out/bench_c4.json 2> $out/c4.err; echo "c4 rc $?"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA -d $out/pmc -o p -- python bench.py $A --steps 20 --warmup 5 --no-lookup-profile > $out/pmc.json 2> $out/pmc.err; echo "pmc rc $?"
python tools/pmc_summary.py $out/pmc_summary.json $out/pmc > $out/pmc_summary.txt 2>&1 || true
python - <<'P'
import json
for v in ("a","b","c","rocprof","c4"):
    try:
        d=json.loads(open(f"gpurun_out/r04_b6/bench_{v}.json").read().strip().splitlines()[-1]); r=d["roofline"]
        print(v, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms", round(d["device_ms_per_step_median"],5), "| roofline", round(r.get("frac") or 0,3), r.get("mean_launch_us"), "| launches", d["config"].get("launches_per_step"), (d.get("mfma") or {}).get("frac"))
    except Exception as e: print(v, "ERR", e)
s=json.load(open("gpurun_out/r04_b6/pmc_summary.json"))
for k,v in s.items():
    if "score" in k: print(k, v)
P
head -14 $out/kstats.txt
