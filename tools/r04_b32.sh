#!/bin/bash
# kernel stats of the one-GPU step on 100 M + 10 M-row tables (every row from HBM) beside the 1 M + 1 M-row step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b32; mkdir -p $out
A="--no-extra-legs --no-cpu-baseline --no-h2d"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_100m -o k -- python bench.py $A --rows-notice 100000000 --rows-company 10000000 --steps 300 --warmup 50 > $out/bench_100m.json 2> $out/prof_100m.err; echo "rc $?"
python tools/kstats.py $out/prof_100m/k_kernel_stats.csv > $out/kstats_100m.txt; head -16 $out/kstats_100m.txt
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.txt 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 $out/pytest_gpu.txt
if grep -q "Memory access fault" $out/*.txt $out/*.err; then echo FAULT; exit 1; fi
for i in 1 2; do timeout -k 10 300 python bench.py $A > $out/bench_1m_$i.json 2> $out/bench_1m.err; done
python - <<'P'
import json
for f in ("bench_100m", "bench_1m_1", "bench_1m_2"):
    d=json.loads(open(f"gpurun_out/r04_b32/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],5), "ms loss", d["final_loss"])
P
