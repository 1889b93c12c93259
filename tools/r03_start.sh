#!/bin/bash
# round-3 starting point: bench line, kernel stats and two SQ counter passes of the configs[1] step (all kernels)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_start; mkdir -p $out
timeout -k 10 300 python bench.py --no-cpu-baseline --no-h2d --steps 100 --warmup 20 > $out/bench.json 2> $out/bench.err
echo "bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o k -- python bench.py --no-cpu-baseline --no-h2d --steps 50 > $out/bench_rocprof.json 2> $out/prof.err
python tools/kstats.py $out/prof/k_kernel_stats.csv > $out/kstats.txt || true
echo "stats done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d $out/pmc1 -o p -- python bench.py --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $out/pmc1.json 2> $out/pmc1.err
echo "pmc1 done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES -d $out/pmc2 -o p -- python bench.py --no-cpu-baseline --no-h2d --steps 20 --warmup 5 > $out/pmc2.json 2> $out/pmc2.err
echo "pmc2 done"
python tools/pmc_summary.py $out/pmc_summary.json $out/pmc1 $out/pmc2 > $out/pmc_summary.txt
