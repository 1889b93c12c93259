#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_c4pmc; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES -d $out/pmc1 -o p -- python bench.py --batch 65536 --final-dim 256 --score-dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-h2d > $out/pmc1.json 2> $out/pmc1.err
echo pmc1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES -d $out/pmc2 -o p -- python bench.py --batch 65536 --final-dim 256 --score-dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-h2d > $out/pmc2.json 2> $out/pmc2.err
echo pmc2
python tools/pmc_summary.py $out/pmc_summary.json $out/pmc1 $out/pmc2 > $out/pmc_summary.txt
python - <<'P'
import json
d=json.load(open('gpurun_out/r03_c4pmc/pmc_summary.json'))
for k,v in d.items():
    if 'score' in k: print(k, v)
P
