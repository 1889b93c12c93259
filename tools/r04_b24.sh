#!/bin/bash
# soak (20,000 replayed steps) and the real loop (scripts/train.py --fast, three epochs at B = 8192) on the final library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_b24; mkdir -p $out
timeout -k 10 600 python bench.py --no-extra-legs --no-cpu-baseline --no-h2d --steps 20000 --warmup 100 > $out/soak20k.json 2> $out/soak.err; echo "soak rc $?"
cd /tmp && timeout -k 10 600 python $GRAFT_REPO_ROOT/scripts/train.py --entities 100000 --pairs 2000000 --batch-size 8192 --fast --epochs 3 --output-dir /tmp/tt_models --results-csv /tmp/tt_results.csv > $GRAFT_REPO_ROOT/$out/train_fast_b8192.txt 2>&1; echo "train rc $?"
cd $GRAFT_REPO_ROOT
if grep -q "Memory access fault" $out/*.err $out/*.txt; then echo FAULT; exit 1; fi
python - <<'P'
import json
d=json.loads(open("gpurun_out/r04_b24/soak20k.json").read().strip().splitlines()[-1])
print("soak:", d["steps"], "steps", round(d["value"]/1e6,3), "M pairs/s", round(d["ms_per_step"],5), "ms/step, final loss", d["final_loss"])
P
grep -i "ms/step\|pairs/s\|epoch\|recall\|loss" $out/train_fast_b8192.txt | tail -14 | cut -c1-220
