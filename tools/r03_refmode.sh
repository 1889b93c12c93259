#!/bin/bash
# the reference driver's own default mode through the drop-in (scripts/train.py:84-134 of the reference: batch 256, towers [512, 256] -> 128,
# f32, eager launches from Python, torch.optim.Adam on dense table gradients, the real 60,024 + 4,117-row tables) -- the figure
# comparable to the reference README's 23 it/s -- and the same shapes on the fast path (captured step, bf16, sparse gradients)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_refmode; mkdir -p $out
S="--batch 256 --hidden 512,256 --final-dim 128 --rows-notice 60024 --rows-company 4117 --steps 300 --warmup 30 --no-cpu-baseline --no-h2d"
timeout -k 10 300 python bench.py $S --score-dtype fp32 --mlp-dtype fp32 --optimizer torch_adam --mode eager > $out/reference_mode.json 2> $out/a.err
timeout -k 10 300 python bench.py $S > $out/fast_mode.json 2> $out/b.err
python - <<'P'
import json
for f in ("reference_mode","fast_mode"):
    d=json.loads(open(f"gpurun_out/r03_refmode/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], round(1e3/d["ms_per_step"],1), "it/s", d["config"]["launch"], d["config"]["optimizer"])
P
