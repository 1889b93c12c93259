"""Debug aid: UnrolledTrainStep against single-step replays with dropout / LR schedule switched off one at a time; prints where they part."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "tests", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import jodalrob_twotower_amd as tt  # noqa: E402
from jodalrob_twotower_amd.graph import GraphedTrainStep  # noqa: E402
from jodalrob_twotower_amd.unrolled import UnrolledTrainStep  # noqa: E402
from jodalrob_twotower_amd.optim import FusedAdam  # noqa: E402
from params_init import init_state_numpy, synth_batch_numpy  # noqa: E402
from test_gpu_parity import make_task, load_state, to_batch  # noqa: E402

DEV = "cuda:0"
cfg = dict(json.load(open(ROOT / "tests" / "golden" / "manifest.json"))["cases"]["wide_b40"])
cfg["B"] = 256
U, n_steps = 2, 4
batches = [synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 1900 + i, oob=True) for i in range(n_steps)]
for drop, sched_on in ((0.0, False), (0.0, True), (0.1, False)):
    out = {}
    for mode in ("single", "unrolled"):
        task = make_task(tt, cfg, embedding_grad="sparse", score_dtype="bf16", mlp_dtype="bf16", dropout_rate=drop)
        load_state(task, init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 57))
        task.train()
        task._pair_check_done = True
        opt = FusedAdam.for_task(task, lr=1e-2, weight_decay=1e-5)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, (lambda s: (s + 1) / 4 if s < 3 else 1.0) if sched_on else (lambda s: 1.0))
        tb = [to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]) for b in batches]
        losses = []
        if mode == "single":
            gs = GraphedTrainStep(task, opt, tb[0], warmup=2)
            for i, b in enumerate(tb):
                torch.manual_seed(1000 + i)
                losses.append(gs.step(b)["loss"].item())
                sched.step()
            statics = None
        else:
            gs = UnrolledTrainStep(task, opt, tb[0], unroll=U, warmup=2)
            nxt = [0]

            def after_each():
                sched.step()
                nxt[0] += 1
                torch.manual_seed(1000 + nxt[0])
            for i in range(0, n_steps, U):
                torch.manual_seed(1000 + i)
                nxt[0] = i
                res = gs.step_many(tb[i:i + U], after_each=after_each)
                losses += [r["loss"].item() for r in res]
            torch.cuda.synchronize()
            # did the re-pointed hand-over nodes copy the LAST launch's batches into their lanes?
            for j, lane in enumerate(gs._lanes):
                want = tb[n_steps - U + j]
                print(f"   lane {j}: dense copied {torch.equal(lane['static']['notice']['dense'], want['notice']['dense'])}, "
                      f"ids copied {torch.equal(lane['static']['notice']['kjt'].values(), want['notice']['kjt'].values())}")
        out[mode] = losses
        gs.close()
    print(f"dropout {drop} schedule {sched_on}: single {out['single']}\n{'':31}unrolled {out['unrolled']}  equal {out['single'] == out['unrolled']}", flush=True)
