#!/usr/bin/env python3
"""Micro-benchmark of the table-gradient tail at the bench shape: rows-only lookup -> keyed dedup plan ->
segmented gradient reduction -> row-sparse Adam.  Prints the segment-length histogram of the plan and HIP-event
times per stage (run under `rocprofv3 --kernel-trace --stats` for per-kernel times)."""
import argparse, json, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import jodalrob_twotower_amd  # noqa
from jodalrob_twotower_amd import ops, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--zipf", type=float, default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")
B, E = a.batch, 32
schema = synthetic.load_real_schema(ROOT / "jodalrob-twotower_amd" / "schema_real.json")
keys_n, keys_c = schema["notice"]["categorical"], schema["company"]["categorical"]
vn = synthetic.scale_vocabs(schema["notice"]["vocab_sizes"], a.rows)
vc = synthetic.scale_vocabs(schema["company"]["vocab_sizes"], a.rows)
batch = synthetic.make_batch(B, vn, vc, keys_n, keys_c, 256, 128, dev, 7, a.zipf)
Kn, Kc = len(vn), len(vc)


def offs(v, base):
    o, acc = [], base
    for x in v:
        o.append(acc); acc += x
    return torch.tensor(o, dtype=torch.int64, device=dev), acc


on, end_n = offs(vn, 0)
oc, total = offs(vc, end_n)
sides = [ops.LookupSide(batch["notice"]["kjt"].values(), on, torch.tensor(vn, dtype=torch.int64, device=dev), None, Kn),
         ops.LookupSide(batch["company"]["kjt"].values(), oc, torch.tensor(vc, dtype=torch.int64, device=dev), None, Kc)]
rows = ops.embed_lookup(None, sides, B, True, E=E, table_rows=total)
plan = ops.dedup_plan_keyed(rows, [Kn, Kc], B)
torch.cuda.synchronize()
U = int(plan.n_unique.item())
seg = plan.seg_offsets[:U + 1].cpu()
ln = (seg[1:] - seg[:-1])
hist = {f"<= {t}": int((ln <= t).sum()) for t in (1, 2, 4, 8, 16, 32, 64, 128, 256)}
slots_in = {f"<= {t}": int(ln[ln <= t].sum()) for t in (1, 2, 4, 8, 16, 32, 64, 128, 256)}
print(json.dumps({"M": plan.M, "unique": U, "max_len": int(ln.max()), "rows_with_len": hist, "slots_in_rows_with_len": slots_in}))

dxn = torch.randn(B, 128 + Kn * E, device=dev)
dxc = torch.randn(B, 128 + Kc * E, device=dev)
srcs = [(dxn[:, 128:], Kn), (dxc[:, 128:], Kc)]
out = torch.empty(plan.M, E, device=dev)
table = torch.randn(total, E, device=dev)
m, v = torch.zeros_like(table), torch.zeros_like(table)
t = {"plan": [], "grad": [], "adam": []}
for i in range(a.iters + 5):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    p2 = ops.dedup_plan_keyed(rows, [Kn, Kc], B)
    ev[1].record()
    ops.embed_grad(plan, srcs, B, E, ops.TT_GRAD_SPARSE, out)
    ev[2].record()
    ops.adam_sparse(table, m, v, plan, out, i + 1, 1e-3, 0.9, 0.999, 1e-8, 0.0)
    ev[3].record()
    if i >= 5:
        torch.cuda.synchronize()
        for k, j in (("plan", 0), ("grad", 1), ("adam", 2)):
            t[k].append(ev[j].elapsed_time(ev[j + 1]) * 1e3)
print(json.dumps({k: round(sum(x) / len(x), 1) for k, x in t.items()}))
