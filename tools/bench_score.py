#!/usr/bin/env python3
"""Micro-benchmark of the bf16 score kernels (pack, fwd both directions, bwd both directions) at B x D."""
import argparse, json, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import jodalrob_twotower_amd  # noqa
from jodalrob_twotower_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--dim", type=int, default=64)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--no-col-rank", action="store_true")
ap.add_argument("--top1", action="store_true")
ap.add_argument("--prescale", action="store_true", help="pack the notice image times inv_t*log2(e): the kernels' unit form")
ap.add_argument("--fp8", action="store_true", help="e4m3 operands for the S products (tt_score_pack2_fp8 / tt_score_fwd_sym_fp8 / tt_score_bwd_fp8)")
ap.add_argument("--sym", action="store_true", help="single-pass symmetric forward (tt_score_fwd_sym_bf16) instead of the two-direction kernel")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, D = a.batch, a.dim
n = torch.nn.functional.normalize(torch.randn(B, D, device=dev), dim=1)
c = torch.nn.functional.normalize(torch.randn(B, D, device=dev), dim=1)
one = torch.ones(1, device=dev)
t = {"pack": [], "fwd": [], "bwd": []}
for i in range(a.iters + 5):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    sn = ops.score_unit_scale(1.0) if a.prescale else 1.0
    if a.fp8:
        Np, Cp = ops.score_pack2_fp8(n, c, sn, 1.0)
    else:
        Np, Cp = ops.score_pack_bf16(n, sn), ops.score_pack_bf16(c)
    ev[1].record()
    if a.sym or a.fp8:
        rs, cs, dg, rr, inv, out8, loss = ops.score_fwd_sym(Np, Cp, B, D, 1.0, 1.0, sn, True, fp8=a.fp8)
    else:
        rs, cs, dg, rr, cr, ss, inv = ops.score_fwd_bf16(Np, Cp, B, D, 1.0, 1.0, not a.no_col_rank, not a.top1, sn, with_inv=True)
        out8, loss = ops.score_loss_finish(B, 1.0, rs, cs, dg, rr, rr, ss)
    ev[2].record()
    dN, dC = ops.score_bwd_bf16(Np, Cp, B, D, 1.0, 1.0, rs, cs, one, 1.0 / (2 * B), sn, inv if a.prescale else None, fp8=a.fp8)
    ev[3].record()
    if i >= 5:
        torch.cuda.synchronize()
        t["pack"].append(ev[0].elapsed_time(ev[1]) * 1e3); t["fwd"].append(ev[1].elapsed_time(ev[2]) * 1e3); t["bwd"].append(ev[2].elapsed_time(ev[3]) * 1e3)
fl_f, fl_b = 2 * 2 * B * B * D, 2 * 4 * B * B * D
out = {k: round(sum(v) / len(v), 1) for k, v in t.items()}
out["fwd_TFLOPs"] = round(fl_f / out["fwd"] / 1e6, 1); out["bwd_TFLOPs"] = round(fl_b / out["bwd"] / 1e6, 1)
print(json.dumps(out))
