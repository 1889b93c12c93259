"""Poison every torch.empty / empty_like made by the package (NaN for floats, -1 bits for ints) under the ordinary allocator: a
kernel that reads memory nobody wrote shows up as a wrong / NaN result.  POISON_SKIP=file:line,... leaves those call sites alone."""
import os, sys, json, faulthandler, traceback
from pathlib import Path
faulthandler.enable()
root = Path(__file__).resolve().parents[2]   # tests/debug/ -> the repository
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "tests")); sys.path.insert(0, str(root / "oracle"))
import torch
import numpy as np
_empty, _empty_like = torch.empty, torch.empty_like
skip = set(filter(None, os.environ.get("POISON_SKIP", "").split(",")))
only = set(filter(None, os.environ.get("POISON_ONLY", "").split(",")))
seen = {}
def _site():
    for fr in traceback.extract_stack()[:-2][::-1]:
        if "jodalrob" in fr.filename:
            return f"{Path(fr.filename).name}:{fr.lineno}"
    return None
def _poison(t, site):
    if site is None or site in skip or (only and site not in only) or t.device.type != "cuda" or t.numel() == 0:
        return t
    seen[site] = seen.get(site, 0) + 1
    if t.is_floating_point():
        t.fill_(float("nan"))
    elif t.dtype in (torch.int32, torch.int64, torch.uint8, torch.int16):
        t.view(torch.uint8).fill_(0x7f)
    return t
def empty(*a, **k): return _poison(_empty(*a, **k), _site())
def empty_like(*a, **k): return _poison(_empty_like(*a, **k), _site())
torch.empty, torch.empty_like = empty, empty_like
import jodalrob_twotower_amd as tt
import test_gpu_parity as T
manifest = json.load(open(root / "tests/golden/manifest.json"))
case = os.environ.get("CASE", "tiny_train")
cfg = manifest["cases"][case]
g = T.load_case(case)
batch = T.to_batch(tt, T.split_prefix(g, "in."), cfg["keys_n"], cfg["keys_c"])
def err(a, b): return float(np.nanmax(np.abs(a - b))) if np.isfinite(a).all() else float("nan")
task = T.make_task(tt, cfg)
T.load_state(task, T.split_prefix(g, "state."))
task.train(cfg["train"])
with torch.no_grad():
    ne, ce = task.two_tower_model(batch["notice"], batch["company"])
print("model-only fwd: notice err", err(ne.cpu().numpy(), g["out.notice_emb"]), "company err", err(ce.cpu().numpy(), g["out.company_emb"]), flush=True)
res = task(batch, return_metrics=True)
print("task fwd: loss", res["loss"].item(), "golden", float(g["out.loss"]), "sim err", err(res["similarity_matrix"].cpu().numpy(), g["sim"]), flush=True)
if cfg["train"]:
    res["loss"].backward()
    ref = T.split_prefix(g, "grad.")
    worst = max((err(p.grad.cpu().numpy(), ref[n]), n) for n, p in task.named_parameters())
    print("worst grad err", worst, flush=True)
print("sites", json.dumps(seen), flush=True)
