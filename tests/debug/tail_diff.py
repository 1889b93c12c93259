#!/usr/bin/env python3
"""Debug aid: per-parameter gradient difference between the fused tower tail and the separate kernels (bf16 MLP)."""
import json, os, sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))   # params_init: seeded test inputs only
import jodalrob_twotower_amd as tt
from params_init import init_state_numpy, synth_batch_numpy
GOLD = ROOT / "tests" / "golden"
man = json.loads((GOLD / "manifest.json").read_text())
schema = json.loads((ROOT / "jodalrob-twotower_amd" / "schema_real.json").read_text())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8200
drop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
cfg = dict(man["cases"]["real_schema"])
kn, kc = schema["notice"]["categorical"], schema["company"]["categorical"]
vn, vc = schema["notice"]["vocab_sizes"], schema["company"]["vocab_sizes"]
shapes = {k: tuple(v) for k, v in man["state_dict_keys_real"].items()}
state = init_state_numpy(shapes, 91)
b = synth_batch_numpy(B, vn, vc, cfg["din_n"], cfg["din_c"], 92, oob=False)
dev = torch.device("cuda:0")
outs = {}
for unf in ("1", "0"):
    os.environ["TT_TOWER_UNFUSED_TAIL"] = unf
    torch.manual_seed(1234)
    task = tt.create_two_tower_train_task(kn, kc, metadata_path=str(GOLD / "real_vocab_metadata.csv"), categorical_embedding_dim=cfg["E"],
                                          notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"],
                                          tower_hidden_dims=list(cfg["hidden"]), final_embedding_dim=cfg["D"], dropout_rate=drop,
                                          temperature=cfg["T"], device=dev, mlp_dtype="bf16")
    task.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
    task.train()
    tb = {s: {"dense": torch.from_numpy(b[f"{s}_dense"]).to(dev), "kjt": tt.build_batch_kjt(torch.from_numpy(b[f"{s}_ids"]), k).to(dev)}
          for s, k in (("notice", kn), ("company", kc))}
    from jodalrob_twotower_amd import towers as TW
    TW._DEBUG_KEEP = []
    res = task(tb, return_metrics=True)
    res["loss"].backward()
    torch.cuda.synchronize()
    outs[unf] = {n: p.grad.cpu().numpy() for n, p in task.named_parameters()}
    for ti, k in enumerate(TW._DEBUG_KEEP):
        o, nd, hid, Bk = k["offs"], k["n_dense"], k["hidden"], k["B"]
        buf = k["buf"].cpu().numpy()
        outs[unf][f"~t{ti}.d_y"] = buf[o[nd + 1 + len(hid)]:o[nd + 1 + len(hid)] + Bk * cfg["D"]].copy()
        for i, h in enumerate(hid):
            outs[unf][f"~t{ti}.d_pre{i}"] = buf[o[nd + 1 + i]:o[nd + 1 + i] + Bk * h].copy()
        outs[unf][f"~t{ti}.acts"] = k["acts"].cpu().numpy()
        outs[unf][f"~t{ti}.emb"] = k["emb"].cpu().numpy()
        outs[unf][f"~t{ti}.d_emb"] = k["d_emb"].cpu().numpy()
    TW._DEBUG_KEEP = None
for k, g in outs["1"].items():
    if "embeddings" in k and not k.endswith(f"{kn[0]}.weight"):
        continue
    d = np.linalg.norm(outs["0"][k] - g) / (np.linalg.norm(g) + 1e-30)
    print(f"{d:10.3e}  {int((outs['0'][k] != g).sum()):9d} of {g.size:9d} differ  {k}")
