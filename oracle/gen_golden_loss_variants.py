#!/usr/bin/env python3
"""Golden vectors for the reference's two loss variants (two_tower_train_task.py:114-160): cross-entropy with label
smoothing and loss_type="cosine_embedding".  Same harness as gen_golden.py (the reference's own modules run on CPU with
numpy-generated parameters and inputs); writes tests/golden/case_loss_*.npz and loss_variants.json and touches no other
fixture.  TEST INFRASTRUCTURE ONLY; runs only where /root/reference exists.

Usage:  python oracle/gen_golden_loss_variants.py
"""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "oracle"))
import gen_golden as G  # noqa: E402
from params_init import init_state_numpy, synth_batch_numpy  # noqa: E402


def main():
    assert G.REF.is_dir(), f"reference not found at {G.REF}"
    G._install_standins()
    sys.path.insert(0, str(G.REF))
    os.chdir(G.REF)
    with G.quiet():
        from src.towers.two_tower_train_task import create_two_tower_train_task
        from src.towers.pairs.unified_bid_data_loader import _build_batch_kjt
    torch.manual_seed(0)
    torch.set_num_threads(4)
    syn = json.loads((G.GOLD / "schema_synthetic.json").read_text())
    kn, kc = syn["notice"]["categorical"], syn["company"]["categorical"]
    vn, vc = syn["notice"]["vocab_sizes"], syn["company"]["vocab_sizes"]
    meta = G.GOLD / "synthetic_metadata.csv"
    base = dict(E=8, din_n=12, din_c=6, hidden=(16, 8), D=8)
    cases = {
        "loss_smooth_0p1": dict(base, T=1.0, B=16, seed=500, loss_type="cross_entropy", label_smoothing=0.1),
        "loss_smooth_0p3_temp": dict(base, T=0.5, B=24, seed=510, loss_type="cross_entropy", label_smoothing=0.3),
        "loss_cosine": dict(base, T=1.0, B=16, seed=520, loss_type="cosine_embedding", label_smoothing=0.0),
        "loss_cosine_temp": dict(base, T=0.25, B=24, seed=530, loss_type="cosine_embedding", label_smoothing=0.0),
    }
    manifest = {}
    for name, c in cases.items():
        with G.quiet():
            task = create_two_tower_train_task(kn, kc, metadata_path=str(meta), categorical_embedding_dim=c["E"],
                                               notice_dense_input_dim=c["din_n"], company_dense_input_dim=c["din_c"],
                                               tower_hidden_dims=list(c["hidden"]), final_embedding_dim=c["D"], dropout_rate=0.0,
                                               temperature=c["T"], loss_type=c["loss_type"], device=torch.device("cpu"))
        task.label_smoothing = c["label_smoothing"]           # (the reference factory has no argument for it: :211-249)
        shapes = {k: tuple(v.shape) for k, v in task.state_dict().items()}
        state = init_state_numpy(shapes, c["seed"])
        G.load_numpy_state(task, state)
        b = synth_batch_numpy(c["B"], vn, vc, c["din_n"], c["din_c"], c["seed"] + 1, oob=False)
        batch = G.make_batch(_build_batch_kjt, kn, kc, b)
        task.train(True)
        with G.quiet():
            res = task(batch, return_metrics=True)
        res["loss"].backward()
        out = {"in." + k: v for k, v in b.items()}
        out.update({"state." + k: np.asarray(v) for k, v in state.items()})
        for n, p in task.named_parameters():
            out["grad." + n] = p.grad.detach().numpy().copy()
        out["sim"] = res["similarity_matrix"].detach().numpy().copy()
        for k in ("loss", "accuracy", "positive_similarity_mean", "negative_similarity_mean", "similarity_gap"):
            out["out." + k] = np.asarray(res[k].detach().numpy())
        np.savez(G.GOLD / f"case_{name}.npz", **out)
        manifest[name] = {**{k: (list(v) if isinstance(v, tuple) else v) for k, v in c.items()}, "keys_n": kn, "keys_c": kc,
                          "vocab_n": vn, "vocab_c": vc, "loss": float(res["loss"].detach())}
        print(name, float(res["loss"]))
    (G.GOLD / "loss_variants.json").write_text(json.dumps({"torch": torch.__version__, "cases": manifest}, indent=1))


if __name__ == "__main__":
    main()
