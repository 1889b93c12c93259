#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE's own hot-path code on CPU and freezes
inputs/outputs as small fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container (needs /root/reference);
nothing here travels as product code and nothing on the GPU box imports the reference.

How the reference is made importable here -- reference files untouched:
  * `torchrec` is not installed.  The reference uses it ONLY for the KeyedJaggedTensor
    container class (src/towers/cat_embed.py:8,92-94; src/towers/tower/base_tower.py:5;
    src/towers/pairs/unified_bid_data_loader.py:827-841) -- no TorchRec arithmetic is on
    the path -- so a container stand-in with keys()/values()/lengths()/to() is registered
    under that module name.
  * `dotenv` (imported by data/database_connector.py:6, reached through the loader module's
    import chain) gets a no-op `load_dotenv`.
  * BaseTower.__init__ ends in an unconditional self.to("cuda:0") (base_tower.py:69);
    nn.Module.to is wrapped so cuda* -> cpu while the harness runs (no GPU here).
All arithmetic that produces the vectors is the reference's: nn.Embedding / nn.Linear /
nn.BatchNorm1d / F.normalize / torch.mm / F.cross_entropy / optim.Adam (torch 2.10 CPU).

Parameters are overwritten with numpy-generated values (oracle/params_init.py) so that
fixtures do not depend on torch's RNG stream.

Usage:  python oracle/gen_golden.py            (writes tests/golden/*)
"""
from __future__ import annotations

import contextlib
import io
import json
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF = Path(os.environ.get("TT_REFERENCE_ROOT", "/root/reference"))
GOLD = ROOT / "tests" / "golden"
sys.path.insert(0, str(ROOT / "oracle"))
from params_init import init_state_numpy, synth_batch_numpy  # noqa: E402


# ----------------------------------------------------------------------------- stand-ins
class _KJT:
    """Container stand-in for torchrec.KeyedJaggedTensor (ids container only)."""

    def __init__(self, keys, values, lengths):
        self._keys, self._values, self._lengths = list(keys), values, lengths

    @classmethod
    def from_lengths_sync(cls, keys, values, lengths):
        return cls(keys, values, lengths)

    def keys(self):
        return self._keys

    def values(self):
        return self._values

    def lengths(self):
        return self._lengths

    def to(self, device, non_blocking=False):
        return _KJT(self._keys, self._values.to(device), self._lengths.to(device))

    def device(self):
        return self._values.device

    def pin_memory(self):
        return self


def _install_standins():
    tr = types.ModuleType("torchrec")
    tr.KeyedJaggedTensor = _KJT
    sp = types.ModuleType("torchrec.sparse")
    jt = types.ModuleType("torchrec.sparse.jagged_tensor")
    jt.KeyedJaggedTensor = _KJT
    sp.jagged_tensor = jt
    tr.sparse = sp
    sys.modules.update({"torchrec": tr, "torchrec.sparse": sp, "torchrec.sparse.jagged_tensor": jt})
    de = types.ModuleType("dotenv")
    de.load_dotenv = lambda *a, **k: False
    sys.modules["dotenv"] = de

    orig_to = torch.nn.Module.to

    def to_cpu(self, *args, **kwargs):
        def fix(a):
            if isinstance(a, torch.device) and a.type == "cuda":
                return torch.device("cpu")
            if isinstance(a, str) and a.startswith("cuda"):
                return "cpu"
            return a
        return orig_to(self, *[fix(a) for a in args], **{k: fix(v) for k, v in kwargs.items()})

    torch.nn.Module.to = to_cpu


@contextlib.contextmanager
def quiet():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield buf


# ----------------------------------------------------------------------------- synthetic metadata
HEADER = "테이블명,컬럼명,타입,사용 여부,NULL 전략,범주형 여부,범주 갯수,NULL 갯수,길이,PK,NN,국문 설명,비고"
SYN_ROWS = [
    # table, column, type, use, nullstrat, is_cat, n_cat, n_null, len, pk, nn, desc, note
    ("notice", "nid", "text", "Y", "", "", "", 0, "", "Y", "Y", "id", ""),
    ("notice", "nord", "text", "Y", "", "", "", 0, "", "Y", "Y", "ord", ""),
    ("notice", "n0", "integer", "Y", "", "", "", 0, "", "", "", "", ""),
    ("notice", "n1", "double precision", "Y", "", "", "", 0, "", "", "", "", ""),
    ("notice", "n2", "numeric", "Y", "", "", "", 0, "", "", "", "", ""),
    ("notice", "na", "character(1)", "Y", "", "Y", 2, 0, "", "", "", "", ""),
    ("notice", "nb", "text", "Y", "", "Y", 5, 0, "", "", "", "", ""),
    ("notice", "nskip", "text", "N", "", "Y", 9, 0, "", "", "", "unused", ""),
    ("notice", "nc", "text", "Y", "", "Y", 17, 0, "", "", "", "", ""),
    ("notice", "nd", "text", "Y", "", "Y", 40, 0, "", "", "", "", ""),
    ("notice", "ne", "text", "Y", "", "Y", "", 0, "", "", "", "no count -> vocab 1000", ""),
    ("notice", "ntitle", "text", "Y", "", "N", "", 0, "", "", "", "free text", ""),
    ("company", "bizno", "text", "Y", "", "", "", 0, "", "Y", "Y", "id", ""),
    ("company", "c0", "bigint", "Y", "", "", "", 0, "", "", "", "", ""),
    ("company", "ca", "character(1)", "Y", "", "Y", 3, 0, "", "", "", "", ""),
    ("company", "cb", "text", "Y", "", "Y", 50, 0, "", "", "", "", ""),
]


def write_synth_metadata(path: Path):
    lines = [HEADER] + [",".join(str(x) for x in r) for r in SYN_ROWS]
    path.write_text("\n".join(lines) + "\n", encoding="utf-8")


# ----------------------------------------------------------------------------- helpers
def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def load_numpy_state(task, state):
    with torch.no_grad():
        sd = task.state_dict()
        for k, v in state.items():
            sd[k].copy_(torch.from_numpy(np.asarray(v)))


def make_batch(build_kjt, keys_n, keys_c, b):
    return {
        "notice": {"dense": torch.from_numpy(b["notice_dense"]),
                   "kjt": build_kjt(torch.from_numpy(b["notice_ids"]), keys_n)},
        "company": {"dense": torch.from_numpy(b["company_dense"]),
                    "kjt": build_kjt(torch.from_numpy(b["company_ids"]), keys_c)},
    }


def run_case(create_task, build_kjt, *, keys_n, keys_c, vocab_n, vocab_c, meta, E, din_n, din_c, hidden, D,
             T, B, seed, train, oob):
    with quiet():
        task = create_task(keys_n, keys_c, metadata_path=str(meta), categorical_embedding_dim=E,
                           notice_dense_input_dim=din_n, company_dense_input_dim=din_c,
                           tower_hidden_dims=list(hidden), final_embedding_dim=D, dropout_rate=0.0,
                           temperature=T, loss_type="cross_entropy", device=torch.device("cpu"))
    shapes = {k: tuple(v.shape) for k, v in task.state_dict().items()}
    state = init_state_numpy(shapes, seed)
    load_numpy_state(task, state)
    b = synth_batch_numpy(B, vocab_n, vocab_c, din_n, din_c, seed + 1, oob=oob)
    batch = make_batch(build_kjt, keys_n, keys_c, b)
    task.train(train)
    with quiet():
        res = task(batch, return_metrics=True)
    out = {"in." + k: v for k, v in b.items()}
    if train:
        res["loss"].backward()
        for n, p in task.named_parameters():
            out["grad." + n] = p.grad.detach().numpy().copy()
    # tower embeddings (recomputed in eval of same mode would change BN stats; use hooks instead)
    out["sim"] = res["similarity_matrix"].numpy().copy()
    for k in ("loss", "accuracy", "positive_similarity_mean", "negative_similarity_mean", "similarity_gap"):
        out["out." + k] = np.asarray(res[k].detach().numpy())
    out.update({"state_after." + k: v for k, v in sd_to_np(task.state_dict()).items()
                if "running" in k or "num_batches" in k})
    return task, state, b, out


def main():
    assert REF.is_dir(), f"reference not found at {REF}"
    GOLD.mkdir(parents=True, exist_ok=True)
    _install_standins()
    sys.path.insert(0, str(REF))
    os.chdir(REF)  # reference resolves meta/metadata.csv relative to cwd
    with quiet():
        from src.towers.two_tower_train_task import create_two_tower_train_task
        from src.towers.cat_embed import CategoricalEmbedder
        from src.torchrec_preprocess.schema import build_torchrec_schema_from_meta
        from src.torchrec_preprocess.feature_projector import FeatureProjector
        from src.torchrec_preprocess.feature_preprocessor import FeaturePreprocessor
        from src.towers.pairs.unified_bid_data_loader import _build_batch_kjt
        from src.evaluation.evaluator import TwoTowerEvaluator

    torch.manual_seed(0)
    torch.set_num_threads(4)
    manifest = {"torch": torch.__version__, "numpy": np.__version__, "cases": {}}

    # ---- schema fixtures --------------------------------------------------------------
    syn_meta = GOLD / "synthetic_metadata.csv"
    write_synth_metadata(syn_meta)
    schema_kw = dict(notice_table="notice", company_table="company", pair_table="bid_two_tower",
                     pair_notice_id_cols=["bidntceno", "bidntceord"], pair_company_id_cols=["bizno"])
    syn = build_torchrec_schema_from_meta(metadata_path=str(syn_meta), **schema_kw)
    real = build_torchrec_schema_from_meta(metadata_path=str(REF / "meta/metadata.csv"), **schema_kw)

    def vocabs(keys, table, meta):
        with quiet():
            emb = CategoricalEmbedder(keys=keys, metadata_path=str(meta), table_name=table, embedding_dim=2,
                                      device="cpu")
        return [int(emb.vocab_sizes[k]) for k in keys]

    def schema_json(s, meta):
        return {
            side: {"table": getattr(s, side).table, "pk_cols": getattr(s, side).pk_cols,
                   "numeric": getattr(s, side).numeric, "categorical": getattr(s, side).categorical,
                   "text": getattr(s, side).text,
                   "vocab_sizes": vocabs(getattr(s, side).categorical, side, meta)}
            for side in ("notice", "company")
        } | {"pair": {"table": s.pair.table, "notice_id_cols": s.pair.notice_id_cols,
                      "company_id_cols": s.pair.company_id_cols}}

    syn_js = schema_json(syn, syn_meta)
    real_js = schema_json(real, REF / "meta/metadata.csv")
    # unknown key -> 1000-row table (cat_embed.py:65-68)
    syn_js["unknown_key_vocab"] = vocabs(["not_in_meta"], "notice", syn_meta)[0]
    (GOLD / "schema_synthetic.json").write_text(json.dumps(syn_js, ensure_ascii=False, indent=1))
    (GOLD / "schema_real.json").write_text(json.dumps(real_js, ensure_ascii=False, indent=1))

    # key names + category counts of the real schema as a metadata-format fixture (data only)
    lines = [HEADER]
    for t in ("notice", "company"):
        for c, v in zip(real_js[t]["categorical"], real_js[t]["vocab_sizes"]):
            lines.append(f"{t},{c},text,Y,,Y,{v - 10},0,,,,,")
    (GOLD / "real_vocab_metadata.csv").write_text("\n".join(lines) + "\n", encoding="utf-8")

    kn, kc = syn.notice.categorical, syn.company.categorical
    vn, vc = syn_js["notice"]["vocab_sizes"], syn_js["company"]["vocab_sizes"]

    # ---- a2: id wire format (unified_bid_data_loader.py:827-841) -----------------------
    ids = torch.arange(12, dtype=torch.long).reshape(4, 3) * 7 % 11
    kjt = _build_batch_kjt(ids, ["a", "b", "c"])
    np.savez(GOLD / "kjt_wire.npz", ids=ids.numpy(), values=kjt.values().numpy(), lengths=kjt.lengths().numpy())

    common = dict(keys_n=kn, keys_c=kc, vocab_n=vn, vocab_c=vc, meta=syn_meta)
    cfgs = {
        "tiny_train": dict(E=8, din_n=12, din_c=6, hidden=(16, 8), D=8, T=1.0, B=16, seed=100, train=True, oob=True),
        "tiny_eval": dict(E=8, din_n=12, din_c=6, hidden=(16, 8), D=8, T=1.0, B=16, seed=100, train=False, oob=True),
        "deep_temp": dict(E=4, din_n=10, din_c=5, hidden=(16, 12, 8), D=6, T=0.25, B=24, seed=200, train=True, oob=False),
        "wide_b40": dict(E=16, din_n=20, din_c=9, hidden=(32, 24), D=16, T=0.5, B=40, seed=300, train=True, oob=True),
        "single_hidden": dict(E=4, din_n=7, din_c=3, hidden=(8,), D=4, T=1.0, B=8, seed=400, train=True, oob=False),
    }
    ev = TwoTowerEvaluator(device="cpu")
    for name, cfg in cfgs.items():
        task, state, b, out = run_case(create_two_tower_train_task, _build_batch_kjt, **common, **cfg)
        out.update({"state." + k: v for k, v in state.items()})
        sim = torch.from_numpy(out["sim"])
        with quiet():
            out["eval.recall@5"] = np.asarray(ev.compute_recall_at_k(sim, 5).numpy())
            out["eval.recall@10"] = np.asarray(ev.compute_recall_at_k(sim, 10).numpy())
            out["eval.mrr"] = np.asarray(ev.compute_mrr(sim).numpy())
        # tower outputs in the same mode without touching BN stats again: recompute under no_grad on a copy
        import copy
        t2 = copy.deepcopy(task)
        load_numpy_state(t2, state)
        t2.train(cfg["train"])
        with quiet(), torch.no_grad():
            ne, ce = t2.two_tower_model(make_batch(_build_batch_kjt, kn, kc, b)["notice"],
                                        make_batch(_build_batch_kjt, kn, kc, b)["company"])
        out["out.notice_emb"], out["out.company_emb"] = ne.numpy().copy(), ce.numpy().copy()
        if name == "tiny_eval":
            with quiet():
                pr = task.predict_batch(make_batch(_build_batch_kjt, kn, kc, b), top_k=5)
            out["predict.top_similarities"] = pr["top_similarities"].numpy().copy()
            out["predict.top_indices"] = pr["top_indices"].numpy().copy()
        np.savez_compressed(GOLD / f"case_{name}.npz", **out)
        manifest["cases"][name] = {**{k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()},
                                   "keys_n": kn, "keys_c": kc, "vocab_n": vn, "vocab_c": vc,
                                   "loss": float(out["out.loss"])}

    # ---- a17: Adam(wd=1e-5) + LambdaLR warm-up trajectory (scripts/train.py:231-242,319-330) ----
    cfg = cfgs["tiny_train"]
    with quiet():
        task = create_two_tower_train_task(kn, kc, metadata_path=str(syn_meta), categorical_embedding_dim=cfg["E"],
                                           notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"],
                                           tower_hidden_dims=list(cfg["hidden"]), final_embedding_dim=cfg["D"],
                                           dropout_rate=0.0, temperature=1.0, device=torch.device("cpu"))
    shapes = {k: tuple(v.shape) for k, v in task.state_dict().items()}
    state = init_state_numpy(shapes, 500)
    load_numpy_state(task, state)
    opt = torch.optim.Adam(task.parameters(), lr=1e-3, weight_decay=1e-5)
    warm = 2
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: s / warm if s < warm else 1.0, last_epoch=-1)
    traj = {"state." + k: v for k, v in state.items()}
    task.train()
    n_steps = 4
    for s in range(n_steps):
        b = synth_batch_numpy(cfg["B"], vn, vc, cfg["din_n"], cfg["din_c"], 600 + s, oob=False)
        for k, v in b.items():
            traj[f"step{s}.in.{k}"] = v
        opt.zero_grad()
        with quiet():
            res = task(make_batch(_build_batch_kjt, kn, kc, b), return_metrics=True)
        res["loss"].backward()
        traj[f"step{s}.lr"] = np.asarray(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
        traj[f"step{s}.loss"] = np.asarray(res["loss"].item())
        traj[f"step{s}.accuracy"] = np.asarray(res["accuracy"].item())
    traj.update({"final." + k: v for k, v in sd_to_np(task.state_dict()).items()})
    traj["n_steps"], traj["warmup_steps"] = np.asarray(n_steps), np.asarray(warm)
    np.savez_compressed(GOLD / "adam_trajectory.npz", **traj)
    manifest["cases"]["adam_trajectory"] = {**{k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()},
                                            "n_steps": n_steps, "warmup_steps": warm, "lr": 1e-3, "weight_decay": 1e-5}

    # ---- real 32+6 key schema, E=32, [128,64]->64 (README dims); params by recipe, not stored ----
    rk_n, rk_c = real.notice.categorical, real.company.categorical
    rv_n, rv_c = real_js["notice"]["vocab_sizes"], real_js["company"]["vocab_sizes"]
    rcfg = dict(E=32, din_n=256, din_c=128, hidden=(128, 64), D=64, T=1.0, B=64, seed=700, train=True, oob=False)
    task, state, b, out = run_case(create_two_tower_train_task, _build_batch_kjt, keys_n=rk_n, keys_c=rk_c,
                                   vocab_n=rv_n, vocab_c=rv_c, meta=REF / "meta/metadata.csv", **rcfg)
    small = {k: v for k, v in out.items() if not (k.startswith("grad.") and "embeddings" in k)}
    for k, v in out.items():  # embedding grads: keep touched rows only
        if k.startswith("grad.") and "embeddings" in k:
            nz = np.flatnonzero(np.abs(v).sum(axis=1))
            small[k + ".rows"], small[k + ".vals"] = nz.astype(np.int64), v[nz]
    small["n_params"] = np.asarray(sum(p.numel() for p in task.parameters()))
    np.savez_compressed(GOLD / "case_real_schema.npz", **small)
    manifest["cases"]["real_schema"] = {**{k: (list(v) if isinstance(v, tuple) else v) for k, v in rcfg.items()},
                                        "loss": float(out["out.loss"]), "n_params": int(small["n_params"]),
                                        "state": "recipe: oracle/params_init.init_state_numpy(shapes, seed)"}
    manifest["state_dict_keys_real"] = {k: list(v.shape) for k, v in task.state_dict().items()}

    # ---- a18: FeatureProjector + _apply_projection concat order -----------------------
    rng = np.random.default_rng(800)
    proj = FeatureProjector(num_dim=3, text_dim=768, num_proj_dim=16, text_proj_dim=8)
    pstate = init_state_numpy({k: tuple(v.shape) for k, v in proj.state_dict().items()}, 801)
    with torch.no_grad():
        for k, v in pstate.items():
            proj.state_dict()[k].copy_(torch.from_numpy(v))
    store = {"numeric": rng.standard_normal((37, 3)).astype(np.float32),
             "text": {"ntitle": rng.standard_normal((37, 768)).astype(np.float32)},
             "categorical": rng.integers(0, 5, (37, 5)), "ids": [(f"N{i}", "00") for i in range(37)]}
    fp = FeaturePreprocessor(schema=syn, device="cpu", num_proj_dim=16, text_proj_dim=8, batch_size=16)
    fp.projectors["notice"] = proj
    with quiet():
        st = fp._apply_projection(store, proj, syn.notice, "notice")
    cstore = {"numeric": rng.standard_normal((11, 1)).astype(np.float32), "text": {},
              "categorical": rng.integers(0, 3, (11, 2)), "ids": [f"{1000 + i}" for i in range(11)]}
    cproj = FeatureProjector(num_dim=1, text_dim=768, num_proj_dim=16, text_proj_dim=8)
    cstate = init_state_numpy({k: tuple(v.shape) for k, v in cproj.state_dict().items()}, 802)
    with torch.no_grad():
        for k, v in cstate.items():
            cproj.state_dict()[k].copy_(torch.from_numpy(v))
    with quiet():
        cst = fp._apply_projection(cstore, cproj, syn.company, "company")
    n2i, c2i = fp.build_id_mappings({"notice": store, "company": cstore})
    np.savez_compressed(GOLD / "projector.npz", numeric=store["numeric"], text_ntitle=store["text"]["ntitle"],
                        dense_projected=st["dense_projected"], c_numeric=cstore["numeric"],
                        c_dense_projected=cst["dense_projected"],
                        **{"state." + k: v for k, v in pstate.items()}, **{"cstate." + k: v for k, v in cstate.items()})
    (GOLD / "id_mappings.json").write_text(json.dumps({
        "notice_ids": [list(t) for t in store["ids"]], "company_ids": cstore["ids"],
        "notice_id_to_idx": [[list(k), v] for k, v in n2i.items()],
        "company_id_to_idx": [[k, v] for k, v in c2i.items()]}, indent=1))

    (GOLD / "manifest.json").write_text(json.dumps(manifest, ensure_ascii=False, indent=1))
    print("golden fixtures written to", GOLD)
    for p in sorted(GOLD.iterdir()):
        print(f"  {p.name:32s} {p.stat().st_size:>9d} B")


if __name__ == "__main__":
    main()
