"""CPU tests of the host side: schema / vocab rules against the reference-produced fixtures, C-ABI library
loads and exports every symbol include/twotower.h declares (no compute calls), ctypes struct layouts equal
the C layouts, drop-in state-dict layout, loud failure without a GPU."""
import ctypes
import json
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import GOLD, ROOT

import jodalrob_twotower_amd as tt
from jodalrob_twotower_amd import _lib, schema as S, synthetic


def test_schema_matches_reference(schema_syn, schema_real):
    kw = dict(notice_table="notice", company_table="company", pair_table="bid_two_tower",
              pair_notice_id_cols=["bidntceno", "bidntceord"], pair_company_id_cols=["bizno"])
    s = S.build_torchrec_schema_from_meta(metadata_path=GOLD / "synthetic_metadata.csv", **kw)
    for side in ("notice", "company"):
        ref, got = schema_syn[side], getattr(s, side)
        assert (got.table, got.pk_cols, got.numeric, got.categorical, got.text) == \
               (ref["table"], ref["pk_cols"], ref["numeric"], ref["categorical"], ref["text"])
    assert s.pair.notice_id_cols == schema_syn["pair"]["notice_id_cols"]
    r = S.build_torchrec_schema_from_meta(metadata_path=GOLD / "real_vocab_metadata.csv", **kw)
    assert r.notice.categorical == schema_real["notice"]["categorical"] and len(r.notice.categorical) == 32
    assert r.company.categorical == schema_real["company"]["categorical"] and len(r.company.categorical) == 6


def test_vocab_rule(schema_syn, schema_real, capsys):
    emb = tt.CategoricalEmbedder(schema_syn["notice"]["categorical"], GOLD / "synthetic_metadata.csv", "notice", 4, device="cpu")
    assert [emb.vocab_sizes[k] for k in emb.keys] == schema_syn["notice"]["vocab_sizes"]         # count + 10, 1000 if empty
    unk = tt.CategoricalEmbedder(["not_in_meta"], GOLD / "synthetic_metadata.csv", "notice", 4, device="cpu")
    assert unk.vocab_sizes["not_in_meta"] == schema_syn["unknown_key_vocab"] == 1000
    bad = tt.CategoricalEmbedder(["a", "b"], GOLD / "does_not_exist.csv", "notice", 4, device="cpu")
    assert bad.vocab_sizes == {"a": 1000, "b": 1000}
    real = tt.CategoricalEmbedder(schema_real["company"]["categorical"], GOLD / "real_vocab_metadata.csv", "company", 4, device="cpu")
    assert [real.vocab_sizes[k] for k in real.keys] == schema_real["company"]["vocab_sizes"]
    assert sum(schema_real["notice"]["vocab_sizes"]) == 60024 and sum(schema_real["company"]["vocab_sizes"]) == 4117


def test_state_dict_layout_is_the_references(manifest, schema_real):
    task = tt.create_two_tower_train_task(schema_real["notice"]["categorical"], schema_real["company"]["categorical"],
                                          metadata_path=str(GOLD / "real_vocab_metadata.csv"), categorical_embedding_dim=32,
                                          notice_dense_input_dim=256, company_dense_input_dim=128, tower_hidden_dims=[128, 64],
                                          final_embedding_dim=64, dropout_rate=0.1, device="cpu")
    got = {k: list(v.shape) for k, v in task.state_dict().items()}
    assert got == manifest["state_dict_keys_real"]
    assert sum(p.numel() for p in task.parameters()) == 2204832
    # per-key parameters are views of ONE fused table; load_state_dict writes through them
    store = task.two_tower_model.embedding_store
    assert store.rows == 60024 + 4117
    k0 = "two_tower_model.company_tower.categorical_embedder.embeddings.rgnnm.weight"
    new = torch.full_like(task.state_dict()[k0], 3.0)
    task.load_state_dict({k0: new}, strict=False)
    emb = task.two_tower_model.company_tower.categorical_embedder
    off = int(emb._key_row_offset[emb.keys.index("rgnnm")])
    assert torch.equal(store.weight[off:off + new.shape[0]], new)


def test_error_conventions(schema_syn):
    kn, kc = schema_syn["notice"]["categorical"], schema_syn["company"]["categorical"]
    with pytest.raises(ValueError):
        tt.create_two_tower_train_task(kn, kc, metadata_path=str(GOLD / "synthetic_metadata.csv"), loss_type="hinge", device="cpu")
    task = tt.create_two_tower_train_task(kn, kc, metadata_path=str(GOLD / "synthetic_metadata.csv"), categorical_embedding_dim=4,
                                          notice_dense_input_dim=3, company_dense_input_dim=2, tower_hidden_dims=[8, 4],
                                          final_embedding_dim=4, device="cpu")
    mk = lambda b: {"notice": {"dense": torch.zeros(b[0], 3), "kjt": tt.build_batch_kjt(torch.zeros(b[0], 5, dtype=torch.long), kn)},
                    "company": {"dense": torch.zeros(b[1], 2), "kjt": tt.build_batch_kjt(torch.zeros(b[1], 2, dtype=torch.long), kc)}}
    with pytest.raises(ValueError):
        task(mk((4, 5)))                                       # batch-size mismatch (two_tower_train_task.py:64-67)
    with pytest.raises(_lib.TwoTowerHipError):
        task(mk((4, 4)))                                       # no CPU fallback: fails loudly off-GPU
    with pytest.raises(AssertionError):
        tt.TwoTowerModel({"categorical_keys": kn, "metadata_path": str(GOLD / "synthetic_metadata.csv"), "final_embedding_dim": 8,
                          "device": "cpu"},
                         {"categorical_keys": kc, "metadata_path": str(GOLD / "synthetic_metadata.csv"), "final_embedding_dim": 4,
                          "device": "cpu"}, final_embedding_dim=8, device="cpu")


def test_kjt_and_id_mappings_golden():
    z = np.load(GOLD / "kjt_wire.npz")
    k = tt.build_batch_kjt(torch.from_numpy(z["ids"]), ["a", "b", "c"])
    assert np.array_equal(k.values().numpy(), z["values"]) and np.array_equal(k.lengths().numpy(), z["lengths"])
    assert k.keys() == ["a", "b", "c"] and k.to("cpu").device().type == "cpu"
    j = json.loads((GOLD / "id_mappings.json").read_text())
    pre = tt.FeaturePreprocessor.__new__(tt.FeaturePreprocessor)
    n2i, c2i = pre.build_id_mappings({"notice": {"ids": [tuple(t) for t in j["notice_ids"]]}, "company": {"ids": j["company_ids"]}})
    assert n2i == {tuple(k_): v for k_, v in j["notice_id_to_idx"]} and c2i == {k_: v for k_, v in j["company_id_to_idx"]}


def test_scale_vocabs(schema_real):
    for total in (1_000_000, 100_000_000, 10_000_000):
        v = synthetic.scale_vocabs(schema_real["notice"]["vocab_sizes"], total)
        assert sum(v) == total and min(v) >= 2 and len(v) == 32


def test_c_abi_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "twotower.h").read_text()
    declared = set(re.findall(r"^\s*(?:int|size_t|uint64_t|void|float|const char\*)\s+(tt_\w+)\s*\(", header, flags=re.M))
    assert declared, "no declarations parsed"
    lib = _lib.load()                                          # dlopen + resolve (no GPU needed)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/twotower.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.tt_abi_version() == 2
    assert lib.tt_dedup_workspace_bytes(1000) > 0 and lib.tt_score_pack_bytes(100, 64) == 4 * 128 * 64


def test_ctypes_structs_match_c_layout(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "twotower.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(tt_embed_side),sizeof(tt_grad_src),sizeof(tt_adam_tensor),sizeof(tt_tower_params),sizeof(tt_tower_acts),'
                   'sizeof(tt_tower_grads),sizeof(tt_score_fwd_dir),sizeof(tt_score_bwd_dir),sizeof(tt_store_side),sizeof(tt_ingest_lookup));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    sizes = list(map(int, subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()))
    mine = [ctypes.sizeof(c) for c in (_lib.EmbedSide, _lib.GradSrc, _lib.AdamTensor, _lib.TowerParams, _lib.TowerActs,
                                       _lib.TowerGrads, _lib.ScoreFwdDir, _lib.ScoreBwdDir, _lib.StoreSide, _lib.IngestLookup)]
    assert sizes == mine


def test_fused_adam_load_state_dict_keeps_sharded_moments():
    """FusedAdam.load_state_dict on a row-wise sharded store: the shard's exp_avg / exp_avg_sq / step survive the round trip
    and self.state[shard] aliases the store-level buffers the kernels update (they used to be dropped and
    re-created as zeros with step 0 on the next step)."""
    import torch
    from jodalrob_twotower_amd.distributed import ShardedStore
    from jodalrob_twotower_amd.optim import FusedAdam

    def make():
        store = ShardedStore(4, 37, 1, 3, "cpu", "sparse", seed=5)
        w = torch.nn.Parameter(torch.ones(3, 2))
        return store, w, FusedAdam([w, store.shard_param], lr=1e-3, stores=[store])

    store, w, opt = make()
    st = opt._state_of(store)
    st["m"].copy_(torch.arange(st["m"].numel(), dtype=torch.float32).view_as(st["m"]))
    st["v"].fill_(0.25)
    st["step"] = 7
    opt.state[store.shard_param]["step"] = torch.tensor(7.0)
    opt.state[w] = {"step": torch.tensor(7.0), "exp_avg": torch.full((3, 2), 2.0), "exp_avg_sq": torch.full((3, 2), 3.0)}
    sd = opt.state_dict()
    store2, w2, opt2 = make()
    opt2.load_state_dict(sd)
    st2 = opt2._store_state[id(store2)]
    assert st2["step"] == 7 and torch.equal(st2["m"], st["m"]) and torch.equal(st2["v"], st["v"])
    sh = opt2.state[store2.shard_param]
    assert float(sh["step"]) == 7.0 and sh["exp_avg"].data_ptr() == st2["m"].data_ptr() and sh["exp_avg_sq"].data_ptr() == st2["v"].data_ptr()
    assert opt2._state_of(store2) is st2                      # the next step keeps using the loaded buffers
    assert torch.equal(opt2.state[w2]["exp_avg"], torch.full((3, 2), 2.0))


def test_bench_self_launch_plumbing():
    """`python bench.py --gpus N` with no launcher around it starts its own N rank processes (before anything touches a GPU),
    hands each RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, and only rank 0 writes to stdout: ONE JSON line.  TT_BENCH_LAUNCH_ONLY
    stops each rank right after the argument / environment plumbing (no GPU here)."""
    import json as _json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TT_BENCH_LAUNCH_ONLY"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "3", "--steps", "5", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                              # rank 0 only
    info = _json.loads(lines[0])
    assert info["RANK"] == "0" and info["WORLD_SIZE"] == "3" and info["MASTER_ADDR"] == "127.0.0.1" and info["gpus"] == 3
    others = [_json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith("{") and "launch_only" in ln]
    assert sorted(o["RANK"] for o in others) == ["1", "2"] and all(o["MASTER_PORT"] == info["MASTER_PORT"] for o in others)
    # under an external launcher (WORLD_SIZE already set) it does not spawn again
    env2 = dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r2 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=60)
    assert r2.returncode == 0 and _json.loads(r2.stdout.strip())["RANK"] == "1"
    # a failing rank fails the launch
    r3 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--no-such-flag"], env=env, capture_output=True, text=True, timeout=60)
    assert r3.returncode != 0


def test_train_test_split_is_sklearns():
    """Membership and order of the train / test pair lists == sklearn.model_selection.train_test_split(random_state=seed), the
    call the reference's test-mode loader makes (unified_bid_data_loader.py:1222-1226).  Fixture: oracle/gen_split_fixture.py
    (scikit-learn itself, run in the build container).  Bit-exact: index work."""
    from jodalrob_twotower_amd.data_loader import sklearn_split_indices
    fx = json.loads((GOLD / "split_indices.json").read_text())
    for c in fx["cases"]:
        tr, te = sklearn_split_indices(c["n"], c["test_size"], c["seed"])
        assert (len(tr), len(te)) == (c["n_train"], c["n_test"]), c
        assert tr.dtype == np.int64 and te.dtype == np.int64
        if "train" in c:
            assert tr.tolist() == c["train"] and te.tolist() == c["test"], (c["n"], c["seed"])
        else:
            w = np.arange(1, c["n"] + 1, dtype=np.int64)
            assert tr[:16].tolist() == c["train_head"] and tr[-16:].tolist() == c["train_tail"]
            assert te[:16].tolist() == c["test_head"] and te[-16:].tolist() == c["test_tail"]
            assert int((tr * w[:len(tr)]).sum() % (2 ** 61 - 1)) == c["train_checksum"]
            assert int((te * w[:len(te)]).sum() % (2 ** 61 - 1)) == c["test_checksum"]
        assert sorted(tr.tolist() + te.tolist()) == list(range(c["n"]))        # a partition
    tr, te = sklearn_split_indices(17, 0.0, 42)                                 # test_split == 0: every pair trains, in order (:1227-1228)
    assert tr.tolist() == list(range(17)) and len(te) == 0
