"""GPU tests of the rows either side of the training step: FeatureProjector / FeaturePreprocessor (a18),
device-resident feature store + batch gather (a1, a2), evaluator (Recall@K / MRR), the train driver."""
import json
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle_np as O
from conftest import GOLD, ROOT, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def tt():
    import jodalrob_twotower_amd as m
    return m


def test_feature_projector_golden(tt):
    z = np.load(GOLD / "projector.npz")
    proj = tt.FeatureProjector(num_dim=3, text_dim=768, num_proj_dim=16, text_proj_dim=8).to(DEV)
    proj.load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state.")})
    pn, pt = proj(torch.from_numpy(z["numeric"]).to(DEV), {"ntitle": torch.from_numpy(z["text_ntitle"]).to(DEV)})
    got = torch.cat([pn, pt["ntitle"]], dim=1).cpu().numpy()
    np.testing.assert_allclose(got, z["dense_projected"], rtol=2e-5, atol=2e-6)


def test_feature_preprocessor_pipeline_golden(tt, schema_syn):
    from jodalrob_twotower_amd import schema as S
    z = np.load(GOLD / "projector.npz")
    sch = S.build_torchrec_schema_from_meta(notice_table="notice", company_table="company", pair_table="p",
                                            pair_notice_id_cols=["a"], pair_company_id_cols=["b"],
                                            metadata_path=GOLD / "synthetic_metadata.csv")
    pre = tt.FeaturePreprocessor(sch, device=DEV, num_proj_dim=16, text_proj_dim=8, batch_size=16)
    pre.projectors["notice"].load_state_dict({k[6:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state.")})
    pre.projectors["company"].load_state_dict({k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("cstate.")})

    class Src:
        def build_feature_store(self, table, side, chunksize=5000, limit=None):
            if table == "notice":
                return {"ids": [("N", "0")] * 37, "numeric": z["numeric"], "text": {"ntitle": z["text_ntitle"]},
                        "categorical": np.zeros((37, 5), np.int64)}
            return {"ids": ["1"] * 11, "numeric": z["c_numeric"], "text": {}, "categorical": np.zeros((11, 2), np.int64)}
    stores = pre.preprocess_all(Src(), feature_chunksize=10)
    np.testing.assert_allclose(stores["notice"]["dense_projected"], z["dense_projected"], rtol=2e-5, atol=2e-6)    # 3 ragged chunks
    np.testing.assert_allclose(stores["company"]["dense_projected"], z["c_dense_projected"], rtol=2e-5, atol=2e-6)
    assert stores["notice"]["categorical_keys"] == schema_syn["notice"]["categorical"]


def test_device_loader_batches_bit_exact(tt):
    from jodalrob_twotower_amd.data_loader import DeviceFeatureStore, DevicePairLoader
    rng = np.random.default_rng(5)
    ns = {"dense_projected": rng.standard_normal((300, 24)).astype(np.float32), "categorical": rng.integers(0, 50, (300, 5))}
    cs = {"dense_projected": rng.standard_normal((120, 8)).astype(np.float32), "categorical": rng.integers(0, 9, (120, 2))}
    pairs = np.stack([rng.integers(0, 300, 1000), rng.integers(0, 120, 1000)], axis=1)
    loader = DevicePairLoader(DeviceFeatureStore(ns, list("abcde"), DEV), DeviceFeatureStore(cs, list("xy"), DEV), pairs, 256, shuffle=False)
    assert len(loader) == 4
    seen = 0
    for b in loader:
        n = b["notice"]["dense"].shape[0]
        sel = pairs[seen:seen + n]
        assert np.array_equal(b["notice"]["dense"].cpu().numpy(), ns["dense_projected"][sel[:, 0]])
        assert np.array_equal(b["company"]["dense"].cpu().numpy(), cs["dense_projected"][sel[:, 1]])
        assert np.array_equal(b["notice"]["kjt"].values().cpu().numpy(), ns["categorical"][sel[:, 0]].reshape(-1))   # sample-major
        assert np.array_equal(b["company"]["kjt"].values().cpu().numpy(), cs["categorical"][sel[:, 1]].reshape(-1))
        seen += n
    assert seen == 1000                                        # ragged last batch (232)
    with pytest.raises(KeyError):
        DevicePairLoader(loader.notice, loader.company, np.array([[300, 0]]), 8, shuffle=False)


def test_evaluator_golden(tt):
    ev = tt.TwoTowerEvaluator(device=DEV)
    for case in ("tiny_train", "deep_temp", "wide_b40"):
        g = load_case(case)
        S = torch.from_numpy(g["sim"]).to(DEV)
        assert ev.compute_recall_at_k(S, 5).item() == pytest.approx(float(g["eval.recall@5"]), abs=1e-7)
        assert ev.compute_recall_at_k(S, 10).item() == pytest.approx(float(g["eval.recall@10"]), abs=1e-7)
        assert ev.compute_mrr(S).item() == pytest.approx(float(g["eval.mrr"]), rel=1e-6)
    # ties: rank counts equal scores BEFORE the diagonal (first-index-wins, as argsort/argmax)
    S = torch.tensor([[1.0, 1.0, 0.5], [2.0, 1.0, 1.0], [3.0, 3.0, 3.0]], device=DEV)
    from jodalrob_twotower_amd import ops
    assert ops.diag_rank_rows(S).tolist() == [0, 1, 2]


def test_train_driver_smoke(tmp_path):
    """scripts/train.py at BASELINE configs[0]'s own sizes -- 10,000 notice x 10,000 company entities, 100,000 pairs, batch 256, E = 32,
    towers [128, 64] -> 64 (the script's defaults) -- in the reference's loop (a few steps, then resume) and over a whole epoch
    of the fast loop; artefacts as the reference driver leaves them: checkpoints with its dict keys (:506-511), the state-dict
    keys / shapes of the real schema, and a results CSV with its columns in its order (:37-57; tests/golden/api_surface.json)."""
    import csv
    out, res_csv = tmp_path / "models", tmp_path / "train_results.csv"
    base = [sys.executable, str(ROOT / "scripts" / "train.py"), "--results-csv", str(res_csv)]
    r = subprocess.run(base + ["--steps", "12", "--output-dir", str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (out / "final_model.pt").exists() and (out / "model_weights.pt").exists() and (out / "best_model.pt").exists()
    ck = torch.load(out / "final_model.pt", map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}                    # scripts/train.py:506-511
    man = json.loads((GOLD / "manifest.json").read_text())
    assert {k: list(v.shape) for k, v in ck["model_state_dict"].items()} == man["state_dict_keys_real"]
    # resume restores model and optimiser state
    r2 = subprocess.run(base + ["--steps", "3", "--output-dir", str(out), "--resume", str(out / "best_model.pt")],
                        capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    # the fast loop over the whole epoch: captured steps (full batches + the ragged one) fed from the device stores, same artefacts
    out3 = tmp_path / "models_fast"
    r3 = subprocess.run(base + ["--output-dir", str(out3), "--fast"], capture_output=True, text=True, timeout=900)
    assert r3.returncode == 0, r3.stdout[-2000:] + r3.stderr[-2000:]
    assert "fast: captured step fed from the device stores" in r3.stdout and "--- 추론 예제 ---" in r3.stdout
    ck3 = torch.load(out3 / "final_model.pt", map_location="cpu", weights_only=True)
    assert {k: list(v.shape) for k, v in ck3["model_state_dict"].items()} == man["state_dict_keys_real"]
    losses = [float(ln.split("loss")[1].split()[0]) for ln in r3.stdout.splitlines() if ln.startswith("step ")]
    assert len(losses) >= 10 and all(np.isfinite(losses))                 # 313 steps, a line every 20
    # results CSV: the reference's columns, one row per run, values where the reference's row has them
    surface = json.loads((GOLD / "api_surface.json").read_text())["harness"]["scripts/train.py"]
    cols = [c["column"] for c in surface["results_csv"]["columns"]]
    with open(res_csv, newline="", encoding="utf-8") as f:
        rows = list(csv.reader(f))
    assert rows[0] == cols and len(rows) == 4
    fast = dict(zip(rows[0], rows[3]))
    assert fast["batch_size"] == "256" and fast["model_params"] == "2204832" and fast["hidden_dims"] == "[128, 64]" and fast["epochs"] == "1"
    assert fast["train_batches"] == "313" and fast["test_batches"] == "79"            # sklearn's split: 80,000 / 20,000 pairs
    for k in ("train_loss", "train_acc", "val_loss", "val_acc", "recall_at_5", "recall_at_10", "mrr", "similarity_gap"):
        assert np.isfinite(float(fast[k])), (k, fast[k])
    assert abs(float(fast["train_loss"]) - float(np.log(256))) < 1.0


def test_reference_driver_sequence(tt, tmp_path, capsys):
    """The calls of the reference's scripts/train.py in its order (:207-242 factory + torch Adam + LambdaLR, :313-336 loop,
    :367-423 validation, :444-452 evaluator + prediction demo, :497-527 checkpoint) on the drop-in's objects -- every method
    named in tests/golden/api_surface.json for that file, executed."""
    from jodalrob_twotower_amd import synthetic
    from jodalrob_twotower_amd.data_loader import create_unified_bid_dataloaders
    vn, vc = [12, 40, 7], [9, 5]
    meta = synthetic.write_metadata(tmp_path / "metadata.csv", {"notice": {f"n{i}": v for i, v in enumerate(vn)},
                                                               "company": {f"c{i}": v for i, v in enumerate(vc)}})
    with open(meta, "a", encoding="utf-8") as f:
        f.write("notice,bidntceno,text,Y,,,,0,,Y,Y,,\nnotice,bidntceord,text,Y,,,,0,,Y,Y,,\ncompany,bizno,text,Y,,,,0,,Y,Y,,\n"
                "notice,amount,numeric,Y,,,,0,,,,,\ncompany,size,numeric,Y,,,,0,,,,,\n")
    schema = tt.build_torchrec_schema_from_meta(notice_table="notice", company_table="company", pair_table="bid_two_tower",
                                                pair_notice_id_cols=["bidntceno", "bidntceord"], pair_company_id_cols=["bizno"],
                                                metadata_path=str(meta))
    src = synthetic.SyntheticSource(400, 300, 1500, vn, vc)
    train_loader, test_loader = create_unified_bid_dataloaders(db_engine=src, schema=schema, batch_size=128, test_split=0.2, shuffle_seed=42,
                                                               num_workers=0, pin_memory=False, streaming=False, load_all_features=True,
                                                               chunk_size=1000000, feature_chunksize=1000, use_preprocessor=True,
                                                               test_mode=True, pair_limit=1500, device=DEV)
    assert len(train_loader) == 10 and len(test_loader) == 3
    din_n, din_c = next(iter(train_loader))["notice"]["dense"].shape[1], next(iter(train_loader))["company"]["dense"].shape[1]
    train_task = tt.create_two_tower_train_task(notice_categorical_keys=schema.notice.categorical, company_categorical_keys=schema.company.categorical,
                                                metadata_path=str(meta), categorical_embedding_dim=8, notice_dense_input_dim=din_n,
                                                company_dense_input_dim=din_c, tower_hidden_dims=[32, 16], final_embedding_dim=16,
                                                dropout_rate=0.1, temperature=1.0, loss_type="cross_entropy", device=torch.device(DEV))
    optimizer = torch.optim.Adam(train_task.parameters(), lr=1e-3, weight_decay=1e-5)
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: s / 2 if s < 2 else 1.0, last_epoch=-1)
    evaluator = tt.TwoTowerEvaluator(device=torch.device(DEV))
    train_task.train()
    for batch in train_loader:
        optimizer.zero_grad()
        result = train_task(batch, return_metrics=True)
        loss, accuracy = result["loss"], result["accuracy"]
        loss.backward()
        optimizer.step()
        scheduler.step()
        assert np.isfinite(loss.item()) and 0.0 <= accuracy.item() <= 1.0
        _ = (result.get("positive_similarity_mean", torch.tensor(0.0)).item(), result.get("negative_similarity_mean", torch.tensor(0.0)).item(),
             result.get("similarity_gap", torch.tensor(0.0)).item())
    train_task.eval()
    with torch.no_grad():
        for batch in test_loader:
            result = train_task(batch, return_metrics=True)
            assert np.isfinite(result["loss"].item())
    capsys.readouterr()
    test_metrics = evaluator.evaluate_comprehensive(train_task, test_loader, verbose=True)
    sample_metrics = evaluator.evaluate_single_batch(train_task, next(iter(train_loader)), verbose=True)
    evaluator.demonstrate_predictions(train_task, next(iter(train_loader)), top_k=10)
    text = capsys.readouterr().out
    for line in ("테스트 배치 수: 3", "--- 성능 평가 ---", "배치 크기: 128", "--- 랜덤 기준선과 비교 ---", "--- 추론 예제 ---", "유사도 행렬 크기: torch.Size([128, 128])"):
        assert line in text, line
    for k in ("loss", "accuracy", "recall@5", "recall@10", "mrr", "similarity_gap", "num_batches"):          # evaluator.py:197-205
        assert k in test_metrics
    assert test_metrics["num_batches"] == len(test_loader) and sample_metrics["batch_size"] == 128
    # recall / MRR of evaluate_comprehensive equal the reference's formulae on the materialised matrices.  The pairs of a batch
    # share companies here (300 companies, 1500 pairs), so rows hold scores EQUAL to the positive's: ranks follow the stable
    # descending order (oracle diag_rank_stable: torch leaves the order among ties unspecified)
    r5 = mrr = 0.0
    ties = 0
    with torch.no_grad():
        for batch in test_loader:
            S = train_task(batch, return_metrics=True)["similarity_matrix"].cpu().numpy()
            rank = O.diag_rank_stable(S)
            ties += int((rank != O.diag_rank(S)).sum())
            r5 += float((rank < 5).mean())
            mrr += float((1.0 / (rank + 1.0)).mean())
    assert ties > 0, "the batches were meant to contain duplicate companies"
    assert test_metrics["recall@5"] == pytest.approx(r5 / 3, abs=1e-6) and test_metrics["mrr"] == pytest.approx(mrr / 3, rel=1e-5)
    ckpt = {"epoch": 0, "model_state_dict": train_task.state_dict(), "optimizer_state_dict": optimizer.state_dict(), "loss": 0.0}
    torch.save(ckpt, tmp_path / "final_model.pt")
    back = torch.load(tmp_path / "final_model.pt", map_location=DEV, weights_only=True)
    train_task.load_state_dict(back["model_state_dict"])
    optimizer.load_state_dict(back["optimizer_state_dict"])


@pytest.mark.parametrize("dims,B,use_order", [((24, 8), 300, True), ((7, 5), 300, True), ((256, 128), 8192, True), ((16, 4), 64, False), ((12, 12), 1, False)])
def test_ingest_store_equals_gather_then_ingest(tt, dims, B, use_order):
    """tt_batch_ingest_store (pair -> entity row -> dense features, ids, key-major fused rows, + copy segments: ONE launch, no batch
    tensors) == tt_batch_gather per side followed by tt_batch_ingest, bit for bit: static dense buffers, sample-major ids,
    rows_km and the copied scalars.  Ragged last 64-sample tile, feature widths that are not a multiple of four (scalar pieces),
    out-of-range ids in the store (the hand-over clamps like the lookup: cat_embed.py:114-117), with and without a permutation."""
    from jodalrob_twotower_amd import ops
    rng = np.random.default_rng(B + dims[0])
    dev = torch.device(DEV)
    N, M, P = 1000, 400, 3 * B + 17
    vocab = [[40, 9, 3000, 12, 7], [50, 6]]
    Ks = [5, 2]
    dense = [torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(dev) for n, d in zip((N, M), dims)]
    cat = [torch.from_numpy(np.stack([rng.integers(-3, v + 3, n) for v in vs], axis=1).astype(np.int64)).to(dev) for n, vs in zip((N, M), vocab)]
    pairs = torch.from_numpy(np.stack([rng.integers(0, N, P), rng.integers(0, M, P)], axis=1).astype(np.int64)).to(dev)
    order = torch.from_numpy(rng.permutation(P).astype(np.int64)).to(dev) if use_order else None
    lo = B + 3 if P >= 2 * B + 3 else 0
    offs, base = [], 0
    for vs in vocab:
        offs.append(torch.tensor(base + np.concatenate([[0], np.cumsum(vs)[:-1]]), dtype=torch.int64, device=dev))
        base += sum(vs)
    vocs = [torch.tensor(vs, dtype=torch.int64, device=dev) for vs in vocab]
    scal_src = torch.arange(12, dtype=torch.float32, device=dev)

    def fresh():
        return ([torch.full((B, d), -7.0, device=dev) for d in dims], [torch.full((B * k,), -9, dtype=torch.int64, device=dev) for k in Ks],
                torch.full((B * sum(Ks),), -1, dtype=torch.int32, device=dev), torch.zeros(12, device=dev))

    # A: the two-step path
    sd_a, sv_a, km_a, sc_a = fresh()
    sel = pairs[lo:lo + B] if order is None else pairs[order[lo:lo + B]]
    got = [ops.batch_gather(sel[:, i].contiguous(), dense[i], cat[i]) for i in range(2)]
    sides_a = [ops.LookupSide(got[i][1], offs[i], vocs[i], None, Ks[i]) for i in range(2)]
    ops.batch_ingest([(sd_a[0], got[0][0]), (sv_a[0], got[0][1]), (sd_a[1], got[1][0]), (sv_a[1], got[1][1]), (sc_a, scal_src)], sides_a, B, km_a)
    # B: one launch
    sd_b, sv_b, km_b, sc_b = fresh()
    flat = pairs.view(-1)
    base_el = 0 if order is not None else 2 * lo
    sides_b = [ops.LookupSide(None, offs[i], vocs[i], None, Ks[i]) for i in range(2)]
    stores = [ops.StoreSide(flat[base_el + i:], 2, dense[i], cat[i], sd_b[i], sv_b[i]) for i in range(2)]
    ops.batch_ingest_store([(sc_b, scal_src)], sides_b, stores, B, order, km_b, lo if order is not None else 0)
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(sd_a[i], sd_b[i]) and torch.equal(sv_a[i], sv_b[i]), i
        assert np.array_equal(sd_b[i].cpu().numpy(), dense[i].cpu().numpy()[sel[:, i].cpu().numpy()])           # and numpy's fancy index
        assert np.array_equal(sv_b[i].cpu().numpy(), cat[i].cpu().numpy()[sel[:, i].cpu().numpy()].reshape(-1))
    assert torch.equal(km_a, km_b) and int(km_b.min()) >= 0 and torch.equal(sc_a, sc_b) and torch.equal(sc_b, scal_src)
    # rows_km may be left out (batches beyond the keyed plan's reach)
    sd_c, sv_c, _, sc_c = fresh()
    stores_c = [ops.StoreSide(flat[base_el + i:], 2, dense[i], cat[i], sd_c[i], sv_c[i]) for i in range(2)]
    ops.batch_ingest_store([(sc_c, scal_src)], sides_b, stores_c, B, order, None, lo if order is not None else 0)
    assert all(torch.equal(sd_c[i], sd_b[i]) and torch.equal(sv_c[i], sv_b[i]) for i in range(2))


@pytest.mark.parametrize("E,Ks,vocab,B,out_dtype,use_order", [
    (32, [32, 6], None, 8192, "bf16", True),            # the bench shape (real 32 + 6 keys scaled to 1 M rows per tower)
    (32, [32, 6], None, 1000, "f32", False),            # ragged tiles, f32 rows (parity mode)
    (8, [5, 2], [[40, 9, 3000, 12, 7], [50, 6]], 300, "bf16", True),
    (16, [64, 1], None, 70, "f32", True),               # K = 64: 8-sample tiles; a one-key side
    (64, [3, 33], None, 129, "bf16", False),            # K = 33: 8-sample tiles with a partial last chunk
])
def test_ingest_lookup_equals_separate_launches(tt, schema_real, E, Ks, vocab, B, out_dtype, use_order):
    """Hand-over AND lookup in one launch (tt_batch_ingest_lookup / tt_batch_ingest_store_lookup) == the hand-over followed by
    tt_embed_lookup_fwd, bit for bit: the embedding columns of the towers' inputs (f32 copies / RNE bf16), the untouched projection
    columns, rows_km, the static ids and dense buffers, the copied scalars -- from batch tensors and from the device stores, with
    out-of-range ids (clamp: cat_embed.py:114-117), ragged tiles, every supported row width."""
    from jodalrob_twotower_amd import ops, synthetic
    rng = np.random.default_rng(B + E)
    dev = torch.device(DEV)
    if vocab is None:
        if Ks == [32, 6]:
            vocab = [synthetic.scale_vocabs(schema_real["notice"]["vocab_sizes"], 1_000_000), synthetic.scale_vocabs(schema_real["company"]["vocab_sizes"], 1_000_000)]
        else:
            vocab = [[int(v) for v in rng.integers(2, 5000, k)] for k in Ks]
    dims, h0 = (24, 8), 64
    N, M, P = 1500, 700, 3 * B + 17
    R = sum(sum(v) for v in vocab)
    table = torch.from_numpy(rng.standard_normal((R, E)).astype(np.float32)).to(dev)
    dense = [torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).to(dev) for n, d in zip((N, M), dims)]
    cat = [torch.from_numpy(np.stack([rng.integers(-3, v + 3, n) for v in vs], axis=1).astype(np.int64)).to(dev) for n, vs in zip((N, M), vocab)]
    pairs = torch.from_numpy(np.stack([rng.integers(0, N, P), rng.integers(0, M, P)], axis=1).astype(np.int64)).to(dev)
    order = torch.from_numpy(rng.permutation(P).astype(np.int64)).to(dev) if use_order else None
    lo = B + 3
    offs, base = [], 0
    for vs in vocab:
        offs.append(torch.tensor(base + np.concatenate([[0], np.cumsum(vs)[:-1]]), dtype=torch.int64, device=dev))
        base += sum(vs)
    vocs = [torch.tensor(vs, dtype=torch.int64, device=dev) for vs in vocab]
    scal_src = torch.arange(12, dtype=torch.float32, device=dev)
    odt = torch.bfloat16 if out_dtype == "bf16" else torch.float32

    def fresh():
        return ([torch.full((B, d), -7.0, device=dev) for d in dims], [torch.full((B * k,), -9, dtype=torch.int64, device=dev) for k in Ks],
                torch.full((B * sum(Ks),), -1, dtype=torch.int32, device=dev), torch.zeros(12, device=dev),
                [torch.full((B, h0 + k * E), 3.0, dtype=odt, device=dev) for k in Ks])

    sel = pairs[lo:lo + B] if order is None else pairs[order[lo:lo + B]]
    got = [ops.batch_gather(sel[:, i].contiguous(), dense[i], cat[i]) for i in range(2)]      # the batch as tensors
    # A: hand-over, then the lookup launch
    sd_a, sv_a, km_a, sc_a, x_a = fresh()
    copies = lambda sd, sv, sc: [(sd[0], got[0][0]), (sv[0], got[0][1]), (sd[1], got[1][0]), (sv[1], got[1][1]), (sc, scal_src)]
    ops.batch_ingest(copies(sd_a, sv_a, sc_a), [ops.LookupSide(got[i][1], offs[i], vocs[i], None, Ks[i]) for i in range(2)], B, km_a)
    rows_a = ops.embed_lookup(table, [ops.LookupSide(sv_a[i], offs[i], vocs[i], x_a[i][:, h0:], Ks[i]) for i in range(2)], B, want_rows=True)
    # B: one launch from the batch tensors
    sd_b, sv_b, km_b, sc_b, x_b = fresh()
    sides_b = [ops.LookupSide(got[i][1], offs[i], vocs[i], x_b[i][:, h0:], Ks[i]) for i in range(2)]
    assert ops.ingest_lookup_supported(table, sides_b)
    ops.batch_ingest(copies(sd_b, sv_b, sc_b), sides_b, B, km_b, table=table)
    # C: one launch from the device stores
    sd_c, sv_c, km_c, sc_c, x_c = fresh()
    flat = pairs.view(-1)
    base_el = 0 if order is not None else 2 * lo
    sides_c = [ops.LookupSide(None, offs[i], vocs[i], x_c[i][:, h0:], Ks[i]) for i in range(2)]
    stores_c = [ops.StoreSide(flat[base_el + i:], 2, dense[i], cat[i], sd_c[i], sv_c[i]) for i in range(2)]
    ops.batch_ingest_store([(sc_c, scal_src)], sides_c, stores_c, B, order, km_c, lo if order is not None else 0, table=table)
    # D / E: the hand-over (from batch tensors / from the stores) also leaves the rows in slot order, the lookup reads those
    # (tt_embed_lookup_rows_fwd: what a captured step does) -- the same bits in x, and rows_sm == the id lookup's own rows
    sd_d, sv_d, km_d, sc_d, x_d = fresh()
    sm_d = torch.full_like(km_d, -1)
    ops.batch_ingest(copies(sd_d, sv_d, sc_d), [ops.LookupSide(got[i][1], offs[i], vocs[i], None, Ks[i]) for i in range(2)], B, km_d, rows_sm=sm_d)
    ops.embed_lookup_rows(table, sm_d, [ops.LookupSide(None, None, None, x_d[i][:, h0:], Ks[i]) for i in range(2)], B)
    sd_e, sv_e, km_e, sc_e, x_e = fresh()
    sm_e = torch.full_like(km_e, -1)
    stores_e = [ops.StoreSide(flat[base_el + i:], 2, dense[i], cat[i], sd_e[i], sv_e[i]) for i in range(2)]
    ops.batch_ingest_store([(sc_e, scal_src)], [ops.LookupSide(None, offs[i], vocs[i], None, Ks[i]) for i in range(2)], stores_e, B, order, km_e,
                           lo if order is not None else 0, rows_sm=sm_e)
    ops.embed_lookup_rows(table, sm_e, [ops.LookupSide(None, None, None, x_e[i][:, h0:], Ks[i]) for i in range(2)], B)
    torch.cuda.synchronize()
    assert torch.equal(sm_d, rows_a) and torch.equal(sm_e, rows_a)
    for name, (sd, sv, km, sc, x) in (("batch", (sd_b, sv_b, km_b, sc_b, x_b)), ("store", (sd_c, sv_c, km_c, sc_c, x_c)),
                                      ("batch + rows lookup", (sd_d, sv_d, km_d, sc_d, x_d)), ("store + rows lookup", (sd_e, sv_e, km_e, sc_e, x_e))):
        for i in range(2):
            assert torch.equal(x[i].view(torch.int16 if odt == torch.bfloat16 else torch.int32), x_a[i].view(torch.int16 if odt == torch.bfloat16 else torch.int32)), (name, i)
            assert torch.equal(sd[i], sd_a[i]) and torch.equal(sv[i], sv_a[i]), (name, i)
        assert torch.equal(km, km_a) and torch.equal(sc, sc_a), name
    assert bool((x_b[0][:, :h0] == 3.0).all()) and int(km_b.min()) >= 0              # projection columns untouched
    # the rows really are the table's (first and last sample, every key), not merely equal on both paths
    r = rows_a.cpu().numpy()
    t = table.cpu()
    for b in (0, B - 1):
        for k in range(Ks[0]):
            want = t[r[b * Ks[0] + k]].to(odt)
            assert torch.equal(x_b[0][b, h0 + k * E: h0 + (k + 1) * E].cpu(), want), (b, k)
    # a pair list the batch runs past is an error here, not a device fault (reference: IndexError / KeyError)
    with pytest.raises(ValueError):
        bad = [ops.StoreSide(flat[2 * (P - B + 5) + i:], 2, dense[i], cat[i], sd_c[i], sv_c[i]) for i in range(2)]
        ops.batch_ingest_store([], sides_c, bad, B, None, km_c, 0, table=table)


def test_step_from_store_equals_step_on_loader_batches(tt, tmp_path):
    """An epoch driven through DevicePairLoader.step_batches (GraphedTrainStep.step_from_store: the batch gathered out of the device
    stores by the step's own hand-over launch; ragged last batch through the eager step) == the same epoch with the loader's
    batch tensors handed to GraphedTrainStep.step / the eager step, bit for bit: per-step losses and the final state.  Third leg
    (ADVICE round 3): the same two epochs with EVERY batch through the eager step and no captured step at all -- the capture's
    warm-up steps must leave no trace (preserve_state) and the replays' Adam step numbers must follow the optimiser's own count
    across the eager ragged batches (FusedAdam.peek_step): also bit for bit (pairs % batch != 0, two epochs).  Fourth leg (round 4):
    the store-fed epochs with TWO batches per graph launch (unrolled.UnrolledTrainStep.steps_from_store) -- bit for bit again."""
    from jodalrob_twotower_amd import synthetic
    from jodalrob_twotower_amd.data_loader import create_unified_bid_dataloaders
    from jodalrob_twotower_amd.graph import GraphedTrainStep
    from jodalrob_twotower_amd.optim import FusedAdam
    vn, vc = [12, 400, 7, 90], [9, 50]
    meta = synthetic.write_metadata(tmp_path / "metadata.csv", {"notice": {f"n{i}": v for i, v in enumerate(vn)}, "company": {f"c{i}": v for i, v in enumerate(vc)}})
    with open(meta, "a", encoding="utf-8") as f:
        f.write("notice,bidntceno,text,Y,,,,0,,Y,Y,,\nnotice,bidntceord,text,Y,,,,0,,Y,Y,,\ncompany,bizno,text,Y,,,,0,,Y,Y,,\n"
                "notice,amount,numeric,Y,,,,0,,,,,\ncompany,size,numeric,Y,,,,0,,,,,\n")
    schema = tt.build_torchrec_schema_from_meta(notice_table="notice", company_table="company", pair_table="bid_two_tower",
                                                pair_notice_id_cols=["bidntceno", "bidntceord"], pair_company_id_cols=["bizno"], metadata_path=str(meta))
    finals = {}
    for mode in ("tensors", "store", "eager", "store_unrolled"):
        torch.manual_seed(123)                                           # the preprocessor's frozen random projectors draw from it
        src = synthetic.SyntheticSource(900, 700, 1200, vn, vc)
        train_loader, _ = create_unified_bid_dataloaders(src, schema, batch_size=256, test_split=0.0, shuffle_seed=7, test_mode=True, pair_limit=1200, device=DEV)
        assert len(train_loader) == 5                                    # 4 full batches + 176 pairs
        first = next(iter(train_loader))
        train_loader._gen.manual_seed(7)                                 # the peek consumed a permutation: start the epoch anew
        task = tt.create_two_tower_train_task(schema.notice.categorical, schema.company.categorical, metadata_path=str(meta), categorical_embedding_dim=16,
                                              notice_dense_input_dim=first["notice"]["dense"].shape[1], company_dense_input_dim=first["company"]["dense"].shape[1],
                                              tower_hidden_dims=[64, 32], final_embedding_dim=32, dropout_rate=0.0, device=DEV, embedding_grad="sparse",
                                              score_dtype="bf16", mlp_dtype="bf16")
        torch.manual_seed(0)
        with torch.no_grad():
            for p in task.parameters():
                p.copy_(0.05 * torch.randn(p.shape, device=DEV))
        task.train()
        opt = FusedAdam.for_task(task, lr=1e-2, weight_decay=1e-5)
        if mode == "store_unrolled":                                     # two full batches per graph launch (steps_from_store), the ragged one eager
            from jodalrob_twotower_amd.unrolled import UnrolledTrainStep
            gs = UnrolledTrainStep(task, opt, first, unroll=2, warmup=1)
        else:
            gs = GraphedTrainStep(task, opt, first, warmup=1) if mode != "eager" else None

        def eager(b):
            opt.zero_grad()
            r = task(b, return_metrics=True)
            r["loss"].backward()
            opt.step()
            return r
        losses = []
        for ep in range(2):
            if mode in ("store", "store_unrolled"):
                for r in train_loader.step_batches(gs, eager):
                    losses.append(r["loss"].item())
            else:
                for b in train_loader:
                    r = gs.step(b) if (gs is not None and b["notice"]["dense"].shape[0] == 256) else eager(b)
                    losses.append(r["loss"].item())
        torch.cuda.synchronize()
        assert len(losses) == 10
        finals[mode] = (losses, {k: v.detach().cpu().clone() for k, v in task.state_dict().items()}, opt.current_step())
        if gs is not None:
            gs.close()
    assert finals["tensors"][0] == finals["store"][0] and len(set(finals["store"][0])) == 10
    for k, v in finals["tensors"][1].items():
        assert torch.equal(v, finals["store"][1][k]), k
    assert finals["eager"][2] == finals["store"][2] == 10                # ten optimiser steps, whoever ran them
    assert finals["eager"][0] == finals["store"][0], (finals["eager"][0], finals["store"][0])
    for k, v in finals["eager"][1].items():
        assert torch.equal(v, finals["store"][1][k]), k
    assert finals["store_unrolled"][2] == 10 and finals["store_unrolled"][0] == finals["store"][0], (finals["store_unrolled"][0], finals["store"][0])
    for k, v in finals["store"][1].items():
        assert torch.equal(v, finals["store_unrolled"][1][k]), k


def test_fast_evaluation_equals_batch_loop(tt, tmp_path):
    """evaluate_comprehensive over a device-resident loader (GraphedEvalStep: every full batch gathered from the stores and
    replayed, metrics summed on the device, ragged last batch eager) == the reference-shaped loop over the loader's batches
    (evaluate_single_batch per batch, Python means): every key of the reference's result, 1e-6."""
    from jodalrob_twotower_amd import synthetic
    from jodalrob_twotower_amd.data_loader import create_unified_bid_dataloaders
    vn, vc = [12, 400, 7, 90], [9, 50]
    meta = synthetic.write_metadata(tmp_path / "metadata.csv", {"notice": {f"n{i}": v for i, v in enumerate(vn)}, "company": {f"c{i}": v for i, v in enumerate(vc)}})
    with open(meta, "a", encoding="utf-8") as f:
        f.write("notice,bidntceno,text,Y,,,,0,,Y,Y,,\nnotice,bidntceord,text,Y,,,,0,,Y,Y,,\ncompany,bizno,text,Y,,,,0,,Y,Y,,\n"
                "notice,amount,numeric,Y,,,,0,,,,,\ncompany,size,numeric,Y,,,,0,,,,,\n")
    schema = tt.build_torchrec_schema_from_meta(notice_table="notice", company_table="company", pair_table="bid_two_tower",
                                                pair_notice_id_cols=["bidntceno", "bidntceord"], pair_company_id_cols=["bizno"], metadata_path=str(meta))
    torch.manual_seed(5)
    src = synthetic.SyntheticSource(900, 700, 2000, vn, vc)
    _, test_loader = create_unified_bid_dataloaders(src, schema, batch_size=256, test_split=0.5, shuffle_seed=7, test_mode=True, pair_limit=2000, device=DEV)
    assert len(test_loader) == 4                                          # 3 full batches + 232 pairs
    first = next(iter(test_loader))
    task = tt.create_two_tower_train_task(schema.notice.categorical, schema.company.categorical, metadata_path=str(meta), categorical_embedding_dim=16,
                                          notice_dense_input_dim=first["notice"]["dense"].shape[1], company_dense_input_dim=first["company"]["dense"].shape[1],
                                          tower_hidden_dims=[64, 32], final_embedding_dim=32, dropout_rate=0.1, device=DEV, score_dtype="bf16", mlp_dtype="bf16")
    task._pair_check_done = True
    ev = tt.TwoTowerEvaluator(device=DEV)
    fast = ev.evaluate_comprehensive(task, test_loader, verbose=False)
    assert ev._fast_eval(task, test_loader) is not None and fast["num_batches"] == 4
    per = [ev.evaluate_single_batch(task, b, verbose=False) for b in test_loader]
    for k in ("loss", "accuracy", "recall@5", "recall@10", "mrr", "similarity_gap", "positive_similarity_mean", "negative_similarity_mean"):
        want = sum(m[k] for m in per) / len(per)
        assert fast[k] == pytest.approx(want, rel=1e-6, abs=1e-7), (k, fast[k], want)
    two = ev.evaluate_comprehensive(task, test_loader, verbose=False, max_batches=2)           # the cached graph again, two batches
    assert two["num_batches"] == 2 and two["loss"] == pytest.approx(sum(m["loss"] for m in per[:2]) / 2, rel=1e-6)
    ev.close()


def test_bench_multi_gpu_path_with_two_ranks_on_one_gpu():
    """`bench.py --gpus 2` end to end with two REAL ranks sharing the one GPU (collectives staged through gloo, the sharded step captured in
    segments): the weak leg incl. everything the bench does around a captured step -- bucket calibration over the batch pool, the lookup
    stamps, the dispatch-overhead calibration that re-launches the step's lookup, the roofline object, teardown before the process group
    goes.  Round 4 found that calibration indexing a rank's table SHARD with global key offsets (a GPU memory fault on every N > 1 run,
    harmless at world 1); this is the regression test the suite lacked: rc 0, ONE JSON line with the N > 1 metric text, no fault."""
    import os
    import socket
    root = ROOT
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()      # a free rendezvous port
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--rows-notice", "400000", "--rows-company",
                        "200000", "--no-cpu-baseline", "--dist-segmented", "--legs", "weak", "--pool", "3"], capture_output=True, text=True, timeout=600, env=env, cwd=str(root))
    assert "Memory access fault" not in r.stderr and "Memory access fault" not in r.stdout, r.stderr[-2000:]
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "PER GPU" in d["metric"]
    assert d["config"]["launch"].startswith("segmented graph replay") and d["roofline"]["mean_launch_us"] > 0
