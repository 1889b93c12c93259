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
    out = tmp_path / "models"
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "train.py"), "--entities", "2000", "--pairs", "8192", "--batch-size", "256",
                        "--steps", "12", "--output-dir", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert (out / "final_model.pt").exists() and (out / "model_weights.pt").exists() and (out / "best_model.pt").exists()
    ck = torch.load(out / "final_model.pt", map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}                    # scripts/train.py:506-511
    man = json.loads((GOLD / "manifest.json").read_text())
    assert {k: list(v.shape) for k, v in ck["model_state_dict"].items()} == man["state_dict_keys_real"]
    # resume restores model and optimiser state
    r2 = subprocess.run([sys.executable, str(ROOT / "scripts" / "train.py"), "--entities", "2000", "--pairs", "8192", "--batch-size",
                         "256", "--steps", "3", "--output-dir", str(out), "--resume", str(out / "best_model.pt")],
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
