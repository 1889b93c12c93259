"""CPU test of the drop-in claim: every attribute, call shape, dict key, import path and signature that the REFERENCE's
callers (scripts/train.py, src/evaluation/evaluator.py, the towers' use of the id container) use on the objects this
repository replaces exists here with a compatible signature.

The expectations come from tests/golden/api_surface.json, produced by oracle/gen_api_surface.py (an `ast` walk of the
reference in the build container; data only).  Nothing here needs a GPU: objects are built on "cpu" and no forward runs
-- except the evaluator's printers / demo, which are driven with a stub model.
"""
import contextlib
import importlib
import inspect
import io
import json
import sys

import pytest
import torch

from conftest import GOLD, ROOT

import jodalrob_twotower_amd as tt
from jodalrob_twotower_amd import data_loader

SURFACE = json.loads((GOLD / "api_surface.json").read_text(encoding="utf-8"))
DROPIN = ROOT / "jodalrob-twotower_amd" / "dropin"


@pytest.fixture(scope="module")
def objects(schema_syn):
    kw = dict(notice_table="notice", company_table="company", pair_table="bid_two_tower",
              pair_notice_id_cols=["bidntceno", "bidntceord"], pair_company_id_cols=["bizno"])
    schema = tt.build_torchrec_schema_from_meta(metadata_path=GOLD / "synthetic_metadata.csv", **kw)
    task = tt.create_two_tower_train_task(schema.notice.categorical, schema.company.categorical,
                                          metadata_path=str(GOLD / "synthetic_metadata.csv"), categorical_embedding_dim=4,
                                          notice_dense_input_dim=8, company_dense_input_dim=8, tower_hidden_dims=[8, 8],
                                          final_embedding_dim=8, device="cpu")
    kjt = tt.build_batch_kjt(torch.zeros(3, len(schema.notice.categorical), dtype=torch.int64), schema.notice.categorical)
    return {"task": task, "evaluator": tt.TwoTowerEvaluator(device="cpu"), "schema": schema, "kjt": kjt,
            "loader": data_loader.DevicePairLoader, "result": None, "metrics": None, "predictions": None}


def _resolve(obj, dotted):
    for part in dotted.split("."):
        if inspect.isroutine(obj) or isinstance(obj, type(None)):          # "values.device": attribute of a call's RESULT
            return "result-of-call"
        obj = getattr(obj, part)
    return obj


def _accepts(fn, n_pos, kws):
    """True if fn(*n_pos positional, **kws) binds."""
    try:
        inspect.signature(fn).bind(*([None] * n_pos), **{k: None for k in kws})
        return True
    except TypeError:
        return False


@pytest.mark.parametrize("caller", sorted(SURFACE["uses"]))
def test_every_use_of_the_callers_resolves(caller, objects):
    for role, use in SURFACE["uses"][caller].items():
        target = objects[role]
        if target is None:                                   # dict-like roles: checked by key in the tests below
            continue
        for attr in use["attributes"]:
            head = attr.split(".")[0]
            assert hasattr(target, head), f"{caller}: {role}.{attr} is used by the reference and missing here"
            _resolve(target, attr)
        for name, shapes in use["calls"].items():
            fn = getattr(target, name)
            for n_pos, kws in shapes:
                assert _accepts(fn, n_pos, kws), f"{caller}: {role}.{name}({n_pos} positional, {kws}) does not bind to {inspect.signature(fn)}"
        for n_pos, kws in use["called_directly"]:
            assert _accepts(target.forward, n_pos, kws), f"{caller}: {role}(...) with {n_pos} positional, {kws}"


def test_result_metric_and_prediction_keys():
    """Keys the callers read out of the task's result dict, the evaluator's metric dicts and predict_batch's dict."""
    want = {"result": set(), "metrics": set(), "predictions": set()}
    for uses in SURFACE["uses"].values():
        for role in want:
            want[role] |= set(uses.get(role, {}).get("keys", []))
    assert want["result"] == {"loss", "accuracy", "positive_similarity_mean", "negative_similarity_mean", "similarity_gap", "similarity_matrix"}
    assert want["predictions"] == {"top_similarities", "top_indices", "all_similarities"}
    ev = tt.TwoTowerEvaluator(device="cpu")
    basic = {"loss": 1.0, "accuracy": 0.5, "similarity_gap": 0.1, "positive_similarity_mean": 0.2, "negative_similarity_mean": 0.1}
    single = ev.metrics_from_ranks(torch.tensor([0, 3, 11, 1]), basic)
    assert want["metrics"] - {"num_batches"} <= set(single), want["metrics"] - set(single)
    # the product's own sources name the same keys (forward's result dict and predict_batch are GPU paths: check the source text)
    src = inspect.getsource(sys.modules[tt.TwoTowerTrainTask.__module__])
    for k in want["result"] | want["predictions"]:
        assert f'"{k}"' in src, k


def test_import_paths_of_the_reference_driver_resolve():
    sys.path.insert(0, str(DROPIN))
    try:
        for caller, imports in SURFACE["imports"].items():
            for imp in imports:
                mod = importlib.import_module(imp["module"])
                for name in imp["names"]:
                    assert hasattr(mod, name), f"{caller}: from {imp['module']} import {name}"
    finally:
        sys.path.remove(str(DROPIN))
        for m in [m for m in sys.modules if m == "src" or m.startswith("src.")]:
            del sys.modules[m]


def _ours(path, name):
    return {"src/evaluation/evaluator.py": tt.TwoTowerEvaluator, "TwoTowerTrainTask": tt.TwoTowerTrainTask,
            "create_two_tower_train_task": tt.create_two_tower_train_task, "TwoTowerModel": tt.TwoTowerModel,
            "create_two_tower_model": tt.create_two_tower_model, "build_torchrec_schema_from_meta": tt.build_torchrec_schema_from_meta,
            "create_unified_bid_dataloaders": data_loader.create_unified_bid_dataloaders, "FeaturePreprocessor": tt.FeaturePreprocessor,
            "FeatureProjector": tt.FeatureProjector, "TwoTowerEvaluator": tt.TwoTowerEvaluator}[name]


def _check_signature(where, ref_sig, fn):
    """Ours must take the reference's parameters under the same names, in the same order, with the same defaults; it may
    append further parameters only if they have defaults."""
    sig = inspect.signature(fn)
    ours = [p for p in sig.parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    ours_kw = {p.name: p for p in sig.parameters.values() if p.kind == p.KEYWORD_ONLY}
    ref = [p for p in ref_sig["params"] if p["name"] != "self"]
    ours = [p for p in ours if p.name != "self"]
    assert [p.name for p in ours[:len(ref)]] == [p["name"] for p in ref], f"{where}: {[p.name for p in ours]} vs {[p['name'] for p in ref]}"
    for mine, theirs in zip(ours, ref):
        if theirs["default"] is None:
            continue                                              # required in the reference: either is compatible
        assert mine.default is not inspect.Parameter.empty, f"{where}: {mine.name} has a default in the reference"
        assert mine.default == eval(theirs["default"], {}), f"{where}: default of {mine.name}: {mine.default!r} vs {theirs['default']}"
    for extra in ours[len(ref):]:
        assert extra.default is not inspect.Parameter.empty, f"{where}: extra parameter {extra.name} needs a default"
    for p in ref_sig["kwonly"]:
        assert p["name"] in ours_kw or p["name"] in {q.name for q in ours}, f"{where}: keyword-only {p['name']}"
        if p["default"] is not None and p["name"] in ours_kw:
            assert ours_kw[p["name"]].default == eval(p["default"], {}), f"{where}: default of {p['name']}"


def test_signatures_accept_the_references_arguments():
    n = 0
    for path, defs in SURFACE["signatures"].items():
        for name, ref in defs.items():
            target = _ours(path, name)
            if "methods" in ref:
                for meth, sig in ref["methods"].items():
                    assert hasattr(target, meth), f"{path}: {name}.{meth} missing"
                    _check_signature(f"{name}.{meth}", sig, getattr(target, meth))
                    n += 1
            else:
                _check_signature(name, ref, target)
                n += 1
    assert n >= 25


class _StubTask:
    """Stands in for the task on CPU: predict_batch / eval with the reference's contract (:181-207)."""

    def __init__(self, sim):
        self.sim, self.evals = sim, 0

    def eval(self):
        self.evals += 1

    def predict_batch(self, batch, top_k=10):
        v, i = torch.topk(self.sim, k=min(top_k, self.sim.size(1)), dim=1)
        return {"top_similarities": v, "top_indices": i, "all_similarities": self.sim}


def test_printers_and_demo_run_and_say_what_the_reference_says():
    ev = tt.TwoTowerEvaluator(device="cpu")
    g = torch.Generator().manual_seed(3)
    sim = torch.rand(12, 12, generator=g)
    basic = {"loss": 2.4849, "accuracy": 0.25, "similarity_gap": 0.6, "positive_similarity_mean": 0.7, "negative_similarity_mean": 0.1}
    ranks = (sim > sim.diag()[:, None]).sum(1)
    m = ev.metrics_from_ranks(ranks, basic)
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        ev.print_single_batch_results(m)
        ev.print_comprehensive_results({**m, "num_batches": 7})
        ev.print_performance_assessment({"accuracy": 0.31, "recall@10": 0.39, "similarity_gap": 0.5})
        task = _StubTask(sim)
        ev.demonstrate_predictions(task, {"notice": {"dense": torch.zeros(12, 4)}}, top_k=10)
    text = out.getvalue()
    assert task.evals == 1
    for line in ("배치 크기: 12", "Loss: 2.4849", "Top-1 Accuracy: 0.250", f"MRR: {m['mrr']:.3f}", "랜덤 Top-1 정확도: 0.083 | 현재: 0.250 (개선)",
                 "테스트 배치 수: 7", "평균 Loss: 2.4849", "Top-1 정확도: 보통 (0.15~0.3)", "Recall@10: 실용적 수준 (0.6 이상)",
                 "유사도 구분: 양호 (0.5 이상)", "Top-1 정확도: 우수 (0.3 이상)", "Recall@10: 부족 (0.4 미만)", "유사도 구분: 개선 필요 (0.5 미만)",
                 "--- 추론 예제 ---", "유사도 행렬 크기: torch.Size([12, 12])"):
        assert line in text, line


def test_driver_writes_the_references_results_csv_and_checkpoint(tmp_path):
    """scripts/train.py's own output contracts: the results CSV has the reference's columns in the reference's order and
    default file name (scripts/train.py:24-57), its hyper-parameter / metric dicts carry the reference's keys (:458-485), the
    checkpoint dict its four keys (:506-511).  The reference side comes from the `ast` walk in tests/golden/api_surface.json
    ("harness"); this side from importing this repository's driver (no GPU is touched at import) and writing one row."""
    import ast
    import csv
    import importlib.util
    import inspect
    from conftest import ROOT
    h = SURFACE["harness"]["scripts/train.py"]
    spec = importlib.util.spec_from_file_location("_tt_train_driver", ROOT / "scripts" / "train.py")
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    cols = [c["column"] for c in h["results_csv"]["columns"]]
    assert drv.RESULT_COLUMNS == cols
    sig = inspect.signature(drv.save_training_results)
    assert list(sig.parameters) == ["hyperparams", "metrics", "output_file"] and sig.parameters["output_file"].default == h["results_csv_default_file"]
    from_hp = {c["column"] for c in h["results_csv"]["columns"] if c["read_from"] and c["read_from"][0] == "hyperparams"}
    assert drv._FROM_HYPERPARAMS == from_hp
    out = tmp_path / "r.csv"
    hp = {k: i for i, k in enumerate(h["hyperparams_keys"])}
    hp["hidden_dims"] = [512, 256]
    fm = {k: 0.5 + i for i, k in enumerate(h["final_metrics_keys"])}           # the reference's keys: "recall@5", not "recall_at_5"
    drv.save_training_results(hp, fm, str(out))
    drv.save_training_results(hp, {}, str(out))
    with open(out, newline="", encoding="utf-8") as f:
        rows = list(csv.reader(f))
    assert rows[0] == cols and len(rows) == 3
    first = dict(zip(cols, rows[1]))
    assert first["hidden_dims"] == "[512, 256]" and first["recall_at_5"] == str(fm["recall@5"]) and first["train_loss"] == str(fm["train_loss"])
    assert dict(zip(cols, rows[2]))["val_loss"] == "N/A"                       # the reference's `.get(key, "N/A")`
    # the dict literals the driver builds carry the reference's keys
    src = ast.parse((ROOT / "scripts" / "train.py").read_text(encoding="utf-8"))
    lits = {}
    for node in ast.walk(src):
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) and isinstance(node.value, ast.Dict):
            lits[node.targets[0].id] = [k.value for k in node.value.keys if isinstance(k, ast.Constant)]
    assert lits["hyperparams"] == h["hyperparams_keys"] and lits["final_metrics"] == h["final_metrics_keys"]
    assert lits["ckpt"] == h["checkpoint_keys"]
    missing = [k for k in ("batch_size", "test_split", "shuffle_seed", "pair_limit", "categorical_embedding_dim", "notice_dense_input_dim",
                           "company_dense_input_dim", "tower_hidden_dims", "final_embedding_dim", "dropout_rate", "temperature", "loss_type",
                           "learning_rate", "weight_decay", "num_epochs", "warmup_ratio", "log_interval", "output_dir", "gpu_optimization")
               if k not in lits["config"] or k not in h["config_keys"]]
    assert not missing, missing
