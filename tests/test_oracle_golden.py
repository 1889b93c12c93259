"""Pins the numpy oracle against vectors produced by the reference's own code (tests/golden)."""
import json

import numpy as np
import pytest

import oracle_np as O
from conftest import GOLD, load_case, split_prefix
from params_init import init_state_numpy

CASES = ["tiny_train", "tiny_eval", "deep_temp", "wide_b40", "single_hidden"]
RTOL, ATOL = 2e-5, 2e-6     # fp32 restatement vs fp32 torch (different summation order)


def run(case, cfg, state=None):
    g = load_case(case)
    state = state or split_prefix(g, "state.")
    batch = split_prefix(g, "in.")
    out = O.task_step(state, batch, cfg["keys_n"], cfg["keys_c"], cfg["vocab_n"], cfg["vocab_c"],
                      temperature=cfg["T"], train=cfg["train"], backward=cfg["train"])
    return g, out


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_reference(case, manifest):
    cfg = manifest["cases"][case]
    g, out = run(case, cfg)
    np.testing.assert_allclose(out["notice_emb"], g["out.notice_emb"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out["company_emb"], g["out.company_emb"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out["sim"], g["sim"], rtol=RTOL, atol=5e-6)
    np.testing.assert_allclose(out["loss"], g["out.loss"], rtol=RTOL)
    assert float(out["accuracy"]) == float(g["out.accuracy"])
    for k in ("positive_similarity_mean", "negative_similarity_mean", "similarity_gap"):
        np.testing.assert_allclose(out[k], g["out." + k], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("case", [c for c in CASES if c != "tiny_eval"])
def test_backward_matches_reference(case, manifest):
    cfg = manifest["cases"][case]
    g, out = run(case, cfg)
    ref = split_prefix(g, "grad.")
    assert set(ref) == set(out["grads"])
    for k, v in ref.items():
        np.testing.assert_allclose(out["grads"][k], v, rtol=2e-4, atol=2e-7, err_msg=k)
    # BN running stats after the train-mode forward
    for k, v in split_prefix(g, "state_after.").items():
        np.testing.assert_allclose(out["bn_updates"][k], v, rtol=RTOL, atol=ATOL, err_msg=k)


def test_clamp_pinned():
    # ids [-5, 40000, 1000] on vocabs [12, 28993, 1000] -> [0, 28992, 999]
    got = O.unpack_clamp_ids(np.array([-5, 40000, 1000]), [12, 28993, 1000])
    assert got.tolist() == [[0, 28992, 999]]
    g = load_case("tiny_train")
    ids = O.unpack_clamp_ids(g["in.notice_ids"].reshape(-1), [12, 15, 27, 50, 1000])
    assert ids.min() >= 0 and (ids <= np.array([11, 14, 26, 49, 999])).all()
    assert ids[0, 0] == 0 and ids[1, -1] == 999 and ids[2, 1] == 14


def test_kjt_wire_format():
    z = np.load(GOLD / "kjt_wire.npz")
    v, l = O.build_batch_kjt_values(z["ids"])
    assert np.array_equal(v, z["values"]) and np.array_equal(l, z["lengths"])


def test_real_schema_case(manifest, schema_real):
    cfg = manifest["cases"]["real_schema"]
    g = load_case("real_schema")
    shapes = {k: tuple(v) for k, v in manifest["state_dict_keys_real"].items()}
    state = init_state_numpy(shapes, cfg["seed"])
    assert sum(int(np.prod(s)) for k, s in shapes.items() if "running" not in k and "num_batches" not in k) \
        == cfg["n_params"] == 2204832
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    out = O.task_step(state, split_prefix(g, "in."), kn, kc, vn, vc, temperature=1.0, train=True)
    np.testing.assert_allclose(out["loss"], g["out.loss"], rtol=RTOL)
    np.testing.assert_allclose(out["sim"], g["sim"], rtol=RTOL, atol=5e-6)
    for k, v in split_prefix(g, "grad.").items():
        if k.endswith(".rows"):
            base = k[:-5]
            dense = out["grads"][base]
            nz = np.flatnonzero(np.abs(dense).sum(axis=1))
            assert np.array_equal(nz, v), base                      # bit-exact touched-row set
            np.testing.assert_allclose(dense[nz], g["grad." + base + ".vals"], rtol=2e-4, atol=2e-8, err_msg=base)
        elif not k.endswith(".vals"):
            np.testing.assert_allclose(out["grads"][k], v, rtol=3e-4, atol=3e-8, err_msg=k)


def test_sparse_grad_equals_dense(manifest):
    cfg = manifest["cases"]["wide_b40"]
    g, out = run("wide_b40", cfg)
    E = cfg["E"]
    for side, keys, vocab, pre in (("notice", cfg["keys_n"], cfg["vocab_n"], O.NT), ("company", cfg["keys_c"], cfg["vocab_c"], O.CT)):
        offs = np.concatenate([[0], np.cumsum(vocab)[:-1]])
        rows, vals = O.embed_grad_sparse(out["d_concat_" + side], out["ids_" + side], offs, E)
        fused = np.concatenate([out["grads"][f"{pre}categorical_embedder.embeddings.{k}.weight"] for k in keys])
        assert np.array_equal(rows, np.flatnonzero(np.abs(fused).sum(1) > 0)) or len(rows) >= np.count_nonzero(np.abs(fused).sum(1))
        np.testing.assert_allclose(fused[rows], vals, rtol=1e-5, atol=1e-8)
        mask = np.ones(len(fused), bool); mask[rows] = False
        assert not fused[mask].any()


def test_adam_trajectory(manifest):
    cfg = {**manifest["cases"]["tiny_train"], **manifest["cases"]["adam_trajectory"]}
    z = np.load(GOLD / "adam_trajectory.npz")
    state = {k[6:]: z[k].copy() for k in z.files if k.startswith("state.")}
    pkeys = [k for k in state if "running" not in k and "num_batches" not in k]
    m = {k: np.zeros_like(state[k]) for k in pkeys}
    v = {k: np.zeros_like(state[k]) for k in pkeys}
    for s in range(int(z["n_steps"])):
        batch = {k[len(f"step{s}.in."):]: z[k] for k in z.files if k.startswith(f"step{s}.in.")}
        out = O.task_step(state, batch, cfg["keys_n"], cfg["keys_c"], cfg["vocab_n"], cfg["vocab_c"], 1.0, True)
        np.testing.assert_allclose(out["loss"], z[f"step{s}.loss"], rtol=5e-5)
        lr = O.warmup_lr(cfg["lr"], s, cfg["warmup_steps"])
        np.testing.assert_allclose(lr, z[f"step{s}.lr"], rtol=1e-12)
        for k in pkeys:
            O.adam_step(state[k], out["grads"][k], m[k], v[k], s + 1, lr, wd=cfg["weight_decay"])
        state.update(out["bn_updates"])
    for k in state:
        np.testing.assert_allclose(state[k], z["final." + k], rtol=2e-4, atol=2e-6, err_msg=k)


def test_eval_metrics_and_topk():
    for case in CASES:
        g = load_case(case)
        S = g["sim"]
        np.testing.assert_allclose(O.recall_at_k(S, 5), g["eval.recall@5"], rtol=1e-6)
        np.testing.assert_allclose(O.recall_at_k(S, 10), g["eval.recall@10"], rtol=1e-6)
        np.testing.assert_allclose(O.mrr(S), g["eval.mrr"], rtol=1e-6)
    g = load_case("tiny_eval")
    vals, idx = O.topk_rows(g["sim"], 5)
    assert np.array_equal(idx, g["predict.top_indices"])
    np.testing.assert_allclose(vals, g["predict.top_similarities"], rtol=1e-6)


def test_projector_and_id_mappings():
    z = np.load(GOLD / "projector.npz")
    ps = {k[6:]: z[k] for k in z.files if k.startswith("state.")}
    got = O.project_features(ps, z["numeric"], {"ntitle": z["text_ntitle"]}, ["ntitle"])
    np.testing.assert_allclose(got, z["dense_projected"], rtol=2e-5, atol=2e-6)
    cs = {k[7:]: z[k] for k in z.files if k.startswith("cstate.")}
    got = O.project_features(cs, z["c_numeric"], {}, [])
    np.testing.assert_allclose(got, z["c_dense_projected"], rtol=2e-5, atol=2e-6)
    j = json.loads((GOLD / "id_mappings.json").read_text())
    n2i, c2i = O.build_id_mappings({"notice": {"ids": [tuple(t) for t in j["notice_ids"]]},
                                    "company": {"ids": j["company_ids"]}})
    assert n2i == {tuple(k): v for k, v in j["notice_id_to_idx"]}
    assert c2i == {k: v for k, v in j["company_id_to_idx"]}


def test_q_bf16_is_round_to_nearest_even():
    """the oracle's operand rounding == torch's float32 -> bfloat16 conversion (RNE), incl. ties, subnormal-ish and large values"""
    import torch
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, 20000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.0e38, -3.0e38, 1e-40], dtype=np.float32)])
    want = torch.from_numpy(x).bfloat16().float().numpy()
    assert np.array_equal(O.q_bf16(x), want)
    assert O.q_bf16(x.astype(np.float64)).dtype == np.float64 and np.array_equal(O.q_bf16(x.astype(np.float64)), want.astype(np.float64))


def test_rounded_oracle_reduces_to_plain_oracle_on_bf16_exact_inputs(manifest):
    """rounding="bf16" changes nothing but the operands: on a state / batch whose every GEMM operand is already exactly
    representable in bf16 the forward up to the first non-representable intermediate is identical; and the plain path is
    untouched by the hooks (q=None)."""
    cfg = manifest["cases"]["tiny_train"]
    g = load_case("tiny_train")
    state, b = split_prefix(g, "state."), split_prefix(g, "in.")
    plain = O.task_step(state, b, cfg["keys_n"], cfg["keys_c"], cfg["vocab_n"], cfg["vocab_c"], cfg["T"], True, dtype=np.float64)
    rounded = O.task_step(state, b, cfg["keys_n"], cfg["keys_c"], cfg["vocab_n"], cfg["vocab_c"], cfg["T"], True, dtype=np.float64,
                          rounding="bf16")
    np.testing.assert_allclose(plain["loss"], g["out.loss"], rtol=1e-5)
    assert abs(rounded["loss"] - plain["loss"]) / plain["loss"] < 2e-2              # bf16 operands: close, not equal
    assert rounded["loss"] != plain["loss"]
    for k, v in plain["grads"].items():
        assert rounded["grads"][k].shape == v.shape


@pytest.mark.parametrize("case", CASES)
def test_torch_restatement_matches_reference(case, manifest):
    """oracle/oracle_torch.py (the vectorised torch-CPU restatement bench.py times as `cpu_baseline`) against the same
    reference-generated vectors: loss, similarity matrix, every gradient, BN running statistics."""
    import torch
    import oracle_torch as OT
    cfg = manifest["cases"][case]
    g = load_case(case)
    state = OT.make_state(split_prefix(g, "state."))
    batch = {k: torch.as_tensor(v) for k, v in split_prefix(g, "in.").items()}
    loss, S = OT.task_loss(state, batch, cfg["keys_n"], cfg["keys_c"], cfg["vocab_n"], cfg["vocab_c"], cfg["T"], cfg["train"])
    np.testing.assert_allclose(loss.item(), g["out.loss"], rtol=RTOL)
    np.testing.assert_allclose(S.detach().numpy(), g["sim"], rtol=RTOL, atol=5e-6)
    if cfg["train"]:
        loss.backward()
        for k, v in split_prefix(g, "grad.").items():
            np.testing.assert_allclose(state[k].grad.numpy(), v, rtol=2e-4, atol=2e-7, err_msg=k)
        for k, v in split_prefix(g, "state_after.").items():
            np.testing.assert_allclose(state[k].numpy(), v, rtol=RTOL, atol=ATOL, err_msg=k)


def test_q_e4m3_is_ocp_fp8_rounding():
    """the oracle's fp8 operand rounding == torch's float32 -> float8_e4m3fn conversion (RNE, subnormals) on values inside the
    format's range (the score operands are <= 64 * 1.4427 * 80 in magnitude only at extreme temperatures; unit rows: <= 93)"""
    import torch
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(2.0) ** rng.integers(-12, 8, 20000).astype(np.float32),
                        np.array([0.0, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -9, 0.0625, 1.0, 1.0625, 1.125, 1.1875, 240.0, 448.0, -448.0, 3.0e-4],
                                 dtype=np.float32)])
    x = x[np.abs(x) <= 448.0]
    want = torch.from_numpy(x).to(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(O.q_e4m3(x), want)


LOSS_VARIANTS = ["loss_smooth_0p1", "loss_smooth_0p3_temp", "loss_cosine", "loss_cosine_temp"]


@pytest.mark.parametrize("case", LOSS_VARIANTS)
def test_loss_variants_match_reference(case):
    """Label-smoothed cross-entropy and the cosine-embedding loss (two_tower_train_task.py:114-160) against vectors the
    reference produced (oracle/gen_golden_loss_variants.py).  The cosine loss is F.cosine_embedding_loss on 1-vectors:
    cos = s / sqrt((s^2 + 1e-12)(1 + 1e-12)) is a step function of s whose derivative is O(1) only for |s| < 1e-4, so its
    gradients hang on the last bits of S: loss to rounding, gradients to 5 % of the largest gradient entry."""
    cfg = json.loads((GOLD / "loss_variants.json").read_text())["cases"][case]
    g = load_case(case)
    out = O.task_step(split_prefix(g, "state."), split_prefix(g, "in."), cfg["keys_n"], cfg["keys_c"], cfg["vocab_n"], cfg["vocab_c"],
                      temperature=cfg["T"], train=True, backward=True, loss_type=cfg["loss_type"],
                      label_smoothing=cfg["label_smoothing"])
    np.testing.assert_allclose(out["loss"], g["out.loss"], rtol=2e-6)
    np.testing.assert_allclose(out["sim"], g["sim"], rtol=RTOL, atol=5e-6)
    assert float(out["accuracy"]) == float(g["out.accuracy"])
    ref = split_prefix(g, "grad.")
    assert set(ref) == set(out["grads"])
    gmax = max(float(np.abs(v).max()) for v in ref.values())
    for k, v in ref.items():
        if cfg["loss_type"] == "cosine_embedding":
            assert np.abs(out["grads"][k] - v).max() <= 5e-2 * gmax + 1e-6, k
        else:
            np.testing.assert_allclose(out["grads"][k], v, rtol=2e-4, atol=2e-7, err_msg=k)
