"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
(a) golden vectors produced by the reference's own code and (b) the numpy oracle on seeded inputs.

Bars: ids / row indices / ranks / top-k indices bit-exact; looked-up rows bit-exact (f32 copy);
fp32 results rtol 2e-5 (forward), 3e-4 (gradients: different summation order), stated per assert.
"""
import json

import numpy as np
import pytest
import torch

import oracle_np as O
from conftest import GOLD, load_case, split_prefix
from params_init import init_state_numpy, synth_batch_numpy

from jodalrob_twotower_amd import _lib as _L
from jodalrob_twotower_amd import config as _cfg

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture
def ctx_option():
    """Sets a TT_OPT_* option on the device's context for one test: ctx_option(option, value, default)."""
    changed = {}

    def set_(option, value, default):
        _L.set_option(torch.device(DEV), option, value)
        changed[option] = default
    yield set_
    for option, default in changed.items():
        _L.set_option(torch.device(DEV), option, default)


@pytest.fixture(scope="module")
def tt():
    import jodalrob_twotower_amd as m
    from jodalrob_twotower_amd import _lib
    _lib.load()
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return m


def make_task(tt, cfg, meta=None, **kw):
    return tt.create_two_tower_train_task(
        cfg["keys_n"], cfg["keys_c"], metadata_path=str(meta or GOLD / "synthetic_metadata.csv"),
        categorical_embedding_dim=cfg["E"], notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"],
        tower_hidden_dims=list(cfg["hidden"]), final_embedding_dim=cfg["D"], dropout_rate=kw.pop("dropout_rate", 0.0),
        temperature=cfg["T"], device=DEV, **kw)


def to_batch(tt, b, keys_n, keys_c):
    return {"notice": {"dense": torch.from_numpy(b["notice_dense"]).to(DEV),
                       "kjt": tt.build_batch_kjt(torch.from_numpy(b["notice_ids"]), keys_n).to(DEV)},
            "company": {"dense": torch.from_numpy(b["company_dense"]).to(DEV),
                        "kjt": tt.build_batch_kjt(torch.from_numpy(b["company_ids"]), keys_c).to(DEV)}}


def load_state(task, state):
    task.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})


# ------------------------------------------------------------------------------------------- golden cases
@pytest.mark.parametrize("case", ["tiny_train", "tiny_eval", "deep_temp", "wide_b40", "single_hidden"])
def test_task_matches_reference_golden(tt, manifest, case):
    cfg = manifest["cases"][case]
    g = load_case(case)
    task = make_task(tt, cfg)
    load_state(task, split_prefix(g, "state."))
    task.train(cfg["train"])
    batch = to_batch(tt, split_prefix(g, "in."), cfg["keys_n"], cfg["keys_c"])
    res = task(batch, return_metrics=True)
    np.testing.assert_allclose(res["similarity_matrix"].cpu().numpy(), g["sim"], rtol=2e-5, atol=5e-6)
    np.testing.assert_allclose(res["loss"].item(), g["out.loss"], rtol=2e-5)
    assert res["accuracy"].item() == pytest.approx(float(g["out.accuracy"]), abs=1e-7)
    for k in ("positive_similarity_mean", "negative_similarity_mean", "similarity_gap"):
        np.testing.assert_allclose(res[k].item(), g["out." + k], rtol=1e-4, atol=2e-6)
    if cfg["train"]:
        res["loss"].backward()
        ref = split_prefix(g, "grad.")
        got = {n: p.grad for n, p in task.named_parameters()}
        assert set(ref) == set(got)
        for k, v in ref.items():
            np.testing.assert_allclose(got[k].cpu().numpy(), v, rtol=3e-4, atol=3e-7, err_msg=k)
        sd = task.state_dict()
        for k, v in split_prefix(g, "state_after.").items():
            np.testing.assert_allclose(sd[k].cpu().numpy(), v, rtol=2e-5, atol=2e-6, err_msg=k)
    # tower embeddings through the model API, unit rows
    with torch.no_grad():
        task2 = make_task(tt, cfg)
        load_state(task2, split_prefix(g, "state."))
        task2.train(cfg["train"])
        ne, ce = task2.two_tower_model(batch["notice"], batch["company"])
    np.testing.assert_allclose(ne.cpu().numpy(), g["out.notice_emb"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ce.cpu().numpy(), g["out.company_emb"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ne.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)


@pytest.mark.parametrize("case", ["loss_smooth_0p1", "loss_smooth_0p3_temp", "loss_cosine", "loss_cosine_temp"])
def test_loss_variants_match_reference_golden(tt, case):
    """loss_type="cosine_embedding" and label_smoothing != 0 (two_tower_train_task.py:114-160) run on the dense loss path
    (tt_score_dense_fwd / _bwd: materialised score matrix, f32): loss, metrics and every gradient against the reference's
    own vectors.  (Cosine gradients: see tests/test_oracle_golden.py::test_loss_variants_match_reference.)"""
    import json
    cfg = json.loads((GOLD / "loss_variants.json").read_text())["cases"][case]
    g = load_case(case)
    from jodalrob_twotower_amd import two_tower_train_task as T3
    model_task = make_task(tt, cfg, loss_type=cfg["loss_type"])
    task = T3.TwoTowerTrainTask(model_task.two_tower_model, temperature=cfg["T"], loss_type=cfg["loss_type"],
                                label_smoothing=cfg["label_smoothing"])
    load_state(task, split_prefix(g, "state."))
    task.train()
    res = task(to_batch(tt, split_prefix(g, "in."), cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
    np.testing.assert_allclose(res["loss"].item(), g["out.loss"], rtol=2e-5)
    np.testing.assert_allclose(res["similarity_matrix"].cpu().numpy(), g["sim"], rtol=2e-5, atol=5e-6)
    assert res["accuracy"].item() == pytest.approx(float(g["out.accuracy"]), abs=1e-7)
    for k in ("positive_similarity_mean", "negative_similarity_mean", "similarity_gap"):
        np.testing.assert_allclose(res[k].item(), g["out." + k], rtol=1e-4, atol=2e-6)
    res["loss"].backward()
    ref = split_prefix(g, "grad.")
    got = {n: p.grad.cpu().numpy() for n, p in task.named_parameters()}
    assert set(ref) == set(got)
    gmax = max(float(np.abs(v).max()) for v in ref.values())
    for k, v in ref.items():
        if cfg["loss_type"] == "cosine_embedding":
            assert np.abs(got[k] - v).max() <= 5e-2 * gmax + 1e-6, (k, np.abs(got[k] - v).max(), gmax)
        else:
            np.testing.assert_allclose(got[k], v, rtol=3e-4, atol=3e-7, err_msg=k)


def test_dense_loss_second_backward_over_retained_graph(tt):
    """The dense loss node's backward leaves its saved score matrix alone: two backward passes over one retained graph
    (gradient accumulation over several losses) give the same gradients -- tt_score_dense_bwd writes dS through a raw
    pointer, which autograd's version counters cannot see."""
    from jodalrob_twotower_amd.two_tower_train_task import _DenseLossFn
    g = torch.Generator(device=DEV).manual_seed(11)
    for loss_type, ls in ((0, 0.1), (1, 0.0)):
        n = torch.nn.functional.normalize(torch.randn(70, 24, generator=g, device=DEV), dim=1).requires_grad_()
        c = torch.nn.functional.normalize(torch.randn(70, 24, generator=g, device=DEV), dim=1).requires_grad_()
        loss, _ = _DenseLossFn.apply(n, c, 2.0, loss_type, ls)
        loss.backward(retain_graph=True)
        g1n, g1c = n.grad.clone(), c.grad.clone()
        n.grad = c.grad = None
        loss.backward()
        assert torch.equal(n.grad, g1n) and torch.equal(c.grad, g1c)
        assert float(g1n.abs().max()) > 0


def test_torch_compile_wrapper_survives(tt, manifest):
    """scripts/train.py:223-225 optionally wraps the task in torch.compile(mode="reduce-overhead").  The tracer cannot see
    through ctypes calls, so the modules keep it out (_lib.no_dynamo): the compiled wrapper runs the same HIP step --
    loss equal to the golden value, gradients present."""
    cfg = manifest["cases"]["tiny_train"]
    g = load_case("tiny_train")
    task = make_task(tt, cfg)
    load_state(task, split_prefix(g, "state."))
    task.train()
    batch = to_batch(tt, split_prefix(g, "in."), cfg["keys_n"], cfg["keys_c"])
    compiled = torch.compile(task, mode="reduce-overhead", fullgraph=False)
    for _ in range(2):
        task.zero_grad()
        res = compiled(batch, return_metrics=True)
        res["loss"].backward()
    np.testing.assert_allclose(res["loss"].item(), g["out.loss"], rtol=2e-5)
    ref = split_prefix(g, "grad.")
    for n_, p in task.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref[n_], rtol=3e-4, atol=3e-7, err_msg=n_)


def test_real_schema_golden(tt, manifest, schema_real):
    cfg = dict(manifest["cases"]["real_schema"])
    cfg.update(keys_n=schema_real["notice"]["categorical"], keys_c=schema_real["company"]["categorical"])
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    g = load_case("real_schema")
    meta = GOLD / "real_vocab_metadata.csv"          # real key names + category counts (fixture)
    task = make_task(tt, cfg, meta=meta)
    shapes = {k: tuple(v) for k, v in manifest["state_dict_keys_real"].items()}
    assert {k: tuple(v.shape) for k, v in task.state_dict().items()} == shapes      # drop-in state-dict layout
    assert sum(p.numel() for p in task.parameters()) == 2204832
    load_state(task, init_state_numpy(shapes, cfg["seed"]))
    task.train()
    res = task(to_batch(tt, split_prefix(g, "in."), cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
    np.testing.assert_allclose(res["loss"].item(), g["out.loss"], rtol=2e-5)
    np.testing.assert_allclose(res["similarity_matrix"].cpu().numpy(), g["sim"], rtol=2e-5, atol=5e-6)
    res["loss"].backward()
    got = {n: p.grad.cpu().numpy() for n, p in task.named_parameters()}
    for k, v in split_prefix(g, "grad.").items():
        if k.endswith(".rows"):
            base = k[:-5]
            nz = np.flatnonzero(np.abs(got[base]).sum(axis=1))
            assert np.array_equal(nz, v), base                       # touched-row set bit-exact
            np.testing.assert_allclose(got[base][nz], g["grad." + base + ".vals"], rtol=3e-4, atol=3e-8, err_msg=base)
        elif not k.endswith(".vals"):
            np.testing.assert_allclose(got[k], v, rtol=5e-4, atol=5e-8, err_msg=k)


# ------------------------------------------------------------------------------------------- kernels vs oracle
@pytest.mark.parametrize("E,B,out_dtype", [(32, 257, "f32"), (32, 1024, "bf16"), (8, 33, "f32"), (6, 19, "f32"), (64, 5, "f32")])
def test_lookup_bit_exact(tt, E, B, out_dtype):
    from jodalrob_twotower_amd import ops
    rng = np.random.default_rng(E * 1000 + B)
    vocabs = [[12, 300, 7, 5000], [3, 64]]
    offs, rows = [], 0
    for v in vocabs:
        offs.append(np.concatenate([[0], np.cumsum(v)[:-1]]) + rows)
        rows += sum(v)
    table = rng.standard_normal((rows, E)).astype(np.float32)
    t_table = torch.from_numpy(table).to(DEV)
    sides, ids_all, outs = [], [], []
    h0 = 16
    for v, off in zip(vocabs, offs):
        K = len(v)
        ids = np.stack([rng.integers(-3, vk + 3, B) for vk in v], axis=1).astype(np.int64)       # incl. out-of-range
        ids_all.append(ids)
        out = torch.zeros((B, h0 + K * E), dtype=torch.float32 if out_dtype == "f32" else torch.bfloat16, device=DEV)
        outs.append(out)
        sides.append(ops.LookupSide(torch.from_numpy(ids.reshape(-1)).to(DEV), torch.from_numpy(off.astype(np.int64)).to(DEV),
                                    torch.tensor(v, dtype=torch.int64, device=DEV), out[:, h0:], K))
    rows_out = ops.embed_lookup(t_table, sides, B, want_rows=True)
    exp_rows = []
    for v, off, ids, out in zip(vocabs, offs, ids_all, outs):
        cl = O.unpack_clamp_ids(ids.reshape(-1), v)
        r = cl + off[None, :]
        exp_rows.append(r.reshape(-1))
        exp = table[r.reshape(-1)].reshape(B, -1)
        got = out[:, h0:]
        if out_dtype == "f32":
            assert np.array_equal(got.cpu().numpy(), exp)                                      # bit-exact row copy
        else:
            assert torch.equal(got.cpu(), torch.from_numpy(exp).to(torch.bfloat16))            # RNE rounding
        assert not out[:, :h0].any()                                                           # projection columns untouched
    assert np.array_equal(rows_out.cpu().numpy(), np.concatenate(exp_rows).astype(np.int32))


@pytest.mark.parametrize("M,table_rows,dist", [(1, 10, "uniform"), (4096, 200, "uniform"), (4097, 70000, "uniform"),
                                                (100000, 2_000_000, "uniform"), (50000, 14_000_000, "zipf"),
                                                (30000, 3, "uniform"), (20000, 1 << 25, "zipf")])
@pytest.mark.parametrize("chained", [1, 0])
def test_dedup_plan_bit_exact(tt, ctx_option, M, table_rows, dist, chained):
    """chained: the segment heads as one launch whose tiles chain their counts through the context's buffer (TT_OPT_CHAINED,
    default) / as count + write launches."""
    from jodalrob_twotower_amd import ops
    ctx_option(_L.TT_OPT_CHAINED, chained, 1)
    rng = np.random.default_rng(M + table_rows)
    if dist == "uniform":
        rows = rng.integers(0, table_rows, M)
    else:
        rows = np.minimum(rng.zipf(1.2, M) - 1, table_rows - 1)
        rows = (rows * 2654435761) % table_rows                     # scatter ranks over the row space
    rows = rows.astype(np.int32)
    plan = ops.dedup_plan(torch.from_numpy(rows).to(DEV), table_rows)
    U = int(plan.n_unique.item())
    order = np.argsort(rows, kind="stable")
    uniq, start = np.unique(rows[order], return_index=True)
    assert U == len(uniq)
    assert np.array_equal(plan.sorted_src.cpu().numpy(), order.astype(np.int32))              # stable sort
    assert np.array_equal(plan.unique_rows[:U].cpu().numpy(), uniq)
    assert np.array_equal(plan.seg_offsets[:U + 1].cpu().numpy(), np.concatenate([start, [M]]).astype(np.int32))


def test_chained_plan_timeout_raises_device_error(tt, ctx_option):
    """a tile of the chained segment-head launch whose bounded wait expires must not leave a silently empty (or corrupt)
    plan behind.  With the wait cut to ONE poll (TT_OPT_CHAIN_SPIN) some tile of a 600-tile plan gives up: the context's sticky device
    error word is raised, tt_ctx_check_device_errors reports it (TwoTowerHipError), clears it and zeroes the chain buffers; the next plan,
    with the default wait, is the oracle's again (no stale ready bits) and the check is clean."""
    from jodalrob_twotower_amd import ops
    rng = np.random.default_rng(5)
    M, R = 2_400_000, 3_000_000
    rows = torch.from_numpy(rng.integers(0, R, M).astype(np.int32)).to(DEV)
    _L.check_device_errors(torch.device(DEV))                               # start clean
    ctx_option(_L.TT_OPT_CHAIN_SPIN, 1, 1 << 22)
    raised = 0
    for _ in range(4):                                                       # (whether a tile has to wait is a race: four tries, one error suffices)
        ops.dedup_plan(rows, R)
        try:
            _L.check_device_errors(torch.device(DEV))
        except _L.TwoTowerHipError as e:
            assert "chained segment-head launch" in str(e)
            raised += 1
    assert raised >= 1
    _L.set_option(torch.device(DEV), _L.TT_OPT_CHAIN_SPIN, 1 << 22)
    plan = ops.dedup_plan(rows, R)
    _L.check_device_errors(torch.device(DEV))                               # clean, and the plan is whole
    U = int(plan.n_unique.item())
    want = np.unique(rows.cpu().numpy())
    assert U == len(want) and np.array_equal(plan.unique_rows[:U].cpu().numpy(), want)


@pytest.mark.parametrize("E,form", [(32, "ids"), (32, "rows"), (6, "ids"), (32, "fused")])
def test_lookup_rows_outside_the_table_raise_device_error(tt, E, form):
    """Key offsets / vocabularies are device arrays: the host cannot check them against the table without a synchronisation.  A lookup
    whose decoded rows lie past the table it was handed (here: offsets of a 3000-row space against a 1000-row table -- a rank's shard
    indexed by global rows, the bench bug of round 4 that ended in a GPU memory fault) must not read there: it reads the table's last
    row, raises the sticky device error word, and tt_ctx_check_device_errors reports it.  Forms: from int64 ids (wave kernel, generic
    kernel at E = 6), the hand-over launch that precomputes the rows for tt_embed_lookup_rows_fwd (the check sits where the rows
    are formed), and the fused hand-over + lookup launch.  In-range slots still get their own rows."""
    from jodalrob_twotower_amd import ops
    dev = torch.device(DEV)
    rng = np.random.default_rng(77)
    R, B, K = 1000, 256, 3
    table = torch.from_numpy(rng.standard_normal((R, E)).astype(np.float32)).to(dev)
    off = torch.tensor([0, 500, 2000], dtype=torch.int64, device=dev)            # key 2 starts past the table
    voc = torch.tensor([500, 500, 1000], dtype=torch.int64, device=dev)
    ids = torch.from_numpy(rng.integers(0, 500, (B, K)).astype(np.int64)).to(dev).reshape(-1)
    out = torch.zeros((B, K * E), dtype=torch.float32, device=dev)
    _L.check_device_errors(dev)                                                  # start clean
    side = ops.LookupSide(ids, off, voc, out, K)
    if form == "ids":
        ops.embed_lookup(table, [side], B, want_rows=False)
    elif form == "rows":
        # the captured step's pair: the hand-over forms the fused rows (and checks them against the row space it is told), the lookup
        # from precomputed rows trusts them
        rows_km = torch.empty(B * K, dtype=torch.int32, device=dev)
        rows_sm = torch.empty(B * K, dtype=torch.int32, device=dev)
        ops.batch_ingest([], [ops.LookupSide(ids, off, voc, None, K)], B, rows_km, rows_sm=rows_sm, table_rows=R)
        assert int(rows_sm.max()) == R - 1 and int(rows_km.max()) == R - 1
        ops.embed_lookup_rows(table, rows_sm, [side], B)
    else:
        if not ops.ingest_lookup_supported(table, [side]):
            pytest.skip("fused hand-over + lookup does not take this shape")
        rows_km = torch.empty(B * K, dtype=torch.int32, device=dev)
        ops.batch_ingest([], [side], B, rows_km, table=table)
    with pytest.raises(_L.TwoTowerHipError, match="outside its table"):
        _L.check_device_errors(dev)
    got = out.view(B, K, E).cpu().numpy()
    idn = ids.view(B, K).cpu().numpy()
    tab = table.cpu().numpy()
    assert np.array_equal(got[:, 0], tab[idn[:, 0]]) and np.array_equal(got[:, 1], tab[500 + idn[:, 1]])
    assert np.array_equal(got[:, 2], np.broadcast_to(tab[R - 1], (B, E)))        # the out-of-table key: the last row, not a fault
    _L.check_device_errors(dev)                                                  # the word was cleared
    ops.embed_lookup(table, [ops.LookupSide(ids, torch.tensor([0, 500, 0], dtype=torch.int64, device=dev), voc, out, K)], B, want_rows=False)
    _L.check_device_errors(dev)                                                  # a consistent lookup leaves it clean


@pytest.mark.parametrize("E,B,vocabs,src_dtype", [(32, 2048, [[2, 2, 12, 5000], [3, 100000]], "f32"), (8, 300, [[5, 9], [4]], "f32"),
                                                   (6, 64, [[3, 1000]], "f32"), (64, 512, [[2, 300], [7]], "f32"),
                                                   (16, 700, [[2, 50]], "f32"), (32, 1500, [[3, 40, 9000], [2]], "bf16"),
                                                   (12, 200, [[2, 30]], "bf16")])
def test_embed_grad_sparse_and_dense(tt, E, B, vocabs, src_dtype):
    """lane-group widths 8 / 2 / generic / 16 / 4 of the segmented reduction, f32 and bf16 gradient sources,
    segments longer than the chunking threshold (vocab 2-3 at B >= 512)."""
    from jodalrob_twotower_amd import ops
    rng = np.random.default_rng(E + B)
    offs, rows_total = [], 0
    for v in vocabs:
        offs.append(np.concatenate([[0], np.cumsum(v)[:-1]]) + rows_total)
        rows_total += sum(v)
    srcs, rows_all, d_all = [], [], []
    for v, off in zip(vocabs, offs):
        K = len(v)
        ids = np.stack([rng.integers(0, vk, B) for vk in v], axis=1)
        d = rng.standard_normal((B, 16 + K * E)).astype(np.float32)
        td = torch.from_numpy(d).to(DEV)
        if src_dtype == "bf16":
            td = td.to(torch.bfloat16)
            d = td.float().cpu().numpy()
        srcs.append((td[:, 16:], K))
        rows_all.append((ids + off[None, :]).reshape(-1))
        d_all.append(d[:, 16:].reshape(B * K, E))
    rows = np.concatenate(rows_all).astype(np.int32)
    vals = np.concatenate(d_all)
    plan = ops.dedup_plan(torch.from_numpy(rows).to(DEV), rows_total)
    U = int(plan.n_unique.item())
    dense_ref = np.zeros((rows_total, E), np.float64)
    np.add.at(dense_ref, rows, vals.astype(np.float64))
    out = torch.full((len(rows), E), float("nan"), device=DEV)
    ops.embed_grad(plan, srcs, B, E, ops.TT_GRAD_SPARSE, out)
    uniq = plan.unique_rows[:U].cpu().numpy()
    np.testing.assert_allclose(out[:U].cpu().numpy(), dense_ref[uniq], rtol=2e-5, atol=2e-5)
    dense = torch.zeros((rows_total, E), device=DEV)
    ops.embed_grad(plan, srcs, B, E, ops.TT_GRAD_DENSE_SET, dense)
    ops.embed_grad(plan, srcs, B, E, ops.TT_GRAD_DENSE_ACC, dense)
    np.testing.assert_allclose(dense.cpu().numpy(), 2 * dense_ref, rtol=2e-5, atol=4e-5)
    # bitwise reproducible: same inputs -> identical bits
    out2 = torch.empty_like(out)
    ops.embed_grad(plan, srcs, B, E, ops.TT_GRAD_SPARSE, out2)
    assert torch.equal(out[:U], out2[:U])
    # TT_GRAD_SHORT_SEGMENTS (one lane group per row, no chunk passes) changes the time, not the sums -- also when
    # the caller's promise does not hold and some rows ARE long
    out3 = torch.full_like(out, float("nan"))
    ops.embed_grad(plan, srcs, B, E, ops.TT_GRAD_SPARSE, out3, short_segments=True)
    np.testing.assert_allclose(out3[:U].cpu().numpy(), dense_ref[uniq], rtol=2e-5, atol=2e-5)


def test_score_kernels_vs_oracle(tt):
    from jodalrob_twotower_amd import ops
    rng = np.random.default_rng(7)
    for B, D, T in [(300, 64, 1.0), (129, 16, 0.5), (64, 6, 0.25), (1000, 128, 1.0), (257, 200, 2.0)]:
        n = rng.standard_normal((B, D)).astype(np.float32)
        c = rng.standard_normal((B, D)).astype(np.float32)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        c /= np.linalg.norm(c, axis=1, keepdims=True)
        c[5] = c[3]                                            # duplicated company: exact score ties
        n[7] = n[2]
        loss, met, S, lse = O.score_ce_fwd(n.astype(np.float64), c.astype(np.float64), T)
        dN, dC = O.score_ce_bwd(n.astype(np.float64), c.astype(np.float64), S, lse, T)
        tn, tc = torch.from_numpy(n).to(DEV).requires_grad_(), torch.from_numpy(c).to(DEV).requires_grad_()
        from jodalrob_twotower_amd.two_tower_train_task import _ScoreCEFn
        l, out8, _ = _ScoreCEFn.apply(tn, tc, 1.0 / T, "fp32")
        l.backward()
        np.testing.assert_allclose(l.item(), loss, rtol=5e-6)
        np.testing.assert_allclose(tn.grad.cpu().numpy(), dN, rtol=1e-4, atol=2e-8)
        np.testing.assert_allclose(tc.grad.cpu().numpy(), dC, rtol=1e-4, atol=2e-8)
        np.testing.assert_allclose(out8[2].item(), met["positive_similarity_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(out8[3].item(), met["negative_similarity_mean"], rtol=1e-3, atol=2e-6)
        # ranks: against the f32 score matrix the kernel family itself produces (ties are bit-level)
        Sg = ops.score_matrix(tn.detach(), tc.detach(), 1.0 / T).cpu().numpy()
        _, _, rank, _ = ops.score_dir_fwd(tn.detach(), tc.detach(), 1.0 / T, 1.0 / T, 0, False)
        d = np.diagonal(Sg)[:, None]
        exp_rank = (Sg > d).sum(1) + ((Sg == d) & (np.arange(B)[None, :] < np.arange(B)[:, None])).sum(1)
        assert np.array_equal(rank.cpu().numpy(), exp_rank.astype(np.int32))
        assert out8[1].item() == pytest.approx((exp_rank == 0).mean(), abs=1e-7)
        assert (Sg.argmax(1) == np.arange(B)).mean() == pytest.approx((exp_rank == 0).mean(), abs=1e-12)
        vals, idx = ops.topk_rows(torch.from_numpy(Sg).to(DEV), 7)
        ev, ei = O.topk_rows(Sg, 7)
        assert np.array_equal(idx.cpu().numpy(), ei) and np.array_equal(vals.cpu().numpy(), ev)


def test_score_bf16_path_vs_oracle(tt):
    """bf16-operand MFMA path: exact (to f32 accumulation) against the oracle fed the SAME bf16-rounded
    operands; gradient operands are rounded once more to bf16 (softmax weights), so grads are compared
    norm-wise at 1e-2 (bf16 has 8 significant bits)."""
    from jodalrob_twotower_amd import ops
    from jodalrob_twotower_amd.two_tower_train_task import _ScoreCEFn
    rng = np.random.default_rng(11)
    for B, D, T in [(300, 64, 1.0), (129, 16, 0.5), (64, 6, 0.25), (1000, 128, 1.0), (257, 200, 2.0), (2048, 64, 1.0), (70, 32, 1.0)]:
        n = rng.standard_normal((B, D)).astype(np.float32)
        c = rng.standard_normal((B, D)).astype(np.float32)
        n /= np.linalg.norm(n, axis=1, keepdims=True)
        c /= np.linalg.norm(c, axis=1, keepdims=True)
        c[5] = c[3]
        n[7] = n[2]
        # the notice image is packed times inv_t * log2(e) (the kernels' unit form): bf16(scale * n) / scale is the operand
        sn = ops.score_unit_scale(1.0 / T)
        nb = (torch.from_numpy(n) * np.float32(sn)).bfloat16().float().numpy().astype(np.float64) / float(np.float32(sn))
        cb = torch.from_numpy(c).bfloat16().float().numpy().astype(np.float64)
        loss, met, S, lse = O.score_ce_fwd(nb, cb, T)
        dN, dC = O.score_ce_bwd(nb, cb, S, lse, T)
        tn, tc = torch.from_numpy(n).to(DEV).requires_grad_(), torch.from_numpy(c).to(DEV).requires_grad_()
        l, out8, rank = _ScoreCEFn.apply(tn, tc, 1.0 / T, "bf16")
        l.backward()
        np.testing.assert_allclose(l.item(), loss, rtol=2e-5, err_msg=f"B={B} D={D}")
        np.testing.assert_allclose(out8[2].item(), met["positive_similarity_mean"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(out8[3].item(), met["negative_similarity_mean"], rtol=2e-3, atol=5e-6)
        for got, ref in ((tn.grad, dN), (tc.grad, dC)):
            g = got.cpu().numpy().astype(np.float64)
            assert np.linalg.norm(g - ref) <= 1e-2 * np.linalg.norm(ref), (B, D, np.linalg.norm(g - ref) / np.linalg.norm(ref))
        d = np.diagonal(S)[:, None]
        exp_rank = (S > d).sum(1) + ((S == d) & (np.arange(B)[None, :] < np.arange(B)[:, None])).sum(1)
        got_rank = rank.cpu().numpy()
        assert (got_rank == exp_rank).mean() >= 0.995, (B, D)
        assert got_rank[5] == exp_rank[5] and got_rank[3] == exp_rank[3]       # exact ties (duplicated company)
        # top-1-only mode (what the training step uses) == (full rank == 0), bit for bit
        Np, Cp = ops.score_pack_bf16(tn.detach()), ops.score_pack_bf16(tc.detach())
        full = ops.score_fwd_bf16(Np, Cp, B, D, 1.0 / T, 1.0 / T, True, True)
        top1 = ops.score_fwd_bf16(Np, Cp, B, D, 1.0 / T, 1.0 / T, True, False)
        for k in (3, 4):
            assert torch.equal(top1[k], (full[k] != 0).to(torch.int32)), (B, D, k)
        assert torch.equal(top1[0], full[0]) and torch.equal(top1[1], full[1])
        # against the exact-f32 path on the same inputs: loss within bf16 operand rounding
        l32, _, _ = _ScoreCEFn.apply(tn.detach(), tc.detach(), 1.0 / T, "fp32")
        np.testing.assert_allclose(l.item(), l32.item(), rtol=3e-3)


def test_task_bf16_score_close_to_fp32(tt, manifest):
    cfg = dict(manifest["cases"]["wide_b40"])
    cfg["B"] = 512
    shapes = None
    outs = {}
    for sd in ("fp32", "bf16"):
        task = make_task(tt, cfg, score_dtype=sd)
        shapes = {k: tuple(v.shape) for k, v in task.state_dict().items()}
        load_state(task, init_state_numpy(shapes, 99))
        b = synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 98, oob=False)
        task.train()
        res = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        res["loss"].backward()
        outs[sd] = (res["loss"].item(), {n: p.grad.cpu().numpy() for n, p in task.named_parameters()})
    np.testing.assert_allclose(outs["bf16"][0], outs["fp32"][0], rtol=3e-3)
    for k, g32 in outs["fp32"][1].items():
        gb = outs["bf16"][1][k]
        assert np.linalg.norm(gb - g32) <= 3e-2 * np.linalg.norm(g32) + 1e-9, k


def test_adam_trajectory_golden(tt, manifest):
    cfg = {**manifest["cases"]["tiny_train"], **manifest["cases"]["adam_trajectory"]}
    z = np.load(GOLD / "adam_trajectory.npz")
    task = make_task(tt, cfg)
    load_state(task, {k[6:]: z[k] for k in z.files if k.startswith("state.")})
    opt = torch.optim.Adam(task.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
    warm = cfg["warmup_steps"]
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: s / warm if s < warm else 1.0, last_epoch=-1)
    task.train()
    for s in range(cfg["n_steps"]):
        b = {k[len(f"step{s}.in."):]: z[k] for k in z.files if k.startswith(f"step{s}.in.")}
        opt.zero_grad()
        res = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        res["loss"].backward()
        opt.step()
        sched.step()
        np.testing.assert_allclose(res["loss"].item(), z[f"step{s}.loss"], rtol=5e-5)
    for k, v in task.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), z["final." + k], rtol=3e-4, atol=3e-6, err_msg=k)


def test_predict_batch_golden(tt, manifest):
    cfg = manifest["cases"]["tiny_eval"]
    g = load_case("tiny_eval")
    task = make_task(tt, cfg)
    load_state(task, split_prefix(g, "state."))
    pr = task.predict_batch(to_batch(tt, split_prefix(g, "in."), cfg["keys_n"], cfg["keys_c"]), top_k=5)
    assert np.array_equal(pr["top_indices"].cpu().numpy(), g["predict.top_indices"])
    np.testing.assert_allclose(pr["top_similarities"].cpu().numpy(), g["predict.top_similarities"], rtol=2e-5, atol=2e-6)


def test_medium_batch_vs_oracle(tt, manifest):
    """B=1024 on the synthetic schema with hot ids: full step vs the numpy oracle (f64 truth)."""
    cfg = dict(manifest["cases"]["wide_b40"])
    cfg["B"] = 1024
    task = make_task(tt, cfg)
    shapes = {k: tuple(v.shape) for k, v in task.state_dict().items()}
    state = init_state_numpy(shapes, 4242)
    load_state(task, state)
    b = synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 4343, oob=True)
    task.train()
    res = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
    res["loss"].backward()
    ref = O.task_step(state, b, cfg["keys_n"], cfg["keys_c"], cfg["vocab_n"], cfg["vocab_c"], cfg["T"], True, dtype=np.float64)
    np.testing.assert_allclose(res["loss"].item(), ref["loss"], rtol=1e-5)
    assert res["accuracy"].item() == pytest.approx(float(ref["accuracy"]), abs=2.0 / cfg["B"])
    for n, p in task.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref["grads"][n], rtol=2e-3, atol=2e-7, err_msg=n)


def test_distributed_world1_matches_single_gpu(tt, manifest):
    """The sharded-table / all-to-all path (RCCL backend, world_size 1 on the one GPU of this box) must
    reproduce the single-GPU step exactly: same loss, same dense grads, shard grad == fused table grad."""
    import os
    import torch.distributed as dist
    from jodalrob_twotower_amd.distributed import create_distributed_train_task
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        cfg = dict(manifest["cases"]["wide_b40"])
        cfg["B"] = 256
        single = make_task(tt, cfg, embedding_grad="dense")
        shapes = {k: tuple(v.shape) for k, v in single.state_dict().items()}
        state = init_state_numpy(shapes, 777)
        load_state(single, state)
        dtask = create_distributed_train_task(
            cfg["keys_n"], cfg["keys_c"], metadata_path=str(GOLD / "synthetic_metadata.csv"), categorical_embedding_dim=cfg["E"],
            notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"], tower_hidden_dims=list(cfg["hidden"]),
            final_embedding_dim=cfg["D"], dropout_rate=0.0, temperature=cfg["T"], device=DEV, embedding_grad="dense")
        dtask.load_full_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        full = dtask.full_state_dict()
        assert set(full) == set(state)
        for k, v in state.items():
            assert np.array_equal(full[k].cpu().numpy(), v), k                       # sharded <-> reference layout
        b = synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 778, oob=True)
        single.train(); dtask.train()
        r1 = single(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        r2 = dtask(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        assert r1["loss"].item() == r2["loss"].item()
        r1["loss"].backward(); r2["loss"].backward()
        g1 = {n: p.grad for n, p in single.named_parameters()}
        for n, p in dtask.named_parameters():
            if n != "embedding_shard":
                assert torch.equal(p.grad, g1[n]), n
        fused = torch.cat([g1[name] for name, _, _ in dtask.key_directory()])
        assert torch.equal(dtask.embedding_shard.grad[:fused.shape[0]], fused)
        # sparse mode + FusedAdam on the shard: one step runs and changes only looked-up rows
        from jodalrob_twotower_amd.optim import FusedAdam
        dsp = create_distributed_train_task(
            cfg["keys_n"], cfg["keys_c"], metadata_path=str(GOLD / "synthetic_metadata.csv"), categorical_embedding_dim=cfg["E"],
            notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"], tower_hidden_dims=list(cfg["hidden"]),
            final_embedding_dim=cfg["D"], dropout_rate=0.0, temperature=cfg["T"], device=DEV, embedding_grad="sparse")
        dsp.load_full_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()})
        opt = FusedAdam.for_task(dsp, lr=1e-2)
        before = dsp.embedding_shard.detach().clone()
        dsp.train()
        dsp(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"])).backward()
        opt.step()
        changed = (dsp.embedding_shard.detach() != before).any(dim=1).cpu().numpy()
        touched = np.zeros(len(changed), bool)
        touched[np.flatnonzero(np.abs(fused.cpu().numpy()).sum(1) > 0)] = True
        assert np.array_equal(changed[:len(touched)], touched)
    finally:
        dist.destroy_process_group()


def test_padded_exchange_graph_world1_subprocess(tt):
    """Fixed-capacity exchange == exact-size exchange, eagerly and as ONE captured graph with the RCCL collectives inside
    (world 1); a sharded run resumed from its checkpoint (model + FusedAdam state) continues bit for bit.  Runs in its own
    process so that it owns its process group from init to the ORDINARY teardown: GraphedTrainStep.close() -> objects released
    -> destroy_process_group() -> normal interpreter exit; the return code must be 0 (no os._exit escape)."""
    import subprocess, sys
    from pathlib import Path
    worker = Path(__file__).resolve().parent / "_dist_world1_worker.py"
    r = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=300)
    if "DIST_WORLD1_OK" not in r.stdout or r.returncode != 0:
        log = Path(__file__).resolve().parents[1] / "gpurun_out"
        log.mkdir(exist_ok=True)
        (log / "dist_world1_worker.log").write_text(r.stdout + "\n==== stderr ====\n" + r.stderr)
    assert "DIST_WORLD1_OK" in r.stdout, [ln for ln in r.stderr.splitlines() if "rror" in ln or "what()" in ln][-8:]
    assert r.returncode == 0, (r.returncode, r.stderr.splitlines()[-8:])


def test_fused_adam_matches_oracle(tt, manifest):
    """FusedAdam (dense tower weights + dense-mode tables) == torch.optim.Adam semantics (oracle adam_step);
    sparse mode == the same update restricted to looked-up rows."""
    from jodalrob_twotower_amd.optim import FusedAdam
    cfg = dict(manifest["cases"]["wide_b40"])
    for mode in ("dense", "sparse"):
        task = make_task(tt, cfg, embedding_grad=mode)
        shapes = {k: tuple(v.shape) for k, v in task.state_dict().items()}
        state = init_state_numpy(shapes, 31)
        load_state(task, state)
        opt = FusedAdam.for_task(task, lr=1e-2, weight_decay=1e-4)
        pk = [k for k in state if "running" not in k and "num_batches" not in k]
        m = {k: np.zeros_like(state[k]) for k in pk}
        v = {k: np.zeros_like(state[k]) for k in pk}
        st = {k: np.array(val, copy=True) for k, val in state.items()}
        task.train()
        for s in range(3):
            b = synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 50 + s, oob=False)
            opt.zero_grad()
            task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"])).backward()
            opt.step()
            ref = O.task_step(st, b, cfg["keys_n"], cfg["keys_c"], cfg["vocab_n"], cfg["vocab_c"], cfg["T"], True)
            for k in pk:
                g = ref["grads"][k]
                if mode == "sparse" and "embeddings" in k:
                    rows = np.flatnonzero(np.abs(g).sum(1) > 0)
                    for r in rows:
                        O.adam_step(st[k][r], g[r], m[k][r], v[k][r], s + 1, 1e-2, wd=1e-4)
                else:
                    O.adam_step(st[k], g, m[k], v[k], s + 1, 1e-2, wd=1e-4)
            st.update(ref["bn_updates"])
        for k, val in task.state_dict().items():
            np.testing.assert_allclose(val.cpu().numpy(), st[k], rtol=2e-4, atol=2e-6, err_msg=f"{mode}:{k}")


@pytest.mark.parametrize("mlp_dtype", ["fp32", "bf16"])
def test_graphed_step_equals_eager(tt, manifest, mlp_dtype):
    """HIP-graph replay of the whole step == the same steps launched eagerly (bitwise: same kernels,
    same order), with the learning rate changing between steps (LambdaLR warm-up) and new batches."""
    from jodalrob_twotower_amd.graph import GraphedTrainStep
    from jodalrob_twotower_amd.optim import FusedAdam
    cfg = dict(manifest["cases"]["wide_b40"])
    cfg["B"] = 256
    batches = [synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 900 + i, oob=False) for i in range(6)]
    finals = {}
    for mode in ("eager", "graph"):
        task = make_task(tt, cfg, embedding_grad="sparse", score_dtype="bf16", mlp_dtype=mlp_dtype)
        shapes = {k: tuple(v.shape) for k, v in task.state_dict().items()}
        load_state(task, init_state_numpy(shapes, 55))
        task.train()
        task._pair_check_done = True                # (the first call's alignment check takes the two-direction forward kernel, whose
        #                                             reciprocals differ from the steady state's in the last bit; the capture's warm-up has it behind it)
        opt = FusedAdam.for_task(task, lr=1e-2, weight_decay=1e-5)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: (s + 1) / 4 if s < 3 else 1.0)
        tb = [to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]) for b in batches]
        losses = []
        if mode == "eager":                         # (the graph wrapper's three eager warm-up steps leave no trace: preserve_state)
            for b in tb:
                opt.zero_grad()
                r = task(b, return_metrics=True)
                r["loss"].backward()
                opt.step(); sched.step()
                losses.append(r["loss"].item())
        else:
            gs = GraphedTrainStep(task, opt, tb[0], warmup=3)
            for b in tb:
                r = gs.step(b)
                sched.step()
                losses.append(r["loss"].item())
            assert opt.current_step() == len(tb)
        finals[mode] = (losses, {k: v.detach().cpu().numpy().copy() for k, v in task.state_dict().items()})
    assert finals["eager"][0] == finals["graph"][0]
    for k, v in finals["eager"][1].items():
        assert np.array_equal(v, finals["graph"][1][k]), k


@pytest.mark.parametrize("U", [2, 3])
def test_unrolled_step_equals_single_steps(tt, manifest, U):
    """unrolled.UnrolledTrainStep: U whole steps -- their hand-over launches included, as graph nodes whose arguments the host
    replaces before every launch (tt_handover_retarget) -- per graph launch == the same steps as U single-step replays, bit for
    bit: per-step losses, final state, optimiser step count.  Covers a learning rate that changes between the steps of ONE launch
    (after_each = the scheduler's step), dropout (a fresh seed per step through the ring; the test re-seeds torch's CPU generator,
    which the ring draws from, before every step), out-of-range ids, a batch that arrives in host memory in lane 1; 7 steps = two
    (U = 3) or three (U = 2) launches + a remainder through step() (the single-step sibling, created while training is under way:
    its warm-up must leave no trace), and metric sums shared by both captured steps."""
    from jodalrob_twotower_amd.graph import GraphedTrainStep
    from jodalrob_twotower_amd.unrolled import UnrolledTrainStep
    from jodalrob_twotower_amd.optim import FusedAdam
    cfg = dict(manifest["cases"]["wide_b40"])
    cfg["B"] = 256
    n_steps = 7
    batches = [synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 1900 + i, oob=True) for i in range(n_steps)]
    finals = {}
    for mode in ("single", "unrolled"):
        task = make_task(tt, cfg, embedding_grad="sparse", score_dtype="bf16", mlp_dtype="bf16", dropout_rate=0.1)
        load_state(task, init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 57))
        task.train()
        task._pair_check_done = True
        for tw in (task.two_tower_model.notice_tower, task.two_tower_model.company_tower):
            tw._seed_override = 4242               # a capture bakes a host seed (drawn at capture time) to which the ring's per-step word is added: the same one for both objects
        opt = FusedAdam.for_task(task, lr=1e-2, weight_decay=1e-5)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: (s + 1) / 4 if s < 3 else 1.0)
        tb = [to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]) for b in batches]
        tb[1] = {s_: {"dense": tb[1][s_]["dense"].cpu(), "kjt": type(tb[1][s_]["kjt"])(tb[1][s_]["kjt"].keys(), tb[1][s_]["kjt"].values().cpu())}
                 for s_ in ("notice", "company")}                                     # a host batch (lane 1)
        losses = []
        if mode == "single":
            gs = GraphedTrainStep(task, opt, tb[0], warmup=2, accumulate_metrics=True)
            for i, b in enumerate(tb):
                torch.manual_seed(1000 + i)                                         # the ring draws the step's dropout seed from torch's CPU generator
                r = gs.step(b)
                sched.step()
                losses.append(r["loss"].item())
        else:
            gs = UnrolledTrainStep(task, opt, tb[0], unroll=U, warmup=2, accumulate_metrics=True)
            assert gs.unroll == U and len(gs._nodes) == U
            i, nxt = 0, [0]

            def after_each():
                sched.step()
                nxt[0] += 1
                torch.manual_seed(1000 + nxt[0])
            while i + U <= n_steps:
                torch.manual_seed(1000 + i)
                nxt[0] = i
                res = gs.step_many(tb[i:i + U], after_each=after_each)
                losses += [r["loss"].item() for r in res]
                i += U
            while i < n_steps:                                                      # remainder: the single-step sibling
                gs._single()
                torch.manual_seed(1000 + i)
                r = gs.step(tb[i])
                sched.step()
                losses.append(r["loss"].item())
                i += 1
            assert gs.single is not None or n_steps % U == 0
        torch.cuda.synchronize()
        assert opt.current_step() == n_steps
        finals[mode] = (losses, {k: v.detach().cpu().numpy().copy() for k, v in task.state_dict().items()}, gs.metric_sums.cpu().numpy().copy(),
                        gs.library_launches)
        gs.close()
    assert finals["single"][3] == finals["unrolled"][3]                              # launches per STEP: the hand-over counted once either way
    assert len(set(finals["single"][0])) == n_steps
    assert finals["single"][0] == finals["unrolled"][0], (finals["single"][0], finals["unrolled"][0])
    for k, v in finals["single"][1].items():
        assert np.array_equal(v, finals["unrolled"][1][k]), k
    assert np.array_equal(finals["single"][2], finals["unrolled"][2])                # one epoch total, whoever ran the step


def test_graph_ingest_key_major_plan(tt, manifest, monkeypatch):
    """GraphedTrainStep hands a batch over with ops.batch_ingest (copies + the fused rows of the batch's ids in key-major order,
    which the duplicate-row plan then sorts instead of gathering every key's rows out of the sample-major array) == the same
    replayed steps with plain copies and the strided gather (TT_GRAPH_INGEST=0), bit for bit: ragged last 64-sample tile,
    out-of-range ids (the ingest clamps like the lookup), a batch that arrives in host memory, and static id buffers that
    somebody overwrote with a torch op between replays (the stale key-major rows must not be used).  Round 4: the same steps
    with the captured lookup reading the hand-over launch's slot-order rows (tt_embed_lookup_rows_fwd, the default) or decoding the
    ids itself, and with hand-over and lookup as ONE launch (tt_batch_ingest_lookup) -- all bit for bit."""
    from jodalrob_twotower_amd.graph import GraphedTrainStep
    from jodalrob_twotower_amd.optim import FusedAdam
    from jodalrob_twotower_amd import ops as _ops
    cfg = dict(manifest["cases"]["wide_b40"])
    cfg["B"] = 300
    batches = [synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 930 + i, oob=True) for i in range(5)]
    finals = {}
    for ingest, rows_sm, fused in (("0", False, False), ("1", False, False), ("1", True, False), ("1", False, True)):
        monkeypatch.setattr(_cfg.settings, "graph_ingest", ingest == "1")
        monkeypatch.setattr(_cfg.settings, "graph_ingest_rows", rows_sm)
        monkeypatch.setattr(_cfg.settings, "graph_ingest_lookup", fused)
        task = make_task(tt, cfg, embedding_grad="sparse", score_dtype="bf16", mlp_dtype="bf16", dropout_rate=0.0)
        load_state(task, init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 56))
        task.train()
        opt = FusedAdam.for_task(task, lr=1e-2)
        tb = [to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]) for b in batches]
        seen, plain = [], _ops.dedup_plan_keyed
        monkeypatch.setattr(_ops, "dedup_plan_keyed", lambda rows, ks, B_, key_major=False, E=0: (seen.append(key_major), plain(rows, ks, B_, key_major, E))[1])
        gs = GraphedTrainStep(task, opt, tb[0], warmup=2)
        monkeypatch.setattr(_ops, "dedup_plan_keyed", plain)
        store = task.two_tower_model.embedding_store
        store = store() if callable(store) else store
        assert (gs._ingest is not None) == (ingest == "1") and (store.ingest is not None) == (ingest == "1")
        assert (gs._rows_sm is not None) == rows_sm and (gs._x_static is not None) == fused
        assert len(seen) == 3 and all(k == (ingest == "1") for k in seen)      # warm-up and capture sort the key-major rows
        losses = [gs.step(b)["loss"].item() for b in tb[:3]]
        host = {s: {"dense": tb[3][s]["dense"].cpu(), "kjt": type(tb[3][s]["kjt"])(tb[3][s]["kjt"].keys(), tb[3][s]["kjt"].values().cpu())}
                for s in ("notice", "company")}
        losses.append(gs.step(host)["loss"].item())                       # host batch: ordinary copies into the static buffers
        for s in ("notice", "company"):                                   # somebody writes the static buffers directly ...
            gs.static[s]["dense"].copy_(tb[4][s]["dense"])
            gs.static[s]["kjt"].values().copy_(tb[4][s]["kjt"].values())
        losses.append(gs.step(None)["loss"].item())                       # ... and replays on "whatever the static buffers hold"
        with torch.no_grad():                                             # a write no version counter sees
            gs.static["notice"]["kjt"].values().data.copy_(tb[2]["notice"]["kjt"].values())
        losses.append(gs.step(None)["loss"].item())
        finals[(ingest, rows_sm, fused)] = (losses, {k: v.detach().cpu().numpy().copy() for k, v in task.state_dict().items()}, gs.library_launches)
        gs.close()
    base = finals[("0", False, False)]
    assert len(set(base[0])) > 1
    for key, (losses, state, _) in finals.items():
        assert losses == base[0], key
        for k, v in base[1].items():
            assert np.array_equal(v, state[k]), (key, k)
    assert finals[("1", False, True)][2] == finals[("1", True, False)][2] - 1        # the fused hand-over has no lookup launch in the graph


@pytest.mark.parametrize("B,grad", [(2048, "sparse"), (777, "sparse"), (8192, "dense")])
def test_planned_long_rows_equal_unplanned(tt, manifest, schema_real, monkeypatch, B, grad):
    """The long-row list of the gradient reduction built by the plan's compaction (tt_dedup_plan_keyed_long: row and chunk passes in
    ONE launch) == the list the reduction registers itself (TT_GRAD_PLANNED=0), bit for bit: same chunks, same order.  Real 32 + 6
    key schema (17 two-row keys: rows of thousands of slots) with out-of-range ids (clamping piles slots onto the border rows)."""
    cfg = dict(manifest["cases"]["real_schema"])
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    cfg.update(keys_n=kn, keys_c=kc)
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    b = synth_batch_numpy(B, vn, vc, cfg["din_n"], cfg["din_c"], 610, oob=True)
    outs, state = {}, None
    for planned in ("0", "1"):
        monkeypatch.setattr(_cfg.settings, "grad_planned", planned == "1")
        task = make_task(tt, cfg, meta=GOLD / "real_vocab_metadata.csv", embedding_grad=grad, mlp_dtype="bf16", score_dtype="bf16")
        if state is None:
            state = init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 611)
        load_state(task, state)
        task.train()
        res = task(to_batch(tt, b, kn, kc), return_metrics=True)
        res["loss"].backward()
        store = task.two_tower_model.embedding_store
        store = store() if callable(store) else store
        if grad == "sparse":
            plan, rows = store.sparse_grad
            U = int(plan.n_unique.item())
            assert (plan.grad_ws is not None) == (planned == "1")
            outs[planned] = (plan.unique_rows[:U].cpu().numpy(), rows[:U].cpu().numpy())
        else:
            outs[planned] = (np.zeros(1), store.grad.cpu().numpy())
    assert np.array_equal(outs["0"][0], outs["1"][0]) and np.array_equal(outs["0"][1], outs["1"][1])
    assert np.abs(outs["1"][1]).sum() > 0


@pytest.mark.parametrize("B,consumer", [(2048, "fused"), (777, "fused"), (2048, "finish"), (2048, "sparse_adam")])
def test_deferred_long_finish_equals_immediate(tt, manifest, schema_real, B, consumer):
    """The long rows of the sparse gradient finished inside the optimiser's launch (TT_GRAD_DEFER_FINISH +
    tt_adam_fused_step_finish: what GraphedTrainStep replays), by tt_embed_grad_finish, or ahead of the row-sparse Adam ==
    the reduction's own finish launch, bit for bit: gradient rows, table, both Adam moments and every dense weight after two
    steps.  Real 32 + 6 key schema (17 two-row keys: rows of thousands of slots), out-of-range ids."""
    from jodalrob_twotower_amd.optim import FusedAdam
    from jodalrob_twotower_amd import ops
    cfg = dict(manifest["cases"]["real_schema"])
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    cfg.update(keys_n=kn, keys_c=kc)
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    batches = [synth_batch_numpy(B, vn, vc, cfg["din_n"], cfg["din_c"], 640 + i, oob=True) for i in range(2)]
    outs, state = {}, None
    for defer in (False, True):
        task = make_task(tt, cfg, meta=GOLD / "real_vocab_metadata.csv", embedding_grad="sparse", mlp_dtype="bf16", score_dtype="bf16")
        if state is None:
            state = init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 641)
        load_state(task, state)
        task.train()
        opt = FusedAdam.for_task(task, lr=1e-2, weight_decay=1e-5)
        store = task.two_tower_model.embedding_store
        store = store() if callable(store) else store
        grads = []
        for b in batches:
            opt.zero_grad()
            store.defer_long_finish = defer
            task(to_batch(tt, b, kn, kc), return_metrics=True)["loss"].backward()
            store.defer_long_finish = False
            plan, rows = store.sparse_grad
            assert (plan.finish_deferred is not None) == defer and plan.grad_ws is not None
            if consumer == "finish":
                ops.embed_grad_finish(plan)
                assert plan.finish_deferred is None
                opt.step()
            elif consumer == "sparse_adam":                # the optimiser's unfused path: tower weights first, then the rows
                opt._fusable_store = lambda *a, **k: None
                opt.step()
            else:
                opt.step()
            assert plan.finish_deferred is None
            U = int(plan.n_unique.item())
            grads.append((plan.unique_rows[:U].cpu().numpy(), rows[:U].cpu().numpy()))
        st = opt._state_of(store)
        outs[defer] = (grads, store.weight.cpu().numpy(), st["m"].cpu().numpy(), st["v"].cpu().numpy(),
                       {k: v.detach().cpu().numpy() for k, v in task.state_dict().items()})
    for (u0, g0), (u1, g1) in zip(outs[False][0], outs[True][0]):
        assert np.array_equal(u0, u1) and np.array_equal(g0, g1)
        assert np.abs(g1).sum() > 0
    for i in (1, 2, 3):
        assert np.array_equal(outs[False][i], outs[True][i]), i
    for k, v in outs[False][4].items():
        assert np.array_equal(v, outs[True][4][k]), k


@pytest.mark.parametrize("hidden,D,B", [([128, 64], 64, 2048), ([512, 256], 128, 1024), ([128, 64], 64, 1000)])
def test_riders_in_tail_launches_equal_own_launches(tt, schema_real, hidden, D, B):
    """TT_OPT_DEFER_RIDERS (what GraphedTrainStep replays): the keyed plan's compaction rides in the towers' tail_fwd launch and the
    symmetric score forward's loss reduction in tail_bwd's == both as launches of their own, bit for bit over three replayed steps:
    losses, metrics, every parameter.  Real 32 + 6 key schema; the reference-shaped towers take the fused tail (riders hosted), the
    [512, 256] -> 128 ones do not (the queue is launched by the embedding gradient / at the end of the step), B = 1000 has a
    ragged last tile."""
    from jodalrob_twotower_amd.graph import GraphedTrainStep
    from jodalrob_twotower_amd.optim import FusedAdam
    from jodalrob_twotower_amd import synthetic
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    dev = torch.device(DEV)
    batches = [synthetic.make_batch(B, vn, vc, kn, kc, 256, 128, dev, seed=310 + i) for i in range(4)]
    finals, state = {}, None
    for riders in (False, True):
        torch.manual_seed(9)
        task = tt.create_two_tower_train_task(kn, kc, metadata_path=str(GOLD / "real_vocab_metadata.csv"), categorical_embedding_dim=32,
                                              notice_dense_input_dim=256, company_dense_input_dim=128, tower_hidden_dims=hidden,
                                              final_embedding_dim=D, dropout_rate=0.1, device=DEV, embedding_grad="sparse",
                                              score_dtype="bf16", mlp_dtype="bf16")
        task._pair_check_done = True
        if state is None:
            state = {k: v.detach().clone() for k, v in task.state_dict().items()}
        task.load_state_dict(state)
        task.train()
        for tw in (task.two_tower_model.notice_tower, task.two_tower_model.company_tower):
            tw._seed_override = 77
        opt = FusedAdam.for_task(task, lr=1e-2, weight_decay=1e-5)
        gs = GraphedTrainStep(task, opt, batches[0], warmup=1, defer_riders=riders)
        outs = []
        for b in batches[1:]:
            r = gs.step(b)
            outs.append([r[k].item() for k in ("loss", "accuracy", "positive_similarity_mean", "negative_similarity_mean", "similarity_gap")])
        torch.cuda.synchronize()
        finals[riders] = (outs, {k: v.detach().cpu().clone() for k, v in task.state_dict().items()})
        gs.close()
    assert finals[False][0] == finals[True][0] and len({o[0] for o in finals[True][0]}) == 3
    for k, v in finals[False][1].items():
        assert torch.equal(v, finals[True][1][k]), k


@pytest.mark.parametrize("B,planned,hidden", [(2048, "1", None), (8192, "1", None), (2048, "0", None), (2048, "1", [512, 256]), (1000, "1", None)])
def test_deferred_slab_reduce_equals_immediate(tt, manifest, schema_real, monkeypatch, B, planned, hidden):
    """The slab reduction of the towers' weight gradients run by the first workgroups of the embedding gradient's launch
    (TT_OPT_DEFER_SLAB_REDUCE: what GraphedTrainStep replays; planned workspace), or flushed ahead of it (unplanned: the
    reduction's scratch is the buffer the slabs live in) == its own launch at the end of tt_towers_mlp_bwd, bit for bit: every
    dense gradient and the sparse gradient rows.  Real 32 + 6 key schema, bf16 towers (the one-launch first-block backward)."""
    from jodalrob_twotower_amd import _lib as L
    monkeypatch.setattr(_cfg.settings, "grad_planned", planned == "1")
    cfg = dict(manifest["cases"]["real_schema"])
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    cfg.update(keys_n=kn, keys_c=kc)
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    if hidden is not None:                                   # scripts/train.py's widths: the general backward (separate GEMMs)
        cfg.update(hidden=hidden, D=128)
    b = synth_batch_numpy(B, vn, vc, cfg["din_n"], cfg["din_c"], 650, oob=True)
    outs, state = {}, None
    for defer in (False, True):
        task = make_task(tt, cfg, meta=GOLD / "real_vocab_metadata.csv", embedding_grad="sparse", mlp_dtype="bf16", score_dtype="bf16")
        if state is None:
            state = init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 651)
        load_state(task, state)
        task.train()
        loss = task(to_batch(tt, b, kn, kc), return_metrics=True)["loss"]
        L.set_defer_slab_reduce(loss.device, defer)
        try:
            loss.backward()
            pending = bool(L.load().tt_deferred_pending(L.ctx(loss.device)))
        finally:
            L.set_defer_slab_reduce(loss.device, False)
        assert not pending                                   # consumed by the gradient launch (planned) or flushed ahead of it
        store = task.two_tower_model.embedding_store
        store = store() if callable(store) else store
        plan, rows = store.sparse_grad
        U = int(plan.n_unique.item())
        outs[defer] = ({n: p.grad.cpu().numpy() for n, p in task.named_parameters() if p.grad is not None},
                       plan.unique_rows[:U].cpu().numpy(), rows[:U].cpu().numpy())
    assert len(outs[True][0]) >= 10
    for k, g in outs[False][0].items():
        assert np.array_equal(g, outs[True][0][k]), k
        assert np.isfinite(g).all()
    assert np.array_equal(outs[False][1], outs[True][1]) and np.array_equal(outs[False][2], outs[True][2])


def test_bf16_mlp_close_to_fp32(tt, manifest, schema_real):
    """mlp_dtype='bf16' (GEMM operands rounded to bf16, f32 accumulate, f32 tensors in memory) against the exact-f32
    MFMA path on the real 32+6-key schema: loss within 5e-3, gradients within 6e-2 norm-wise.  This is a SANITY bound on
    what operand rounding costs (two different arithmetics, HIP vs HIP); the parity claim of the bf16 mode rests on
    test_bf16_step_vs_rounded_oracle, which compares it with the f64 oracle fed the same rounded operands at 1e-3 or better."""
    cfg = dict(manifest["cases"]["real_schema"])
    cfg.update(keys_n=schema_real["notice"]["categorical"], keys_c=schema_real["company"]["categorical"], B=512)
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    shapes = {k: tuple(v) for k, v in manifest["state_dict_keys_real"].items()}
    state = init_state_numpy(shapes, 123)
    b = synth_batch_numpy(cfg["B"], vn, vc, cfg["din_n"], cfg["din_c"], 124, oob=False)
    outs = {}
    for md in ("fp32", "bf16"):
        task = make_task(tt, cfg, meta=GOLD / "real_vocab_metadata.csv", mlp_dtype=md)
        load_state(task, state)
        task.train()
        res = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        res["loss"].backward()
        outs[md] = (res["loss"].item(), {n: p.grad.cpu().numpy() for n, p in task.named_parameters()},
                    {k: v.cpu().numpy() for k, v in task.state_dict().items() if "running" in k or "num_batches" in k})
    np.testing.assert_allclose(outs["bf16"][0], outs["fp32"][0], rtol=5e-3)
    for k, g32 in outs["fp32"][1].items():
        gb = outs["bf16"][1][k]
        # three chained bf16-operand GEMMs (8-bit mantissa): 6e-2 norm-wise for matrices; the 1-D gradients (biases,
        # BN scale/shift) are column sums over the batch whose terms largely cancel, so their RELATIVE error is larger
        tol = 6e-2 if g32.ndim > 1 else 1.5e-1
        assert np.linalg.norm(gb - g32) <= tol * np.linalg.norm(g32) + 1e-9, (k, np.linalg.norm(gb - g32) / np.linalg.norm(g32))
    for k, v in outs["fp32"][2].items():
        np.testing.assert_allclose(outs["bf16"][2][k], v, rtol=2e-2, atol=2e-3, err_msg=k)
    assert all(int(v) == 1 for k, v in outs["bf16"][2].items() if "num_batches" in k)       # counter bumped in-kernel


@pytest.mark.parametrize("B,Ks,vocabs,parts", [(8192, [32, 6], None, 0), (1000, [5, 2], None, 0), (1, [3], None, 0), (4097, [4, 3, 2], None, 0),
                                               (8192, [6, 3], "hot", 0), (8192, [8], "mid", 0), (515, [7, 2], "mid", 0),
                                               (8192, [32, 6], None, 1), (8192, [32, 6], None, 2), (1000, [5, 2], None, 8), (515, [7, 2], "mid", 3),
                                               (8192, [6, 3], "hot", 8), (8192, [8], "mid", 5), (70, [4], None, 4), (33, [2, 2], "mid", 2)])
def test_dedup_plan_keyed_equals_general(tt, ctx_option, B, Ks, vocabs, parts):
    """Per-key LDS plan == the general radix-sort plan (and numpy's stable argsort), bit for bit.  Wide row ranges take the
    bucket + rank path, narrow ones and keys with hot rows ("hot": Zipf-like ids, a few rows holding most slots -- a bucket
    overflows) the stable LSD passes; "mid": vocabularies around the 9-bit switch and the per-bucket cap.  parts: workgroups
    per key (0 = the library's choice: 4 from B = 2048 up) -- shares of the key's row range, also odd counts, more shares than
    rows (a two-row key), the LSD fallback (share 0 sorts alone) and batches below the bucket path."""
    from jodalrob_twotower_amd import ops
    ctx_option(_L.TT_OPT_KEYED_PARTS, parts, 0)
    rng = np.random.default_rng(B + len(Ks))
    rows_sides, off = [], 0
    for K in Ks:
        v = rng.choice({"mid": [255, 256, 257, 511, 513, 600, 1023, 1025, 5000]}.get(vocabs, [2, 12, 300, 70000, 1_000_000]), size=K)
        offs = off + np.concatenate([[0], np.cumsum(v)[:-1]])
        if vocabs == "hot":
            ids = np.stack([np.minimum(rng.zipf(1.2, B) - 1, vk - 1) for vk in v], axis=1)
        else:
            ids = np.stack([rng.integers(0, vk, B) for vk in v], axis=1)
        rows_sides.append((ids + offs[None, :]).reshape(-1))
        off += int(v.sum())
    rows = np.concatenate(rows_sides).astype(np.int32)
    t = torch.from_numpy(rows).to(DEV)
    pk = ops.dedup_plan_keyed(t, Ks, B)
    pg = ops.dedup_plan(t, off)
    U = int(pk.n_unique.item())
    assert U == int(pg.n_unique.item())
    order = np.argsort(rows, kind="stable")
    assert np.array_equal(pk.sorted_src.cpu().numpy(), order.astype(np.int32))
    assert torch.equal(pk.sorted_src, pg.sorted_src)
    assert torch.equal(pk.unique_rows[:U], pg.unique_rows[:U]) and torch.equal(pk.seg_offsets[:U + 1], pg.seg_offsets[:U + 1])


@pytest.mark.parametrize("seed", list(range(12)))
def test_dedup_plan_keyed_random_shapes(tt, ctx_option, seed):
    """Randomised shapes for the per-key plan: batch, key counts, vocabularies (two-row keys to millions of rows), id
    distributions (uniform, Zipf-like, constant), share counts, key-major or slot-major rows, with or without the gradient
    reduction's long-row list -- always the order numpy's stable argsort gives, and a chunk list that tiles every long row."""
    from jodalrob_twotower_amd import ops
    rng = np.random.default_rng(1000 + seed)
    B = int(rng.choice([1, 63, 64, 65, 700, 2047, 2048, 3000, 8191, 8192]))
    Ks = [int(k) for k in rng.integers(1, 9, size=int(rng.integers(1, 4)))]
    parts = int(rng.choice([0, 1, 2, 3, 4, 6, 8]))
    key_major = bool(rng.integers(0, 2))
    E = int(rng.choice([0, 16, 32]))
    ctx_option(_L.TT_OPT_KEYED_PARTS, parts, 0)
    sides_sm, sides_km, off = [], [], 0
    for K in Ks:
        v = rng.choice([2, 3, 17, 300, 4096, 4097, 70000, 1_000_000], size=K)
        offs = off + np.concatenate([[0], np.cumsum(v)[:-1]])
        cols = []
        for vk in v:
            mode = rng.integers(0, 3)
            cols.append(rng.integers(0, vk, B) if mode == 0 else np.minimum(rng.zipf(1.3, B) - 1, vk - 1) if mode == 1
                        else np.full(B, int(rng.integers(0, vk))))
        ids = np.stack(cols, axis=1) + offs[None, :]
        sides_sm.append(ids.reshape(-1))
        sides_km.append(ids.T.reshape(-1))
        off += int(v.sum())
    rows = np.concatenate(sides_sm).astype(np.int32)
    t_in = torch.from_numpy(np.concatenate(sides_km).astype(np.int32) if key_major else rows).to(DEV)
    pk = ops.dedup_plan_keyed(t_in, Ks, B, key_major, E=E)
    order = np.argsort(rows, kind="stable")
    assert np.array_equal(pk.sorted_src.cpu().numpy(), order.astype(np.int32)), (B, Ks, parts, key_major, E)
    srt = rows[order]
    heads = np.flatnonzero(np.concatenate([[True], srt[1:] != srt[:-1]]))
    U = int(pk.n_unique.item())
    assert U == len(heads)
    assert np.array_equal(pk.unique_rows[:U].cpu().numpy(), srt[heads])
    seg = pk.seg_offsets[:U + 1].cpu().numpy()
    assert np.array_equal(seg, np.concatenate([heads, [len(rows)]]))
    if E and pk.grad_ws is not None:                      # the long-row list: every row longer than 64 slots, tiled by 64-slot chunks
        ws = pk.grad_ws[0].cpu().numpy()                  # layout: grad_layout() in csrc/tt_embed.hip (256-byte aligned regions)
        M = len(rows)
        al = lambda n: (n + 255) // 256 * 256
        max_long = M // 64 + 1
        max_chunks = M // 64 + max_long + 1
        o = 0
        counters = ws[o:o + 12].view(np.int32); o += al(12)
        long_row = ws[o:o + 4 * max_long].view(np.int32); o += al(4 * max_long)
        long_base = ws[o:o + 4 * max_long].view(np.int32); o += al(4 * max_long)
        chunk_lo = ws[o:o + 4 * max_chunks].view(np.int32); o += al(4 * max_chunks)
        chunk_hi = ws[o:o + 4 * max_chunks].view(np.int32)
        lens = np.diff(seg)
        want_long = set(np.flatnonzero(lens > 64).tolist())
        n_long, n_ch = int(counters[1]), int(counters[0])
        assert set(long_row[:n_long].tolist()) == want_long and n_long == len(want_long)
        assert n_ch == int(sum((lens[u] + 63) // 64 for u in want_long))
        for li in range(n_long):
            u, cb = int(long_row[li]), int(long_base[li])
            nch = (lens[u] + 63) // 64
            assert np.array_equal(chunk_lo[cb:cb + nch], seg[u] + 64 * np.arange(nch))
            assert np.array_equal(chunk_hi[cb:cb + nch], np.minimum(seg[u + 1], seg[u] + 64 * (np.arange(nch) + 1)))


def test_bf16_dense_features_from_the_loader_are_bit_identical(tt, manifest, schema_real):
    """mlp_dtype='bf16': a loader that keeps the dense features in bf16 (half the PCIe bytes of a host-fed step: bench.py's
    value_with_h2d_bf16_dense) changes nothing -- the projection GEMM rounds them to bf16 on the way into LDS anyway: loss and
    every gradient equal to the last bit."""
    cfg = dict(manifest["cases"]["real_schema"])
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    cfg.update(keys_n=kn, keys_c=kc)
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    b = synth_batch_numpy(320, vn, vc, cfg["din_n"], cfg["din_c"], 620, oob=False)
    outs, state = [], None
    for as_bf16 in (False, True):
        task = make_task(tt, cfg, meta=GOLD / "real_vocab_metadata.csv", mlp_dtype="bf16", score_dtype="bf16", dropout_rate=0.0)
        if state is None:
            state = init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 621)
        load_state(task, state)
        task.train()
        batch = to_batch(tt, b, kn, kc)
        if as_bf16:
            for s_ in ("notice", "company"):
                batch[s_]["dense"] = batch[s_]["dense"].to(torch.bfloat16)
        res = task(batch, return_metrics=True)
        res["loss"].backward()
        outs.append((res["loss"].item(), {n: p.grad.cpu().numpy() for n, p in task.named_parameters()}))
    assert outs[0][0] == outs[1][0]
    for k, g in outs[0][1].items():
        assert np.array_equal(outs[1][1][k], g), k


def test_bf16_tower_input_is_bit_identical(tt, manifest, schema_real, monkeypatch):
    """mlp_dtype='bf16' with the tower input x stored in bf16 (default) == the same mode with x in f32: the GEMMs round x
    to bf16 on the way into LDS either way, so loss and every gradient agree to the last bit; a bf16 d_x (opt-in) only
    rounds the per-slot row gradients: table gradients within 1e-2 norm-wise, everything else still bit-identical
    except the projection weights that read d_x."""
    cfg = dict(manifest["cases"]["real_schema"])
    cfg.update(keys_n=schema_real["notice"]["categorical"], keys_c=schema_real["company"]["categorical"], B=300)
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    shapes = {k: tuple(v) for k, v in manifest["state_dict_keys_real"].items()}
    state = init_state_numpy(shapes, 77)
    b = synth_batch_numpy(cfg["B"], vn, vc, cfg["din_n"], cfg["din_c"], 78, oob=False)
    outs = {}
    monkeypatch.setattr(_cfg.settings, "tower_unfused_front", True)      # storage type only: the one-launch front needs bf16 x and sums K in another order
    for io in ("none", "x", "both"):
        monkeypatch.setattr(_cfg.settings, "tower_io_dtype", io)
        task = make_task(tt, cfg, meta=GOLD / "real_vocab_metadata.csv", mlp_dtype="bf16")
        load_state(task, state)
        task.train()
        res = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        res["loss"].backward()
        outs[io] = (res["loss"].item(), {n: p.grad.cpu().numpy() for n, p in task.named_parameters()})
    assert outs["x"][0] == outs["none"][0]
    for k, g in outs["none"][1].items():
        assert np.array_equal(outs["x"][1][k], g), k
    assert outs["both"][0] == outs["none"][0]
    for k, g in outs["none"][1].items():
        gb = outs["both"][1][k]
        if "embeddings" in k or "dense_projection" in k:
            assert np.linalg.norm(gb - g) <= 1e-2 * np.linalg.norm(g) + 1e-12, k
        else:
            assert np.array_equal(gb, g), k


@pytest.mark.parametrize("case,B,drop", [("real_schema", 300, 0.0), ("real_schema", 8200, 0.1), ("wide_b40", 40, 0.1),
                                          ("deep_temp", 24, 0.0), ("single_hidden", 8, 0.0)])
def test_fused_tower_tail_equals_separate_kernels(tt, manifest, schema_real, monkeypatch, case, B, drop):
    """bf16 training pass: the fused tail (slab reduce + BN statistics | BN apply + output Linear + L2 normalise, and the
    mirror image backward) against the same pass run as the separate kernels (TT_TOWER_UNFUSED_TAIL=1).  The fused
    kernels keep the separate kernels' arithmetic and summation order, so the forward is expected bit-identical; the
    output-layer weight / bias gradients are summed over different row chunks (tolerance), the rest to rounding."""
    cfg = dict(manifest["cases"][case])
    if case == "real_schema":
        cfg.update(keys_n=schema_real["notice"]["categorical"], keys_c=schema_real["company"]["categorical"])
        vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
        shapes = {k: tuple(v) for k, v in manifest["state_dict_keys_real"].items()}
        meta = GOLD / "real_vocab_metadata.csv"
    else:
        vn, vc = cfg["vocab_n"], cfg["vocab_c"]
        z = np.load(GOLD / f"case_{case}.npz")
        shapes = {k[6:]: z[k].shape for k in z.files if k.startswith("state.")}
        meta = None
    state = init_state_numpy(shapes, 91)
    b = synth_batch_numpy(B, vn, vc, cfg["din_n"], cfg["din_c"], 92, oob=False)
    outs = {}
    monkeypatch.setattr(_cfg.settings, "tower_unfused_front", True)      # the one-launch front sums K in another order: compared in its own test
    for unfused in ("1", "0"):
        monkeypatch.setattr(_cfg.settings, "tower_unfused_tail", unfused == "1")
        torch.manual_seed(1234)
        task = make_task(tt, cfg, meta=meta, mlp_dtype="bf16", dropout_rate=drop)
        load_state(task, state)
        task.train()
        res = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        res["loss"].backward()
        sd = {k: v.detach().cpu().numpy() for k, v in task.state_dict().items() if "running" in k or "num_batches" in k}
        outs[unfused] = (res["loss"].item(), {n: p.grad.cpu().numpy() for n, p in task.named_parameters()}, sd)
    assert outs["0"][0] == outs["1"][0], (outs["0"][0], outs["1"][0])
    for k, v in outs["1"][2].items():
        assert np.array_equal(outs["0"][2][k], v), k
    for k, g in outs["1"][1].items():
        gf = outs["0"][1][k]
        assert np.isfinite(gf).all(), k
        assert np.linalg.norm(gf - g) <= 5e-6 * np.linalg.norm(g) + 1e-12, (k, np.linalg.norm(gf - g), np.linalg.norm(g))


@pytest.mark.parametrize("case,B", [("real_schema", 300), ("real_schema", 8192), ("wide_b40", 40), ("deep_temp", 24)])
def test_tower_emits_packed_score_operands(tt, manifest, schema_real, monkeypatch, case, B):
    """tt_tower_acts.emb_packed: the operand images the tower pass writes (fused tail kernel, or the pack kernel behind the
    same field on the other paths) are the images tt_score_pack2_bf16 makes from the unit rows -- the step with and without
    the separate pack launch (TT_TOWER_PACK=0) agrees bit for bit, ragged last tiles and padded columns included."""
    cfg = dict(manifest["cases"][case])
    if case == "real_schema":
        cfg.update(keys_n=schema_real["notice"]["categorical"], keys_c=schema_real["company"]["categorical"])
        vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
        shapes = {k: tuple(v) for k, v in manifest["state_dict_keys_real"].items()}
        meta = GOLD / "real_vocab_metadata.csv"
    else:
        vn, vc = cfg["vocab_n"], cfg["vocab_c"]
        z = np.load(GOLD / f"case_{case}.npz")
        shapes = {k[6:]: z[k].shape for k in z.files if k.startswith("state.")}
        meta = None
    state = init_state_numpy(shapes, 141)
    b = synth_batch_numpy(B, vn, vc, cfg["din_n"], cfg["din_c"], 142, oob=False)
    outs = {}
    for pack in ("0", "1"):
        monkeypatch.setattr(_cfg.settings, "tower_pack", pack == "1")
        task = make_task(tt, cfg, meta=meta, mlp_dtype="bf16", score_dtype="bf16")
        assert task.two_tower_model.notice_tower.pack_for_score == (pack == "1")
        load_state(task, state)
        task.train()
        res = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        res["loss"].backward()
        grads = {n: p.grad.cpu().numpy() for n, p in task.named_parameters()}
        task.eval()                                   # eval pass: BN from the running statistics, unfused kernels + the pack kernel
        with torch.no_grad():
            ev = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        outs[pack] = (res["loss"].item(), float(res["accuracy"]), grads, ev["loss"].item())
    assert outs["0"][:2] == outs["1"][:2] and outs["0"][3] == outs["1"][3]
    for k, g in outs["0"][2].items():
        assert np.array_equal(outs["1"][2][k], g), k


def test_dropout_mask_statistics(tt, manifest, schema_real):
    """The counter-based dropout mask (tt_uniform01: 32-bit multiply-xorshift over (seed, global element index)): the kept
    fraction is 1 - p, kept values are scaled by 1 / (1 - p), the two towers and two seeds drop different elements, and
    neighbouring elements / rows are uncorrelated."""
    from jodalrob_twotower_amd import towers as TW
    cfg = dict(manifest["cases"]["real_schema"])
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    cfg.update(keys_n=kn, keys_c=kc)
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    shapes = {k: tuple(v) for k, v in manifest["state_dict_keys_real"].items()}
    state = init_state_numpy(shapes, 161)
    B, p, H = 4096, 0.3, cfg["hidden"][1]
    b = synth_batch_numpy(B, vn, vc, cfg["din_n"], cfg["din_c"], 162, oob=False)
    masks = {}
    for seed in (11, 12):
        task = make_task(tt, cfg, meta=GOLD / "real_vocab_metadata.csv", mlp_dtype="bf16", dropout_rate=p)
        for tw in (task.two_tower_model.notice_tower, task.two_tower_model.company_tower):
            tw._seed_override = seed
        load_state(task, state)
        task.train()
        TW._DEBUG_KEEP = []
        try:
            task(to_batch(tt, b, kn, kc), return_metrics=True)["loss"].backward()
            torch.cuda.synchronize()
            for t, k in enumerate(TW._DEBUG_KEEP):           # activation buffer (bf16 x lives elsewhere): pre | act | ...
                al = (B * H + 63) // 64 * 64
                masks[(seed, t)] = (k["acts"][al:al + B * H].view(B, H) != 0).cpu().numpy()
        finally:
            TW._DEBUG_KEEP = None
    for m in masks.values():
        assert abs(m.mean() - (1 - p)) < 0.005
        assert np.abs(m.mean(0) - (1 - p)).max() < 0.05 and np.abs(m.mean(1) - (1 - p)).max() < 0.3      # per column / per row
        z = m.astype(np.float64) - m.mean()
        assert abs((z[:, 1:] * z[:, :-1]).mean()) < 0.003 and abs((z[1:] * z[:-1]).mean()) < 0.003          # neighbours
    same = lambda a, b: (a == b).mean()
    indep = (1 - p) ** 2 + p ** 2
    assert abs(same(masks[(11, 0)], masks[(11, 1)]) - indep) < 0.01        # towers
    assert abs(same(masks[(11, 0)], masks[(12, 0)]) - indep) < 0.01        # seeds


@pytest.mark.parametrize("B,D,inv_t", [(300, 64, 1.0), (257, 16, 4.0), (1024, 200, 0.5)])
def test_score_prescaled_operands(tt, B, D, inv_t):
    """Notice image packed times tt_score_unit_scale(inv_t) (the kernels' unit form: accumulators start at the exponent
    offset, exp2(acc) with no multiply-add) against unscaled images: every output is scale-free -- sums, diagonal, sum of
    scores, ranks and both gradients agree to the rounding of bf16(scale * x) vs bf16(x); an arbitrary scale (the general
    scaled form) likewise."""
    from jodalrob_twotower_amd import ops
    g = torch.Generator().manual_seed(B + D)
    n = torch.nn.functional.normalize(torch.randn(B, D, generator=g), dim=1).to(DEV)
    c = torch.nn.functional.normalize(n.cpu() + 0.5 * torch.randn(B, D, generator=g), dim=1).to(DEV)
    shift, one = abs(inv_t), torch.ones(1, device=DEV)
    ref = None
    for scale in (1.0, ops.score_unit_scale(inv_t), 0.37):
        Np, Cp = ops.score_pack2_bf16(n, c, scale, 1.0)
        rs, cs, dg, rr, cr, ss, inv = ops.score_fwd_bf16(Np, Cp, B, D, inv_t, shift, True, True, scale, with_inv=True)
        dN, dC = ops.score_bwd_bf16(Np, Cp, B, D, inv_t, shift, rs, cs, one, inv_t / (2 * B), scale)
        dN2, dC2 = ops.score_bwd_bf16(Np, Cp, B, D, inv_t, shift, rs, cs, one, inv_t / (2 * B), scale, inv)   # forward's reciprocals
        for x2, x in ((dN2, dN), (dC2, dC)):         # (1 ulp of the reciprocal can flip the bf16 rounding of a softmax weight)
            assert float(torch.linalg.norm(x2 - x)) <= 2e-3 * float(torch.linalg.norm(x))
        out = [x.float().cpu().numpy() for x in (rs, cs, dg, ss, dN, dC)] + [rr.cpu().numpy(), cr.cpu().numpy()]
        if ref is None:
            ref = out
            continue
        for k, (a, b) in enumerate(zip(out[:6], ref[:6])):
            assert np.linalg.norm(a - b) <= 6e-3 * np.linalg.norm(b) + 1e-6, (scale, k, np.linalg.norm(a - b), np.linalg.norm(b))
        for a, b in zip(out[6:], ref[6:]):                                 # ranks move by one where a score sits within bf16
            assert np.abs(a - b).max() <= max(8, B // 50) and np.abs(a - b).mean() < 1.0      # rounding of the positive's


@pytest.mark.parametrize("B,H,D,drop", [(65, 33, 17, 0.1), (127, 64, 64, 0.0), (4097, 40, 33, 0.1), (64, 63, 1, 0.0), (2, 8, 8, 0.0),
                                        (8191, 24, 48, 0.2), (300, 256, 128, 0.1), (129, 200, 100, 0.0), (64, 65, 70, 0.2),
                                        (8192, 256, 128, 0.1), (1000, 96, 128, 0.0), (8300, 250, 120, 0.1)])
def test_fused_tower_tail_odd_shapes(tt, manifest, monkeypatch, B, H, D, drop):
    """Fused tail against the separate kernels on shapes off the tile grid: widths that are not multiples of 8 or 32, ragged
    last row blocks and chunks, a single output column, two rows -- loss bit-identical, gradients to rounding.  Last hidden
    widths up to 256 and outputs up to 128 (scripts/train.py's own [512, 256] -> 128 among them) take the wide kernels (forward:
    tail_fwd_wide_kernel; backward: tail_bwd_wide_kernel + tail_bwd_apply_wide_kernel -- B = 8300 gives chunks of 65 rows: a
    second, one-row block per workgroup)."""
    cfg = dict(manifest["cases"]["wide_b40"])
    cfg.update(hidden=[32, H], D=D)
    outs = {}
    state = None
    b = synth_batch_numpy(B, cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 171, oob=True)
    monkeypatch.setattr(_cfg.settings, "tower_unfused_front", True)
    for unfused in ("1", "0"):
        monkeypatch.setattr(_cfg.settings, "tower_unfused_tail", unfused == "1")
        task = make_task(tt, cfg, mlp_dtype="bf16", score_dtype="bf16", dropout_rate=drop)
        if state is None:
            state = init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 172)
        for tw in (task.two_tower_model.notice_tower, task.two_tower_model.company_tower):
            tw._seed_override = 5
        load_state(task, state)
        task.train()
        res = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        res["loss"].backward()
        outs[unfused] = (res["loss"].item(), {n: p.grad.cpu().numpy() for n, p in task.named_parameters()},
                         {k: v.cpu().numpy() for k, v in task.state_dict().items() if "running" in k})
    wide = H > 64 or D > 64      # the wide kernels: the output Linear is ONE MFMA chain over H, the separate GEMM
    #                              splits K when the batch is small -- same operands, another association: loss to rounding
    assert np.isfinite(outs["0"][0])
    if wide:
        np.testing.assert_allclose(outs["0"][0], outs["1"][0], rtol=2e-6)
    else:
        assert outs["0"][0] == outs["1"][0]
    for k, v in outs["1"][2].items():
        assert np.array_equal(outs["0"][2][k], v), k          # BN running statistics: the same ordered combine on every path
    for k, g in outs["1"][1].items():
        gf = outs["0"][1][k]
        assert np.isfinite(gf).all(), k
        tol = (1e-3 if wide else 2e-5) if B >= 32 else 2e-4            # (BatchNorm over two rows: the backward terms cancel to rounding noise;
        #                                                                   wide: a y that differs in the last bit moves bf16 operand roundings downstream)
        assert np.linalg.norm(gf - g) <= tol * np.linalg.norm(g) + 1e-10, (k, np.linalg.norm(gf - g), np.linalg.norm(g))


@pytest.mark.parametrize("E,nk_n,nk_c,h0,din_n,din_c,H,D,B,drop", [
    (64, 5, 2, 64, 64, 128, 33, 17, 65, 0.1),        # ragged row block, widths off the 32-column grid
    (64, 5, 2, 128, 128, 64, 64, 64, 127, 0.0),
    (32, 3, 1, 32, 64, 64, 40, 33, 4097, 0.1),       # h0 = 32: a lane's k window is projection tile OR looked-up rows per half-wave
    (32, 3, 1, 96, 192, 64, 64, 48, 8191, 0.2),
    (32, 5, 2, 128, 256, 128, 64, 64, 8200, 0.1),    # keys of the synthetic schema at the reference's widths (company falls back: 192 % 64)
])
def test_tower_front_one_launch_vs_separate(tt, manifest, monkeypatch, E, nk_n, nk_c, h0, din_n, din_c, H, D, B, drop):
    """tower_front_kernel (projection + block Linear + chunk BN statistics in one launch, K split over the waves of a workgroup)
    against the same pass as projection GEMM, split-K block GEMM and tail_head_kernel.  Same operands, same rounding points
    (bf16 x, bf16 MFMA operands, f32 accumulate) but another summation order: a projection element that sits on a bf16
    rounding boundary may land on the other side, so the comparison is to tolerance (the parity claim of the measured mode is
    test_bf16_step_vs_rounded_oracle, which runs the one-launch front)."""
    cfg = dict(manifest["cases"]["wide_b40"])
    cfg.update(E=E, hidden=[h0, H], D=D, din_n=din_n, din_c=din_c, keys_n=cfg["keys_n"][:nk_n], keys_c=cfg["keys_c"][:nk_c],
               vocab_n=cfg["vocab_n"][:nk_n], vocab_c=cfg["vocab_c"][:nk_c])
    b = synth_batch_numpy(B, cfg["vocab_n"], cfg["vocab_c"], din_n, din_c, 181, oob=True)
    outs, state = {}, None
    for unfused in ("1", "0"):
        monkeypatch.setattr(_cfg.settings, "tower_unfused_front", unfused == "1")
        task = make_task(tt, cfg, mlp_dtype="bf16", score_dtype="bf16", dropout_rate=drop)
        if state is None:
            state = init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 182)
        for tw in (task.two_tower_model.notice_tower, task.two_tower_model.company_tower):
            tw._seed_override = 5
        load_state(task, state)
        task.train()
        res = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        res["loss"].backward()
        outs[unfused] = (res["loss"].item(), {n: p.grad.cpu().numpy() for n, p in task.named_parameters() if p.grad is not None},
                         {k: v.cpu().numpy() for k, v in task.state_dict().items() if "running" in k})
    worst = {"loss": abs(outs["0"][0] - outs["1"][0]) / abs(outs["1"][0])}
    for k, v in outs["1"][2].items():
        worst["bn:" + k] = float(np.abs(outs["0"][2][k] - v).max() / (np.abs(v).max() + 1e-12))
    for k, g in outs["1"][1].items():
        gf = outs["0"][1][k]
        assert np.isfinite(gf).all(), k
        worst["grad:" + k] = float(np.linalg.norm(gf - g) / (np.linalg.norm(g) + 1e-12))
    print("front one-launch vs separate:", {k: f"{v:.2e}" for k, v in worst.items()})
    assert np.isfinite(outs["0"][0]) and worst["loss"] <= 2e-5, worst
    for k, v in worst.items():
        if k.startswith("bn:"):
            assert v <= 1e-4, (k, v)
        if k.startswith("grad:"):
            tol = 3e-3 if B >= 32 else 3e-2
            assert v <= tol, (k, v)


@pytest.mark.parametrize("B,drop,hidden,D", [(8192, 0.1, None, None), (640, 0.0, None, None), (64, 0.0, None, None),
                                             (1024, 0.1, [128, 128], 128), (512, 0.0, [64, 256], 64), (512, 0.0, [512, 128], 64)])
def test_tower_first_block_backward_one_launch_vs_separate(tt, manifest, schema_real, monkeypatch, B, drop, hidden, D):
    """gemm_back_kernel (block weight gradient, looked-up rows' gradient and G = d_pre^T . dense in one launch; projection
    gradients = W[:, :h0]^T . G in the slab-reduction launch) against the separate TN / NN / TN launches on the reference's
    tower shapes.  The block's weight / bias gradients and the row gradients keep their k order: bit-identical; the projection's
    gradients are the same numbers formed in another order of operations: the separate form rounds d_proj = d_pre . W to bf16
    as a GEMM operand, the one-launch form rounds W and carries d_pre^T . dense unrounded -- they differ by that rounding
    (1.6e-3 norm-wise at B = 8192); each is pinned against the oracle with ITS rounding in test_bf16_step_vs_rounded_oracle."""
    cfg = dict(manifest["cases"]["real_schema"])
    cfg.update(keys_n=schema_real["notice"]["categorical"], keys_c=schema_real["company"]["categorical"])
    if hidden is not None:                      # wider first blocks (H = 128, 256: several 64-row blocks of h per G tile; the last case falls back)
        cfg.update(hidden=hidden, D=D)
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    shapes = {k: tuple(v) for k, v in manifest["state_dict_keys_real"].items()}
    state = init_state_numpy(shapes, 191) if hidden is None else None
    b = synth_batch_numpy(B, vn, vc, cfg["din_n"], cfg["din_c"], 192, oob=False)
    outs = {}
    for unfused in ("1", "0"):
        monkeypatch.setattr(_cfg.settings, "tower_unfused_back", unfused == "1")
        task = make_task(tt, cfg, meta=GOLD / "real_vocab_metadata.csv", mlp_dtype="bf16", score_dtype="bf16", dropout_rate=drop)
        for tw in (task.two_tower_model.notice_tower, task.two_tower_model.company_tower):
            tw._seed_override = 7
        if state is None:
            state = init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 191)
        load_state(task, state)
        task.train()
        res = task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"]), return_metrics=True)
        res["loss"].backward()
        outs[unfused] = (res["loss"].item(), {n: p.grad.cpu().numpy() for n, p in task.named_parameters()})
    assert outs["0"][0] == outs["1"][0]
    worst = {}
    for k, g in outs["1"][1].items():
        gf = outs["0"][1][k]
        assert np.isfinite(gf).all(), k
        if "dense_projection" in k:
            worst[k] = float(np.linalg.norm(gf - g) / (np.linalg.norm(g) + 1e-30))
            # measured: weight 1.6e-3 (what rounding d_proj to bf16 costs the separate form), bias 4.2e-3 (a sum over the batch
            # that cancels; the separate form sums d_proj made from bf16-rounded d_pre and W, the one-launch form rounds nothing)
            assert worst[k] <= (1.5e-2 if k.endswith("bias") else 4e-3), (k, worst[k])
        else:
            assert np.array_equal(gf, g), k
    print("first-block backward, one launch vs separate:", {k: f"{v:.2e}" for k, v in worst.items()})


def test_lookup_profile_ring(tt):
    """ops.LookupProfile: per-launch kernel durations from in-kernel stamps -- one entry per launch, no extra launch, the ring
    keeps the last n_slots launches, durations plausible (0.5 us .. 1 ms)."""
    from jodalrob_twotower_amd import ops
    g = torch.Generator().manual_seed(3)
    B, K, E, R = 2048, 6, 32, 50000
    table = torch.randn(R, E, device=DEV)
    ids = torch.randint(0, R // K, (B * K,), generator=g).to(DEV)
    off = torch.arange(K, dtype=torch.int64, device=DEV) * (R // K)
    voc = torch.full((K,), R // K, dtype=torch.int64, device=DEV)
    x = torch.empty(B, K * E, device=DEV)
    prof = ops.LookupProfile(DEV, n_slots=8)
    try:
        for n_launch in (3, 11):
            prof.reset()
            for _ in range(n_launch):
                ops.embed_lookup(table, [ops.LookupSide(ids, off, voc, x, K)], B, False)
            d = prof.durations_us()
            assert len(d) == min(n_launch, 8), (n_launch, d)
            assert all(0.5 < v < 1000.0 for v in d), d
    finally:
        prof.close()
    ref = table[(ids.view(B, K) + off).view(-1)].view(B, K * E)
    assert torch.equal(x, ref)


def test_copy_multi(tt):
    """tt_copy_multi: several device segments of odd sizes (16-byte body + byte tail) and a pinned-host source."""
    from jodalrob_twotower_amd import ops
    g = torch.Generator().manual_seed(5)
    srcs = [torch.randint(0, 255, (n,), dtype=torch.uint8, generator=g) for n in (1 << 20, 4099, 16, 7, 0, 33)]
    dev_src = [s.to(DEV) for s in srcs[:-1]] + [srcs[-1].pin_memory()]
    dsts = [torch.zeros(s.numel() + 32, dtype=torch.uint8, device=DEV) for s in srcs]
    ops.copy_multi([(d[:s.numel()], s) for d, s in zip(dsts, dev_src)])
    torch.cuda.synchronize()
    for d, s in zip(dsts, srcs):
        assert torch.equal(d[:s.numel()].cpu(), s) and int(d[s.numel():].sum()) == 0


def test_adam_fused_equals_separate_launches(tt, manifest, monkeypatch):
    """tt_adam_fused_step (tower weights + looked-up rows in one launch) == tt_adam_multi_step + tt_sparse_adam_step."""
    from jodalrob_twotower_amd.optim import FusedAdam
    cfg = dict(manifest["cases"]["wide_b40"])
    batches = [synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 300 + s, oob=False) for s in range(3)]
    finals = []
    for fused in (True, False):
        if not fused:
            monkeypatch.setattr(FusedAdam, "_fusable_store", lambda self, *a, **k: None)
        task = make_task(tt, cfg, embedding_grad="sparse")
        shapes = {k: tuple(v.shape) for k, v in task.state_dict().items()}
        load_state(task, init_state_numpy(shapes, 9))
        opt = FusedAdam.for_task(task, lr=3e-3, weight_decay=1e-5)
        task.train()
        for b in batches:
            opt.zero_grad()
            task(to_batch(tt, b, cfg["keys_n"], cfg["keys_c"])).backward()
            opt.step()
        finals.append({k: v.detach().cpu().numpy().copy() for k, v in task.state_dict().items()})
        finals[-1]["__steps"] = np.array(opt.current_step())
    for k, v in finals[0].items():
        assert np.array_equal(v, finals[1][k]), k


@pytest.mark.parametrize("chained", [1, 0])
@pytest.mark.parametrize("G,C,rows_max", [(1, 5000, 100000), (3, 777, 2000), (8, 12345, 250000), (4, 1, 3)])
def test_dedup_plan_runs_equals_general(tt, ctx_option, G, C, rows_max, chained):
    """tt_dedup_plan_runs (stable merge of G ascending runs, as an owner receives them: distinct ascending ids, pads = the
    largest value at the end) == tt_dedup_plan over the concatenation == numpy's stable argsort, bit for bit."""
    from jodalrob_twotower_amd import ops
    ctx_option(_L.TT_OPT_CHAINED, chained, 1)
    rng = np.random.default_rng(G * 1000 + C)
    runs = []
    for g in range(G):
        n = int(rng.integers(0, C + 1))
        ids = np.sort(rng.choice(rows_max, size=min(n, rows_max), replace=False)).astype(np.int32)
        runs.append(np.concatenate([ids, np.full(C - len(ids), rows_max, np.int32)]))       # pad = rows_max (sorts last)
    rows = np.concatenate(runs)
    t = torch.from_numpy(rows).to(DEV)
    pr = ops.dedup_plan_runs(t, G, C)
    pg = ops.dedup_plan(t, rows_max + 1)
    U = int(pr.n_unique.item())
    assert U == int(pg.n_unique.item()) == len(np.unique(rows))
    assert np.array_equal(pr.sorted_src.cpu().numpy(), np.argsort(rows, kind="stable").astype(np.int32))
    assert torch.equal(pr.sorted_src, pg.sorted_src)
    assert torch.equal(pr.unique_rows[:U], pg.unique_rows[:U]) and torch.equal(pr.seg_offsets[:U + 1], pg.seg_offsets[:U + 1])
    # row_limit: the pad group (ids >= rows_max, always last) is left out of the row count, nothing else changes
    pl = ops.dedup_plan_runs(t, G, C, rows_max)
    has_pad = bool((rows == rows_max).any())
    assert int(pl.n_unique.item()) == U - (1 if has_pad else 0)
    assert torch.equal(pl.sorted_src, pr.sorted_src) and torch.equal(pl.unique_rows[:U], pr.unique_rows[:U])
    assert torch.equal(pl.seg_offsets[:U + 1], pr.seg_offsets[:U + 1])


def test_gather_rows(tt):
    from jodalrob_twotower_amd import ops
    g = torch.Generator().manual_seed(3)
    table = torch.randn(1000, 32, generator=g).to(DEV)
    idx = torch.randint(-5, 1010, (4097,), generator=g, dtype=torch.int32).to(DEV)
    out = ops.gather_rows(table, idx)
    exp = table[idx.long().clamp(0, 999)].clone()
    exp[idx < 0] = 0                                         # negative index: a zero row (unused bucket entries)
    assert torch.equal(out, exp)
    assert torch.equal(ops.gather_rows(table, idx, torch.bfloat16), exp.to(torch.bfloat16))      # RNE on the way out


def test_full_size_properties(tt, schema_real):
    """BASELINE configs[1] at full size (B = 8192, 38 real keys, 1 M + 1 M rows, E = 32) through size-independent properties:
    lookup == table[rows] bit for bit (f32) / exact RNE (bf16); the plan is a stable sort of the rows; the segmented
    reduction conserves the column sums of the slot gradients; the symmetric loss is invariant under swapping the towers;
    a whole training step is bitwise reproducible."""
    from jodalrob_twotower_amd import ops, synthetic
    B, E = 8192, 32
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    vn = synthetic.scale_vocabs(schema_real["notice"]["vocab_sizes"], 1_000_000)
    vc = synthetic.scale_vocabs(schema_real["company"]["vocab_sizes"], 1_000_000)
    batch = synthetic.make_batch(B, vn, vc, kn, kc, 256, 128, torch.device(DEV), seed=99)
    R = sum(vn) + sum(vc)
    g = torch.Generator(device=DEV).manual_seed(5)
    table = torch.randn((R, E), generator=g, device=DEV)
    offs, base = [], 0
    for v in (vn, vc):
        offs.append(torch.tensor([base + sum(v[:i]) for i in range(len(v))], dtype=torch.int64, device=DEV))
        base += sum(v)
    vocs = [torch.tensor(v, dtype=torch.int64, device=DEV) for v in (vn, vc)]
    ids = [batch["notice"]["kjt"].values(), batch["company"]["kjt"].values()]
    Ks = [len(vn), len(vc)]
    exp_rows = torch.cat([(ids[i].view(B, Ks[i]).clamp(min=0).minimum(vocs[i][None, :] - 1) + offs[i][None, :]).reshape(-1)
                          for i in range(2)])
    for dt in (torch.float32, torch.bfloat16):
        outs = [torch.empty((B, 128 + Ks[i] * E), dtype=dt, device=DEV) for i in range(2)]
        sides = [ops.LookupSide(ids[i], offs[i], vocs[i], outs[i][:, 128:], Ks[i]) for i in range(2)]
        rows = ops.embed_lookup(table, sides, B, want_rows=True)
        assert torch.equal(rows.long(), exp_rows)
        for i in range(2):
            sl = exp_rows[:B * Ks[0]] if i == 0 else exp_rows[B * Ks[0]:]
            assert torch.equal(outs[i][:, 128:].reshape(B * Ks[i], E), table[sl].to(dt))
    plan = ops.dedup_plan_keyed(rows, Ks, B)
    U = int(plan.n_unique.item())
    srt = rows[plan.sorted_src.long()]
    assert bool((srt[1:] >= srt[:-1]).all()) and torch.equal(torch.sort(plan.sorted_src).values, torch.arange(rows.numel(), dtype=torch.int32, device=DEV))
    same = srt[1:] == srt[:-1]
    assert bool((plan.sorted_src[1:][same] > plan.sorted_src[:-1][same]).all())              # stable
    assert torch.equal(plan.unique_rows[:U].long(), torch.unique(exp_rows))
    d = [torch.randn((B, 128 + Ks[i] * E), generator=g, device=DEV) for i in range(2)]
    grad_rows = torch.empty((rows.numel(), E), device=DEV)
    ops.embed_grad(plan, [(d[i][:, 128:], Ks[i]) for i in range(2)], B, E, ops.TT_GRAD_SPARSE, grad_rows)
    tot = sum(d[i][:, 128:].reshape(-1, E).double().sum(0) for i in range(2))
    torch.testing.assert_close(grad_rows[:U].double().sum(0), tot, rtol=1e-6, atol=1e-3)
    # symmetric loss: swapping the towers leaves loss untouched and swaps the directional metrics
    n = torch.nn.functional.normalize(torch.randn((B, 64), generator=g, device=DEV), dim=1)
    c = torch.nn.functional.normalize(torch.randn((B, 64), generator=g, device=DEV), dim=1)
    Np, Cp = ops.score_pack2_bf16(n, c)
    f1 = ops.score_fwd_bf16(Np, Cp, B, 64, 1.0, 1.0, True, True)
    f2 = ops.score_fwd_bf16(Cp, Np, B, 64, 1.0, 1.0, True, True)
    assert torch.equal(f1[0], f2[1]) and torch.equal(f1[1], f2[0]) and torch.equal(f1[3], f2[4])   # rowsum <-> colsum, ranks
    o1, l1 = ops.score_loss_finish(B, 1.0, *f1)
    o2, l2 = ops.score_loss_finish(B, 1.0, *f2)
    assert l1.item() == l2.item()


def test_full_size_step_is_reproducible(tt, schema_real, tmp_path):
    """configs[1]'s own shapes (B = 8192, 1 M + 1 M rows): two eager steps twice -- the same bits; and the same two steps REPLAYED
    from a captured graph (round 4: the hand-over launch leaves the lookup its precomputed rows and refreshes the bf16 shadows of the
    tower weights that the one-launch front reads instead of the f32 weights) -- the same bits again, losses and every parameter."""
    from jodalrob_twotower_amd import synthetic
    from jodalrob_twotower_amd.graph import GraphedTrainStep
    from jodalrob_twotower_amd.optim import FusedAdam
    B = 8192
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    vn = synthetic.scale_vocabs(schema_real["notice"]["vocab_sizes"], 1_000_000)
    vc = synthetic.scale_vocabs(schema_real["company"]["vocab_sizes"], 1_000_000)
    meta = synthetic.write_metadata(tmp_path / "m.csv", {"notice": dict(zip(kn, vn)), "company": dict(zip(kc, vc))})
    batches = [synthetic.make_batch(B, vn, vc, kn, kc, 256, 128, torch.device(DEV), seed=7 + i) for i in range(2)]
    finals = []
    for mode in ("eager", "eager", "graph"):
        torch.manual_seed(11)
        task = tt.create_two_tower_train_task(kn, kc, metadata_path=str(meta), categorical_embedding_dim=32, notice_dense_input_dim=256,
                                              company_dense_input_dim=128, tower_hidden_dims=[128, 64], final_embedding_dim=64,
                                              dropout_rate=0.0, temperature=1.0, device=DEV, embedding_grad="sparse", score_dtype="bf16",
                                              mlp_dtype="bf16")
        task.train(); task._pair_check_done = True
        opt = FusedAdam.for_task(task, lr=1e-3, weight_decay=1e-5)
        losses = []
        if mode == "graph":
            gs = GraphedTrainStep(task, opt, batches[0], warmup=1)          # (its warm-up step leaves no trace: preserve_state)
            assert gs._rows_sm is not None and len(gs._shadows) == 2 and all(len(sh) == 2 for _, _, sh in gs._shadows)
            for b in batches:
                losses.append(gs.step(b)["loss"].item())
            for t_, ws, sh in gs._shadows:                                   # the shadows hold bf16(weights as of the last hand-over)
                assert t_._w16 is None
            gs.close()
        else:
            for b in batches:
                opt.zero_grad()
                r = task(b, return_metrics=True)
                r["loss"].backward()
                opt.step()
                losses.append(r["loss"].item())
        finals.append((losses, [p.detach().clone() for p in task.parameters()]))
        del task, opt
    assert finals[0][0] == finals[1][0] and finals[0][0][0] > 8.5           # ln(8192) = 9.01 at random init
    assert finals[2][0] == finals[0][0], (finals[2][0], finals[0][0])
    for a, b, c in zip(finals[0][1], finals[1][1], finals[2][1]):
        assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("G,M,U,C", [(1, 5000, 4000, 4096), (2, 70000, 65000, 40000), (3, 10000, 9999, 3000), (8, 311296, 65731, 12288),
                                     (64, 9000, 9000, 256), (5, 4097, 0, 16), (4, 6000, 6000, 1000), (8, 400000, 200000, 30000)])
@pytest.mark.parametrize("chained", [1, 0])
def test_route_bucket_and_expand(tt, ctx_option, G, M, U, C, chained):
    """tt_route_bucket / tt_route_expand against a plain numpy routing: stable order inside every owner's bucket, pads,
    counts, the overflow flag (capacity too small in one case), and the slot -> bucket-position map; tt_route_bucket_expand gives
    all of it from one call (run twice).  One synthetic plan row has 3000 slots (a hot row).  (chained: TT_OPT_CHAINED only
    concerns the plans' head pass; the routing launches are the same.)"""
    from jodalrob_twotower_amd import ops
    ctx_option(_L.TT_OPT_CHAINED, chained, 1)
    rng = np.random.default_rng(G * 7 + M)
    uniq = np.sort(rng.choice(5_000_000, size=U, replace=False)).astype(np.int32)
    # a synthetic plan: U distinct ascending rows, every slot assigned to one of them
    slot_u = np.concatenate([np.arange(U), rng.integers(0, max(U, 1), M - U)]) if U else np.zeros(0, np.int64)
    if U and M - U > 3000:
        slot_u[U:U + 3000] = U // 2                                        # a hot row
    if U:
        rng.shuffle(slot_u)
        order = np.argsort(slot_u, kind="stable").astype(np.int32)
        seg = np.concatenate([np.searchsorted(slot_u[order], np.arange(U)), [M]]).astype(np.int32)
    else:
        order, seg = np.zeros(M, np.int32), np.zeros(1, np.int32)
    buf_u = np.full(M, -7, np.int32); buf_u[:U] = uniq
    buf_s = np.full(M + 1, -7, np.int32); buf_s[:len(seg)] = seg
    plan = ops.DedupPlan(torch.from_numpy(order).to(DEV), torch.from_numpy(buf_u).to(DEV), torch.from_numpy(buf_s).to(DEV),
                         torch.tensor([U], dtype=torch.int32, device=DEV), M)
    pads = [1_000_000 + g for g in range(G)]
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    send_ids, send_u, pos_u, counts = ops.route_bucket(plan, G, C, pads, -1, flag)
    e_ids = np.repeat(np.asarray(pads, np.int32), C); e_u = np.full(G * C, -1, np.int32)
    e_pos = np.full(U, G * C, np.int32); cnt = np.zeros(G, np.int64)      # rows that do not fit: the zero row behind the buckets
    for u in range(U):
        g = int(uniq[u]) % G
        p = cnt[g]; cnt[g] += 1
        if p < C:
            e_ids[g * C + p] = uniq[u] // G; e_u[g * C + p] = u; e_pos[u] = g * C + p
    assert np.array_equal(counts.cpu().numpy(), cnt.astype(np.int32))
    assert bool(flag.item()) == bool((cnt > C).any())
    assert np.array_equal(send_ids.cpu().numpy(), e_ids) and np.array_equal(send_u.cpu().numpy(), e_u)
    assert np.array_equal(pos_u[:U].cpu().numpy(), e_pos)
    if U:
        idx = ops.route_expand(plan, pos_u).cpu().numpy()
        assert np.array_equal(idx, e_pos[slot_u].astype(np.int64))
        for _ in range(2):
            flag2 = torch.zeros(1, dtype=torch.int32, device=DEV)
            s2, u2, p2, c2, idx2 = ops.route_bucket(plan, G, C, pads, -1, flag2, expand=True)
            assert torch.equal(s2, send_ids) and torch.equal(u2, send_u) and torch.equal(p2[:U], pos_u[:U]) and torch.equal(c2, counts)
            assert flag2.item() == flag.item() and np.array_equal(idx2.cpu().numpy(), idx)


class _ThreadComm:
    """In-process stand-in for the collectives: G threads = G virtual ranks on ONE GPU (each on its own stream), buffers
    handed over through a shared table between two barriers."""

    def __init__(self, world, rank, shared):
        self.world, self.rank, self.sh = world, rank, shared

    def _exchange(self, t):
        torch.cuda.current_stream().synchronize()
        self.sh["slots"][self.rank] = t
        self.sh["bar"].wait()
        got = list(self.sh["slots"])
        self.sh["bar"].wait()
        return got

    def all_to_all_equal(self, send):
        parts = self._exchange(send.contiguous())
        n = send.shape[0] // self.world
        return torch.cat([p[self.rank * n:(self.rank + 1) * n] for p in parts]).clone()

    def all_reduce_max(self, t):
        return torch.stack(self._exchange(t.clone())).max(0).values

    def all_gather(self, t):
        return torch.cat(self._exchange(t.contiguous())).clone()

    def all_reduce_sum(self, t):
        t.copy_(torch.stack(self._exchange(t.clone())).sum(0))
        return t


@pytest.mark.parametrize("G,x_dtype,grad_bf16", [(2, torch.float32, False), (3, torch.float32, False), (3, torch.bfloat16, False),
                                                 (2, torch.bfloat16, True)])
def test_padded_exchange_multi_rank_on_one_gpu(tt, G, x_dtype, grad_bf16):
    """The fixed-capacity exchange with the REAL HIP steps (plan, tt_route_bucket, tt_gather_rows, tt_route_expand, placing
    lookup, local + owner-side reductions, tt_dedup_plan_runs) for G > 1: G virtual ranks as threads on one GPU.  Forward:
    every rank's tower inputs == a direct gather from the unsharded table (bit-exact; with bf16 tower inputs the rows
    travel as bf16 and must equal the RNE rounding of the table rows).  Backward: every owner's summed row gradients ==
    the global scatter-add over ALL ranks' slots restricted to its rows."""
    import threading
    from jodalrob_twotower_amd import ops
    from jodalrob_twotower_amd.distributed import PaddedRowExchange, ShardedStore
    E, B = 32, 700
    vocabs = [[5, 2000, 3, 40], [7, 300]]
    R = sum(map(sum, vocabs))
    rng = np.random.default_rng(11)
    table = torch.from_numpy(rng.standard_normal((R, E)).astype(np.float32))
    shared = {"slots": [None] * G, "bar": threading.Barrier(G)}
    results, errors = [None] * G, []

    def rank_fn(rank):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=DEV)):
                store = ShardedStore(E, R, rank, G, DEV, "sparse")
                store.load_global(table)
                ex = PaddedRowExchange(store, comm=_ThreadComm(G, rank, shared))
                ex.grad_wire_bf16 = grad_bf16                  # opt-in: row gradients as bf16 on the wire
                r2 = np.random.default_rng(500 + rank)
                sides, outs, rows_ref, base = [], [], [], 0
                for v in vocabs:
                    K = len(v)
                    ids = np.stack([r2.integers(-2, vk + 2, B) for vk in v], axis=1).astype(np.int64)
                    off = np.concatenate([[0], np.cumsum(v)[:-1]]) + base
                    base += sum(v)
                    out = torch.zeros((B, 16 + K * E), dtype=x_dtype, device=DEV)[:, 16:]      # a view inside a wider buffer, as x
                    sides.append(ops.LookupSide(torch.from_numpy(ids.reshape(-1)).to(DEV), torch.from_numpy(off.astype(np.int64)).to(DEV),
                                                torch.tensor(v, dtype=torch.int64, device=DEV), out, K))
                    outs.append(out)
                    rows_ref.append((np.minimum(np.maximum(ids, 0), np.array(v)[None, :] - 1) + off[None, :]).reshape(-1))
                state = ex.forward(sides, B, True)
                fwd_ok = all(torch.equal(o.cpu(), table[torch.from_numpy(rr)].view(B, -1).to(x_dtype)) for o, rr in zip(outs, rows_ref))
                d = [torch.from_numpy(r2.standard_normal((B, s.K * E)).astype(np.float32)).to(DEV) for s in sides]
                ex.backward(state, [(dd, s.K) for dd, s in zip(d, sides)], B)
                plan, grad_rows = store.sparse_grad
                torch.cuda.current_stream().synchronize()
                U = int(plan.n_unique.item())
                results[rank] = dict(fwd_ok=fwd_ok, overflow=ex.overflowed(), rows=np.concatenate(rows_ref),
                                     vals=np.concatenate([dd.cpu().numpy().reshape(-1, E) for dd in d]),
                                     uniq=plan.unique_rows[:U].cpu().numpy(), grads=grad_rows[:U].cpu().numpy(), local_rows=store.local_rows)
        except Exception:                                       # pragma: no cover
            import traceback
            errors.append(traceback.format_exc())
            shared["bar"].abort()

    threads = [threading.Thread(target=rank_fn, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors[0]
    ref = np.zeros((R, E), np.float64)
    for r in results:
        assert r["fwd_ok"] and not r["overflow"]
        np.add.at(ref, r["rows"], r["vals"].astype(np.float64))
    for rank, r in enumerate(results):
        mine = ref[rank::G]
        got = np.zeros_like(mine)
        real = r["uniq"] < r["local_rows"]                     # the last distinct "row" is the bucket pad
        assert (~real).sum() <= 1
        got[r["uniq"][real]] = r["grads"][real]
        if grad_bf16:                                          # every rank's row sum was rounded to bf16 before the owner added them
            assert np.linalg.norm(got - mine) <= 4e-3 * np.linalg.norm(mine)
        else:
            np.testing.assert_allclose(got, mine, rtol=1e-5, atol=1e-5)
        assert set(np.flatnonzero(np.abs(mine).sum(1) > 0)) <= set(r["uniq"][real].tolist())


@pytest.mark.parametrize("hidden,D,mlp", [([128, 64], 64, "bf16"), ([512, 256], 128, "bf16"), ([96, 200], 70, "fp32")])
def test_sync_bn_phases_at_world_one_equal_the_whole_pass(tt, manifest, schema_real, hidden, D, mlp):
    """tt_tower_params.sync_phase 1 + 2 with ONE rank (the all-gather hands back the rank's own statistics) == the uncut
    pass, bit for bit: unit rows, BN running statistics and every gradient, dropout on.  [128, 64] -> 64 takes the fused
    tail kernels; scripts/train.py's own [512, 256] -> 128 (/root/reference/scripts/train.py:106-107) and an odd fp32 shape
    take the separate kernels, cut at the same place."""
    class Solo:
        world, rank = 1, 0

        def all_gather(self, t):
            return t.clone()

    cfg = dict(manifest["cases"]["real_schema"])
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    cfg.update(keys_n=kn, keys_c=kc, hidden=hidden, D=D)
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    b = synth_batch_numpy(333, vn, vc, cfg["din_n"], cfg["din_c"], 152, oob=False)
    outs, state = [], None
    for comm in (None, Solo()):
        task = make_task(tt, cfg, meta=GOLD / "real_vocab_metadata.csv", mlp_dtype=mlp, score_dtype="bf16", dropout_rate=0.1)
        for tw in (task.two_tower_model.notice_tower, task.two_tower_model.company_tower):
            tw._seed_override, tw.sync_comm = 77, comm
        if state is None:
            state = init_state_numpy({k: tuple(v.shape) for k, v in task.state_dict().items()}, 151)
        load_state(task, state)
        task.train()
        res = task(to_batch(tt, b, kn, kc), return_metrics=True)
        res["loss"].backward()
        outs.append((res["loss"].item(), {n: p.grad.cpu().numpy() for n, p in task.named_parameters()},
                     {k: v.cpu().numpy() for k, v in task.state_dict().items() if "running" in k}))
    assert outs[0][0] == outs[1][0]
    for k, v in outs[0][2].items():
        assert np.array_equal(outs[1][2][k], v), k
    for k, g in outs[0][1].items():
        assert np.array_equal(outs[1][1][k], g), k


def test_two_processes_equal_single_process(tt):
    """SURVEY 8(e)'s parity definition on one GPU: a 2-rank job (two processes sharing the device, collectives through gloo
    with host staging) with row-wise sharded tables behind the fixed-capacity exchange, global in-batch negatives and SyncBN
    == the single-process task on the same global batch: loss, dense gradients, BN running statistics, parameters after one
    Adam step (tests/_dist_world2_worker.py); then the overflow path with real peers: a 16-row bucket capacity raises
    ExchangeOverflowError on every rank, reset_capacity() recalibrates and the next step is clean.  (Threads cannot stand in for ranks here: the backward passes of all
    threads run on autograd's one device thread, so a collective inside backward deadlocks.)"""
    import os, socket, subprocess, sys
    from pathlib import Path
    worker = Path(__file__).resolve().parent / "_dist_world2_worker.py"

    def run_pair(extra_env):
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        env = dict(os.environ, **extra_env)
        procs = [subprocess.Popen([sys.executable, str(worker), str(r), "2", str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                  text=True, env=env) for r in range(2)]
        outs = []
        try:
            for p in procs:
                outs.append(p.communicate(timeout=240) + (p.returncode,))
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        return outs

    ctrl = run_pair({"TT_W2_NO_SYNC": "1"})          # negative control: per-rank BN statistics must NOT pass the same checks
    assert not any("DIST_WORLD2_OK" in o[0] for o in ctrl), "the check does not see the BN statistics"
    wide = run_pair({"TT_W2_WIDE": "1"})               # towers [48, 80] -> 72: SyncBN on the separate (unfused) kernels
    assert all("DIST_WORLD2_OK" in o[0] for o in wide) and len(wide) == 2, "\n".join(o[1][-1500:] for o in wide)
    outs = run_pair({})
    ok = all("DIST_WORLD2_OK" in o[0] and o[2] == 0 for o in outs) and len(outs) == 2       # rc 0: ordinary teardown, no os._exit escape
    if not ok:
        log = Path(__file__).resolve().parents[1] / "gpurun_out"
        log.mkdir(exist_ok=True)
        (log / "dist_world2_worker.log").write_text("\n".join(f"==== rank {i} stdout ====\n{o[0]}\n==== stderr ====\n{o[1]}" for i, o in enumerate(outs)))
    assert ok, "\n".join(o[1][-1500:] for o in outs)


def test_segmented_capture_with_real_peers_equals_eager(tt):
    """The segmented fallback (segmented.SegmentedTrainStep) with TWO ranks exchanging rows in every step (two processes on the one
    GPU, collectives staged through gloo between the replayed segments): on each rank 4 steps == the same steps launch by launch,
    bit for bit -- losses and the rank's whole state -- on per-rank negatives and on global negatives + SyncBN
    (tests/_dist_world2_worker.py: segmented_equals_eager)."""
    import os, socket, subprocess, sys
    from pathlib import Path
    worker = Path(__file__).resolve().parent / "_dist_world2_worker.py"
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ, TT_W2_SEGMENTED="1")
    procs = [subprocess.Popen([sys.executable, str(worker), str(r), "2", str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300) + (p.returncode,))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    ok = len(outs) == 2 and all("DIST_WORLD2_SEGMENTED_OK" in o[0] and o[2] == 0 for o in outs)
    if not ok:
        log = Path(__file__).resolve().parents[1] / "gpurun_out"
        log.mkdir(exist_ok=True)
        (log / "dist_world2_segmented.log").write_text("\n".join(f"==== rank {i} stdout ====\n{o[0]}\n==== stderr ====\n{o[1]}" for i, o in enumerate(outs)))
    assert ok, "\n".join(o[0][-800:] + o[1][-1500:] for o in outs)


@pytest.mark.parametrize("G,B,D", [(2, 300, 64), (3, 129, 32), (4, 256, 64)])
def test_global_negatives_equal_single_process(tt, G, B, D):
    """Global in-batch negatives on G virtual ranks == the single-process loss over the concatenated batch of G*B pairs
    (the reference's semantics at the global batch): loss = mean of the ranks' losses, every rank's gradient = G x its
    slice of the single-process gradient (the towers scale by 1/G afterwards), metrics from the same rows."""
    import threading
    from jodalrob_twotower_amd.distributed import _GlobalScoreCEFn
    from jodalrob_twotower_amd.two_tower_train_task import _ScoreCEFn
    g = torch.Generator().manual_seed(G * 100 + B)
    n_all = torch.nn.functional.normalize(torch.randn(G * B, D, generator=g), dim=1).to(DEV)
    c_all = torch.nn.functional.normalize(n_all.cpu() + 0.7 * torch.randn(G * B, D, generator=g), dim=1).to(DEV)
    inv_t = 2.0
    ns, cs = n_all.clone().requires_grad_(True), c_all.clone().requires_grad_(True)
    loss_s, out_s, _ = _ScoreCEFn.apply(ns, cs, inv_t, "bf16", True, True, None, None, 1.0)      # unscaled images, as the global form packs them
    loss_s.backward()
    shared = {"slots": [None] * G, "bar": threading.Barrier(G)}
    res, errors = [None] * G, []

    def rank_fn(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=DEV)):
                n = n_all[r * B:(r + 1) * B].clone().requires_grad_(True)
                c = c_all[r * B:(r + 1) * B].clone().requires_grad_(True)
                loss, out8, rank = _GlobalScoreCEFn.apply(n, c, inv_t, _ThreadComm(G, r, shared))
                loss.backward()
                torch.cuda.current_stream().synchronize()
                res[r] = (loss.item(), n.grad.cpu(), c.grad.cpu(), out8.cpu(), rank.cpu())
        except Exception:                                       # pragma: no cover
            import traceback
            errors.append(traceback.format_exc())
            shared["bar"].abort()

    th = [threading.Thread(target=rank_fn, args=(r,)) for r in range(G)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errors, errors[0]
    np.testing.assert_allclose(np.mean([x[0] for x in res]), loss_s.item(), rtol=2e-6)
    for r, (l, dn, dc, out8, rank) in enumerate(res):
        torch.testing.assert_close(dn, G * ns.grad[r * B:(r + 1) * B].cpu(), rtol=2e-5, atol=2e-7)
        torch.testing.assert_close(dc, G * cs.grad[r * B:(r + 1) * B].cpu(), rtol=2e-5, atol=2e-7)
    acc = np.mean([float(x[3][1]) for x in res])
    np.testing.assert_allclose(acc, float(out_s[1]), rtol=1e-6)                     # row top-1 rate over the global batch
    np.testing.assert_allclose(np.mean([float(x[3][3]) for x in res]), float(out_s[3]), rtol=1e-4, atol=1e-6)   # negative mean


# ------------------------------------------------------------------------------- the MEASURED mode, pinned directly
def _rel(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300))


# Per-tensor bounds of the bf16 step against the f64 oracle fed the SAME bf16-rounded operands (oracle_np.task_step(
# rounding="bf16")): what is left is f32 accumulation order, the hardware exp2 / reciprocal, and the rare element whose
# f32 value sits on the other side of a bf16 rounding boundary than the f64 one (each such flip is a 2^-8 relative change
# of ONE operand element, so max-abs bounds are looser than norm-wise ones).  DESIGN.md section 4 quotes this table.
BF16_STEP_BOUNDS = {
    "loss_rtol": 1e-6,                 # |loss - ref| / ref                                   (measured 9e-10 / 5e-8)
    "emb_norm": 1e-4, "emb_maxabs": 1.5e-3,        # unit rows [B, D]                         (measured 2.9e-5 / 3.7e-4)
    "metric_atol": 2e-6,               # positive / negative similarity means, gap
    "dense_grad_matrix_norm": 2e-3,    # Linear weights  (norm-wise, per tensor)              (measured <= 5.2e-4)
    "dense_grad_vector_norm": 2e-3,    # biases, BN scale / shift (column sums over the batch)  (measured <= 5.6e-4)
    "row_grad_norm": 1e-3,             # sparse table gradient rows (all touched rows, norm-wise); row SET bit-exact (measured 2.4e-4)
}


# score_dtype="fp8" (BASELINE configs[4]): the same quantities against the oracle with e4m3 score operands.  Embeddings do not
# depend on the score kernels; the loss sees e4m3 products accumulated in f32 over D = 256.  "fp8_s" (TT_OPT_FP8_GRAD 0): the
# gradient products as in the bf16 case (bf16 softmax weights, bf16 operands) -- measured at B = 2048 / T = 1 and B = 1000 /
# T = 0.5: loss 1e-7 / 4e-7, metrics <= 2.7e-6 (the diagonal is a sum of D = 256 e4m3 products in f32, twice as large at
# T = 0.5), dense gradients <= 1.0e-3, rows 5.3e-4.  "fp8" (the default): block-scaled e4m3 softmax weights and e4m3
# operands in the gradient products too; the score node itself matches its oracle model to 1.4e-5
# (test_score_fp8_vs_rounded_oracle), but a weight that rounds the other way than the f64 oracle's now moves by 2^-4 of itself
# instead of 2^-9, and the last layer's bias gradient is the column sum of d(embedding) over the batch -- a sum whose terms
# cancel to ~1/400 of their norm -- so the vector bound is 6e-3 here (measured: mlp.4.bias 2.6e-3, mlp.2.bias 1.8e-3, every
# matrix <= 5.7e-4, rows 5.1e-4)
FP8_STEP_BOUNDS = dict(BF16_STEP_BOUNDS, loss_rtol=2e-5, metric_atol=6e-6, dense_grad_vector_norm=6e-3)
FP8S_STEP_BOUNDS = dict(BF16_STEP_BOUNDS, loss_rtol=2e-5, metric_atol=6e-6)

# The same quantities against the REFERENCE'S OWN ARITHMETIC -- oracle_np.task_step(rounding=None): every Linear, the score
# matrix and its gradients in unrounded f64, i.e. the reference's f32 forward / backward (two_tower_train_task.py:99-134,
# base_tower.py:83-99) without any operand rounding -- at the benchmarked shape.  This is the stated tolerance of the measured
# mode (north_star: "stated fp tolerance on embeddings / loss"); DESIGN.md section 4 quotes the measured column.
# Measured (round 4, B = 8192, 1 M + 1 M rows): loss 3.0e-7; embeddings 3.5e-3 / 3.7e-3 norm-wise (= bf16's 2^-9 per operand
# element through three Linears), 2.8e-3 max-abs; metrics <= 1.3e-5; accuracy 1 / 8192; Linear weights 5.2e-3 ... 5.7e-3 behind
# the BatchNorm (mlp.2, mlp.4) and 3.9e-2 ... 4.2e-2 in front of it (dense_projection, mlp.0); table-gradient rows 4.1e-2; bias /
# BN vectors 1.1e-2 ... 1.0e-1.  Why the first block's gradients and the vectors sit an order above the operands' 2^-9: at a
# random initialisation the loss is ln B and every gradient is a sum over the batch whose terms cancel -- the BatchNorm backward
# removes the batch mean and the batch-variance direction, the softmax's dS sums to zero along rows and columns -- so the
# independent 2^-9 roundings of the 8192 terms (sqrt(B) growth) stand against a sum that cancelled to about a tenth of its
# random-walk size; column sums (the bias / BN gradients) cancel hardest.  Against the oracle with the SAME operand rounding these
# tensors sit at <= 9.7e-4 (BF16_STEP_BOUNDS): the kernels compute the rounded arithmetic they claim, and the distance from the
# reference is bf16 operand rounding itself, not accumulation or kernel error.  The vector bound (1.2e-1) is below round 3's
# 1.5e-1 at B = 512; it cannot go lower without wider operands (the reference's own TF32 matmuls -- scripts/train.py:147-148 --
# round to 2^-11: a quarter of these figures).
BF16_VS_REFERENCE_BOUNDS = {"loss_rtol": 2e-6, "emb_norm": 6e-3, "emb_maxabs": 5e-3, "metric_atol": 5e-5, "dense_grad_matrix_norm": 6e-2,
                            "dense_grad_vector_norm": 1.2e-1, "row_grad_norm": 6e-2}
# configs[4]'s arithmetic (fp8 score operands + block-scaled e4m3 gradient products) at B = 2048, D = 256.  Measured: loss 3.1e-6;
# embeddings 3.5e-3 / 3.8e-3 (they do not depend on the score kernels); metrics <= 4.6e-5; Linear weights 5.9e-3 ... 8.4e-3 behind
# the BatchNorm, 3.8e-2 ... 4.5e-2 in front of it; rows 4.4e-2; vectors 6.8e-2 ... 1.2e-1 in the first block and 1.8e-1 ... 3.8e-1
# for the BatchNorm shift and the last layer's bias: those two are column sums of d(embedding) over the batch, which cancel to
# ~1/400 of their terms' norm (FP8_STEP_BOUNDS above), so the e4m3 weights' 2^-4 roundings show there and nowhere else
# (TT_OPT_FP8_GRAD 0 -- bf16 gradient products -- puts them back at the bf16 figures).
FP8_VS_REFERENCE_BOUNDS = dict(BF16_VS_REFERENCE_BOUNDS, loss_rtol=1e-5, metric_atol=1e-4, dense_grad_vector_norm=5e-1)


@pytest.mark.parametrize("rows_per_tower,B,T,hidden,D,score_dtype", [(1_000_000, 8192, 1.0, [128, 64], 64, "bf16"), (None, 1000, 0.5, [128, 64], 64, "bf16"),
                                                                     (None, 2240, 1.0, [512, 256], 128, "bf16"), (None, 4096, 0.7, [256, 128], 96, "bf16"),
                                                                     (None, 2048, 1.0, [128, 64], 256, "fp8"), (None, 1000, 0.5, [128, 64], 256, "fp8"),
                                                                     (None, 2048, 1.0, [128, 64], 256, "fp8_s")])
def test_bf16_step_vs_rounded_oracle(tt, schema_real, tmp_path, ctx_option, rows_per_tower, B, T, hidden, D, score_dtype):
    """ONE step of exactly bench.py's task (real 32 + 6 key schema, vocabularies scaled to 1 M + 1 M rows, B = 8192, E = 32,
    towers [128, 64] -> 64, mlp_dtype = score_dtype = "bf16", embedding_grad = "sparse"; dropout 0 so that the oracle needs
    no mask) against the f64 oracle with the kernels' operand rounding: loss, both towers' embeddings, the metrics, every
    dense gradient and the sparse row gradients, each with its own stated bound.  Second case: the real (unscaled)
    vocabularies at a ragged batch and T = 0.5.  Third case: scripts/train.py's own towers ([512, 256] -> 128,
    /root/reference/scripts/train.py:106-107) -- the wide tail kernels and the separate fast GEMMs of the first block (the
    one-launch front / first-block backward do not take h0 = 512), at a batch of 35 x 64 rows.  Last three cases: BASELINE
    configs[4]'s step -- final_embedding_dim 256, score_dtype "fp8" (e4m3 operands for the score products and, by default,
    for the gradient products with block-scaled e4m3 softmax weights; "fp8_s": TT_OPT_FP8_GRAD 0, bf16 gradient products) -- at
    batches the f64 oracle can hold (the same step at B = 65536: test_configs4_whole_step_full_size)."""
    from jodalrob_twotower_amd import synthetic
    score_rounding = score_dtype if score_dtype.startswith("fp8") else None          # the oracle's name of the score arithmetic
    if score_dtype == "fp8_s":                                                        # fp8 S products, bf16 gradient products
        ctx_option(_L.TT_OPT_FP8_GRAD, 0, 1)
        score_dtype = "fp8"
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    vn, vc = schema_real["notice"]["vocab_sizes"], schema_real["company"]["vocab_sizes"]
    if rows_per_tower:
        vn, vc = synthetic.scale_vocabs(vn, rows_per_tower), synthetic.scale_vocabs(vc, rows_per_tower)
    meta = synthetic.write_metadata(tmp_path / "m.csv", {"notice": dict(zip(kn, vn)), "company": dict(zip(kc, vc))})
    torch.manual_seed(1234)
    task = tt.create_two_tower_train_task(kn, kc, metadata_path=str(meta), categorical_embedding_dim=32, notice_dense_input_dim=256,
                                          company_dense_input_dim=128, tower_hidden_dims=hidden, final_embedding_dim=D,
                                          dropout_rate=0.0, temperature=T, device=DEV, embedding_grad="sparse", score_dtype=score_dtype,
                                          mlp_dtype="bf16")
    task.train()
    task._pair_check_done = True
    # weights away from the init's symmetric spots: BN scale / shift and biases random, so their gradients are exercised
    with torch.no_grad():
        g = torch.Generator(device=DEV).manual_seed(77)
        for n_, p in task.named_parameters():
            if p.ndim == 1:
                p.add_(0.1 * torch.randn(p.shape, generator=g, device=DEV))
    state = {k: v.detach().cpu().numpy() for k, v in task.state_dict().items()}
    batch = synthetic.make_batch(B, vn, vc, kn, kc, 256, 128, torch.device(DEV), seed=1234)
    res = task(batch, return_metrics=True)
    res["loss"].backward()
    torch.cuda.synchronize()
    store = task.two_tower_model.embedding_store
    store = store() if callable(store) else store
    plan, grad_rows = store.sparse_grad
    U = int(plan.n_unique.item())
    got_rows, got_grad = plan.unique_rows[:U].cpu().numpy().astype(np.int64), grad_rows[:U].cpu().numpy()
    with torch.no_grad():
        ne, ce = task.two_tower_model(batch["notice"], batch["company"])

    b = {"notice_ids": batch["notice"]["kjt"].values().cpu().numpy().reshape(B, len(kn)),
         "company_ids": batch["company"]["kjt"].values().cpu().numpy().reshape(B, len(kc)),
         "notice_dense": batch["notice"]["dense"].cpu().numpy(), "company_dense": batch["company"]["dense"].cpu().numpy()}
    # B a multiple of 64 (the bench shape): the one-launch first-block backward forms the projection gradients as
    # W[:, :h0]^T . (d_pre^T . dense); other batch sizes take the separate GEMMs (d_proj^T . dense) -- the oracle follows
    ref = O.task_step(state, b, kn, kc, vn, vc, T, True, dtype=np.float64, rounding="bf16", table_grads="none", keep_sim=False,
                      proj_grad="factored" if (B % 64 == 0 and (hidden[1] // 64) * hidden[0] <= 256) else "direct",
                      score_rounding=score_rounding)
    bd = {"fp8": FP8_STEP_BOUNDS, "fp8_s": FP8S_STEP_BOUNDS}.get(score_rounding, BF16_STEP_BOUNDS)
    # measure everything first (the report is printed with -s and quoted in DESIGN.md section 4), then assert
    report = {"loss": abs(res["loss"].item() - ref["loss"]) / ref["loss"]}
    for name, got, want in (("notice_emb", ne, ref["notice_emb"]), ("company_emb", ce, ref["company_emb"])):
        got = got.cpu().numpy()
        report[name] = (_rel(got, want), float(np.abs(got - want).max()))
    for k in ("positive_similarity_mean", "negative_similarity_mean", "similarity_gap", "accuracy"):
        report[k] = abs(res[k].item() - float(ref[k]))
    for n_, p in task.named_parameters():
        if "categorical_embedder" not in n_:
            report[n_] = _rel(p.grad.cpu().numpy(), ref["grads"][n_])
    # sparse row gradients over the fused row space (notice keys, then company keys): same row set, bounded values
    offs_n = np.cumsum([0] + list(vn[:-1]))
    offs_c = sum(vn) + np.cumsum([0] + list(vc[:-1]))
    rn, gn = O.embed_grad_sparse(ref["d_concat_notice"], ref["ids_notice"], offs_n, 32)
    rc, gc = O.embed_grad_sparse(ref["d_concat_company"], ref["ids_company"], offs_c, 32)
    rows_equal = np.array_equal(got_rows, np.concatenate([rn, rc]))
    report["row_grads"] = _rel(got_grad, np.concatenate([gn, gc])) if rows_equal else float("inf")
    print(f"\n[{score_rounding or score_dtype} step vs rounded oracle]", json.dumps({k: (v if not isinstance(v, tuple) else list(v)) for k, v in report.items()}))
    # ---- the same step against the reference's arithmetic (no operand rounding anywhere): the benchmarked shape and configs[4]'s
    # score arithmetic at the largest batch the f64 oracle holds
    ref_metric = {k: float(ref[k]) for k in ("positive_similarity_mean", "negative_similarity_mean", "similarity_gap")}
    if (rows_per_tower and B == 8192) or (score_rounding == "fp8" and B == 2048):
        del ref, rn, gn, rc, gc
        refu = O.task_step(state, b, kn, kc, vn, vc, T, True, dtype=np.float64, rounding=None, table_grads="none", keep_sim=False)
        bu = FP8_VS_REFERENCE_BOUNDS if score_rounding else BF16_VS_REFERENCE_BOUNDS
        ru = {"loss": abs(res["loss"].item() - refu["loss"]) / refu["loss"]}
        for name, got, want in (("notice_emb", ne, refu["notice_emb"]), ("company_emb", ce, refu["company_emb"])):
            got = got.cpu().numpy()
            ru[name] = (_rel(got, want), float(np.abs(got - want).max()))
        for k in ("positive_similarity_mean", "negative_similarity_mean", "similarity_gap", "accuracy"):
            ru[k] = abs(res[k].item() - float(refu[k]))
        for n_, p in task.named_parameters():
            if "categorical_embedder" not in n_:
                ru[n_] = _rel(p.grad.cpu().numpy(), refu["grads"][n_])
        rnu, gnu = O.embed_grad_sparse(refu["d_concat_notice"], refu["ids_notice"], offs_n, 32)
        rcu, gcu = O.embed_grad_sparse(refu["d_concat_company"], refu["ids_company"], offs_c, 32)
        ru["row_grads"] = _rel(got_grad, np.concatenate([gnu, gcu])) if np.array_equal(got_rows, np.concatenate([rnu, rcu])) else float("inf")
        print(f"[{score_rounding or score_dtype} step vs REFERENCE arithmetic (unrounded f64 oracle)]",
              json.dumps({k: (v if not isinstance(v, tuple) else list(v)) for k, v in ru.items()}))
        assert ru["loss"] <= bu["loss_rtol"], ru
        for name in ("notice_emb", "company_emb"):
            assert ru[name][0] <= bu["emb_norm"] and ru[name][1] <= bu["emb_maxabs"], (name, ru[name])
        for k in ("positive_similarity_mean", "negative_similarity_mean", "similarity_gap"):
            assert ru[k] <= bu["metric_atol"], (k, ru[k])
        assert ru["accuracy"] <= 4.0 / B, ru["accuracy"]
        for n_, p in task.named_parameters():
            if "categorical_embedder" not in n_:
                assert ru[n_] <= (bu["dense_grad_matrix_norm"] if p.ndim > 1 else bu["dense_grad_vector_norm"]), (n_, ru[n_])
        assert ru["row_grads"] <= bu["row_grad_norm"], ru["row_grads"]
    assert report["loss"] <= bd["loss_rtol"], report
    for name in ("notice_emb", "company_emb"):
        assert report[name][0] <= bd["emb_norm"] and report[name][1] <= bd["emb_maxabs"], (name, report[name])
    for k in ("positive_similarity_mean", "negative_similarity_mean", "similarity_gap"):
        assert report[k] <= bd["metric_atol"] + 1e-4 * abs(ref_metric[k]), (k, report[k])
    assert report["accuracy"] <= 2.0 / B
    for n_, p in task.named_parameters():
        if "categorical_embedder" not in n_:
            assert report[n_] <= (bd["dense_grad_matrix_norm"] if p.ndim > 1 else bd["dense_grad_vector_norm"]), (n_, report[n_])
    assert rows_equal                                                               # touched-row set: bit-exact
    assert report["row_grads"] <= bd["row_grad_norm"], report["row_grads"]


@pytest.mark.parametrize("B,D,T", [(300, 64, 1.0), (129, 16, 0.5), (64, 6, 0.25), (1000, 128, 1.0), (257, 200, 2.0), (2048, 64, 1.0), (70, 32, 1.0),
                                   (8192, 64, 1.0), (33, 64, 1.0), (1, 8, 1.0), (4100, 64, 0.5)])
@pytest.mark.parametrize("prescale", [True, False])
def test_score_sym_forward(tt, B, D, T, prescale):
    """tt_score_fwd_sym_bf16 (every tile of S computed ONCE for both softmax directions) against (a) the two-direction kernel
    on the same packed operands -- exp-sums 1e-5, the diagonal and the top-1 flags bit for bit -- and (b) the f64 oracle fed
    the same bf16-rounded operands: loss 2e-6, metrics; twice the same bits."""
    from jodalrob_twotower_amd import ops
    rng = np.random.default_rng(B * 7 + D)
    n = rng.standard_normal((B, D)).astype(np.float32)
    c = (0.6 * n + rng.standard_normal((B, D))).astype(np.float32)              # correlated: some positives ARE the row maximum
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    if B > 8:
        c[5] = c[3]                                                              # exact ties: duplicated company rows
        n[7] = n[2]
    inv_t = 1.0 / T
    sn = ops.score_unit_scale(inv_t) if prescale else 1.0
    tn, tc = torch.from_numpy(n).to(DEV), torch.from_numpy(c).to(DEV)
    Np, Cp = ops.score_pack2_bf16(tn, tc, sn, 1.0)
    rs, cs, dg, rk, (ir, ic), out8, loss = ops.score_fwd_sym(Np, Cp, B, D, inv_t, abs(inv_t), sn, True)
    old = ops.score_fwd_bf16(Np, Cp, B, D, inv_t, abs(inv_t), True, False, sn, with_inv=True)
    o_out8, o_loss = ops.score_loss_finish(B, abs(inv_t), *old[:6])
    torch.testing.assert_close(rs, old[0], rtol=1e-5, atol=0)
    torch.testing.assert_close(cs, old[1], rtol=1e-5, atol=0)
    assert torch.equal(dg, old[2])                                               # the positives: the same MFMA results
    assert torch.equal(rk, old[3])                                               # top-1 flags incl. the tie rules
    torch.testing.assert_close(ir, old[6][0], rtol=1e-5, atol=0)
    torch.testing.assert_close(ic, old[6][1], rtol=1e-5, atol=0)
    np.testing.assert_allclose(loss.item(), o_loss.item(), rtol=2e-6, atol=2e-7)
    for k in (1, 2):
        assert out8[k].item() == pytest.approx(o_out8[k].item(), rel=1e-6, abs=1e-7)
    if B > 1:
        np.testing.assert_allclose(out8[3].item(), o_out8[3].item(), rtol=2e-3, atol=2e-6)      # sum of scores: (sum n).(sum c) vs 67 M adds
    # the oracle on the same rounded operands
    nb = (torch.from_numpy(n) * np.float32(sn)).bfloat16().float().numpy().astype(np.float64) / float(np.float32(sn))
    cb = torch.from_numpy(c).bfloat16().float().numpy().astype(np.float64)
    ref_loss, met, S, lse = O.score_ce_fwd(nb, cb, T)
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(out8[2].item(), met["positive_similarity_mean"], rtol=1e-4, atol=1e-6)
    if B > 1:
        np.testing.assert_allclose(out8[3].item(), met["negative_similarity_mean"], rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(rs.cpu().numpy(), np.exp(S - abs(inv_t)).sum(1), rtol=2e-5)
    np.testing.assert_allclose(cs.cpu().numpy(), np.exp(S - abs(inv_t)).sum(0), rtol=2e-5)
    # bitwise reproducible
    again = ops.score_fwd_sym(Np, Cp, B, D, inv_t, abs(inv_t), sn, True)
    assert torch.equal(again[0], rs) and torch.equal(again[1], cs) and again[6].item() == loss.item()
    assert torch.equal(again[5].nan_to_num(nan=-7.0), out8.nan_to_num(nan=-7.0))            # (B = 1: the off-diagonal mean is nan, as torch)


@pytest.mark.parametrize("B,D,T", [(300, 64, 1.0), (1000, 128, 1.0), (257, 200, 2.0), (2048, 64, 0.5), (70, 256, 1.0), (513, 256, 1.0)])
def test_score_backward_large_batch_form(tt, ctx_option, B, D, T):
    """The workgroup-staged backward kernel (waves own rows a, every b tile staged once through LDS: the form used from
    32768 rows up) against the b-split form on the same operands -- the two sum over b in different orders: 2e-5 norm-wise
    -- and against the f64 oracle with the kernels' rounding (operands and softmax weights in bf16): 3e-4 norm-wise."""
    from jodalrob_twotower_amd import ops
    rng = np.random.default_rng(B + D)
    n = rng.standard_normal((B, D)).astype(np.float32)
    c = (0.5 * n + rng.standard_normal((B, D))).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    inv_t = 1.0 / T
    sn = ops.score_unit_scale(inv_t)
    tn, tc = torch.from_numpy(n).to(DEV), torch.from_numpy(c).to(DEV)
    Np, Cp = ops.score_pack2_bf16(tn, tc, sn, 1.0)
    rs, cs, dg, rk, inv, out8, loss = ops.score_fwd_sym(Np, Cp, B, D, inv_t, abs(inv_t), sn, True)
    one = torch.ones(1, device=DEV)
    scale = inv_t / (2.0 * B)
    ctx_option(_L.TT_OPT_SCORE_BWD_ROWS_MIN, 1000000000, 32768)
    dN0, dC0 = ops.score_bwd_bf16(Np, Cp, B, D, inv_t, abs(inv_t), rs, cs, one, scale, sn, inv)
    ctx_option(_L.TT_OPT_SCORE_BWD_ROWS_MIN, 1, 32768)
    dN1, dC1 = ops.score_bwd_bf16(Np, Cp, B, D, inv_t, abs(inv_t), rs, cs, one, scale, sn, inv)
    dN2, dC2 = ops.score_bwd_bf16(Np, Cp, B, D, inv_t, abs(inv_t), rs, cs, one, scale, sn, None)     # reciprocals taken in the kernel
    for a, b in ((dN1, dN0), (dC1, dC0), (dN2, dN0), (dC2, dC0)):
        assert _rel(a.cpu().numpy(), b.cpu().numpy()) <= 2e-5
    nb, cb = O.score_operands_bf16(n.astype(np.float64), c.astype(np.float64), T)
    ref_loss, met, S, lse = O.score_ce_fwd(nb, cb, T)
    rN, rC = O.score_ce_bwd(nb, cb, S, lse, T, q=O.q_bf16)
    assert _rel(dN1.cpu().numpy(), rN) <= 3e-4 and _rel(dC1.cpu().numpy(), rC) <= 3e-4
    assert torch.equal(ops.score_bwd_bf16(Np, Cp, B, D, inv_t, abs(inv_t), rs, cs, one, scale, sn, inv)[0], dN1)      # reproducible


def _unpack_fp8_rows(buf, R, D):
    """[R, Dp] float32 view of the fp8 rows image (tt_score_bf16.h): [tile][k64][part][half][row][16 bytes]"""
    Rp, Dp = (R + 63) // 64 * 64, (64 if D <= 64 else (128 if D <= 128 else 256))
    raw = buf[:Rp * Dp].view(torch.float8_e4m3fn).float().cpu().numpy().reshape(Rp // 32, Dp // 64, 2, 2, 32, 16)
    # axes: tile, s, part, half, row, byte  ->  row = 32 tile + row ; col = 64 s + 32 half + 16 part + byte
    out = raw.transpose(0, 4, 1, 3, 2, 5).reshape(Rp, Dp)
    return out[:R]


def _unpack_fp8_frag(buf, R, D):
    """[R, Dp] float32 view of the fp8 fragment image (tt_score_bf16.h): [pair P][32-column block d][part][half][column c][16 bytes],
    byte j of part p = (row 64 P + 32 p + rowmap(j, half), column 32 d + c)"""
    Rp, Dp = (R + 63) // 64 * 64, (64 if D <= 64 else (128 if D <= 128 else 256))
    raw = buf[3 * Rp * Dp:4 * Rp * Dp].view(torch.float8_e4m3fn).float().cpu().numpy().reshape(Rp // 64, Dp // 32, 2, 2, 32, 4, 4)
    # axes: P, d, part, half, c, j >> 2, j & 3  ->  row = 64 P + 32 part + 8 (j >> 2) + 4 half + (j & 3) ; col = 32 d + c
    out = raw.transpose(0, 2, 5, 3, 6, 1, 4).reshape(Rp, Dp)
    return out[:R]


def _unpack_bf16_frag(buf, R, D):
    """[R, Dp] float32 view of the bf16 fragment image of the fp8 packing: [tile][s][h][d][8], element j of a chunk = row
    32 t + 16 s + 8 (j >> 2) + 4 h + (j & 3), column d"""
    Rp, Dp = (R + 63) // 64 * 64, (64 if D <= 64 else (128 if D <= 128 else 256))
    raw = buf[Rp * Dp:3 * Rp * Dp].view(torch.bfloat16).float().cpu().numpy().reshape(Rp // 32, 2, 2, Dp, 2, 4)
    # axes: t, s, h, d, j >> 2, j & 3  ->  row = 32 t + 16 s + 8 (j >> 2) + 4 h + (j & 3)
    return raw.transpose(0, 1, 4, 2, 5, 3).reshape(Rp, Dp)[:R]


@pytest.mark.parametrize("B,D", [(4096, 256), (4100, 200), (5000, 64), (8192, 128)])
def test_fp8_pack_large_batches(tt, B, D):
    """tt_score_pack2_fp8 from 4096 rows on (pack_fp8_tile_kernel: 64 rows staged in LDS, ONE read of the operand for the three
    images): every image == torch's conversion of the same scaled values, element for element -- rows image and fp8 fragment
    image against float8_e4m3fn of clamp(64 s x), bf16 fragment image against bfloat16 of s x; ragged row count, a width that is
    not its padded width (zero columns), two operands with different scales in one launch."""
    from jodalrob_twotower_amd import ops
    g = torch.Generator().manual_seed(B + D)
    x0 = torch.randn(B, D, generator=g) * 0.3
    x1 = torch.randn(B, D, generator=g) * 0.1
    s0, s1 = 1.4375, 1.0
    p0, p1 = ops.score_pack2_fp8(x0.to(DEV), x1.to(DEV), s0, s1)
    for x, sc, p in ((x0, s0, p0), (x1, s1, p1)):
        w8 = (x * np.float32(sc) * 64.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float().numpy()
        w16 = (x * np.float32(sc)).to(torch.bfloat16).float().numpy()
        Dp = 64 if D <= 64 else (128 if D <= 128 else 256)
        r8, f8, f16 = _unpack_fp8_rows(p, B, D), _unpack_fp8_frag(p, B, D), _unpack_bf16_frag(p, B, D)
        assert np.array_equal(r8[:, :D], w8) and np.array_equal(f8[:, :D], w8) and np.array_equal(f16[:, :D], w16)
        assert not r8[:, D:Dp].any() and not f8[:, D:Dp].any() and not f16[:, D:Dp].any()


@pytest.mark.parametrize("fp8_grad", [1, 0])
@pytest.mark.parametrize("B,D,T", [(300, 64, 1.0), (513, 256, 1.0), (1000, 128, 0.5), (70, 200, 2.0), (2048, 256, 1.0), (1000, 256, 0.05)])
def test_score_fp8_vs_rounded_oracle(tt, ctx_option, B, D, T, fp8_grad):
    """score_dtype="fp8" (BASELINE configs[4]: e4m3 operands on the block-scaled MFMA): (a) the packed fp8 images == torch's
    float8_e4m3fn conversion of 64 * scale * x, element for element (rows image and fragment image); (b) loss / exp-sums /
    metrics against the f64 oracle fed the SAME e4m3-rounded operands: loss 2e-6, sums 2e-5; (c) gradients against the oracle
    with the kernels' rounding -- fp8_grad 1 (default, TT_OPT_FP8_GRAD): e4m3 operands and block-scaled e4m3 softmax weights
    for the gradient products too, the diagonal's weight exact on the bf16 row (oracle: q_block_e4m3); fp8_grad 0: bf16
    softmax weights, bf16 product operands -- 1e-4 norm-wise in both (measured 4e-7 .. 1.4e-5; a weight whose f32 value sits
    within rounding of an e4m3 tie may round the other way than the f64 oracle's: one part in 16 of ONE of a row's B weights); (d) against the
    unrounded f64 oracle: what e4m3 operands cost -- loss within 2e-3, gradients within 8e-2.  T = 0.05 is the peaked case:
    weights from 1 down to e^-40 within a row, the case the per-block scales exist for."""
    ctx_option(_L.TT_OPT_FP8_GRAD, fp8_grad, 1)
    from jodalrob_twotower_amd import ops
    from jodalrob_twotower_amd.two_tower_train_task import _ScoreCEFn
    rng = np.random.default_rng(B * 3 + D)
    n = rng.standard_normal((B, D)).astype(np.float32)
    c = (0.5 * n + rng.standard_normal((B, D))).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    inv_t = 1.0 / T
    sn = ops.score_unit_scale(inv_t)
    tn, tc = torch.from_numpy(n).to(DEV), torch.from_numpy(c).to(DEV)
    Np, Cp = ops.score_pack2_fp8(tn, tc, sn, 1.0)
    want_n = (torch.from_numpy(n) * np.float32(sn) * 64.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float().numpy()   # (saturating pack)
    want_c = (torch.from_numpy(c) * 64.0).to(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(_unpack_fp8_rows(Np, B, D)[:, :D], want_n) and np.array_equal(_unpack_fp8_rows(Cp, B, D)[:, :D], want_c)
    assert np.array_equal(_unpack_fp8_frag(Np, B, D)[:, :D], want_n) and np.array_equal(_unpack_fp8_frag(Cp, B, D)[:, :D], want_c)
    n8, c8 = O.score_operands_fp8(n.astype(np.float64), c.astype(np.float64), T)
    np.testing.assert_allclose(n8 * float(np.float32(sn)) * 64.0, want_n.astype(np.float64), rtol=1e-12, atol=0)     # the oracle's operands ARE the packed ones
    ref_loss, met, S, lse = O.score_ce_fwd(n8, c8, T)
    rs, cs, dg, rk, inv, out8, loss = ops.score_fwd_sym(Np, Cp, B, D, inv_t, abs(inv_t), sn, True, fp8=True)
    k = 1.0 if T >= 0.5 else 10.0                          # 1/T = 20 multiplies the f32 accumulation error of S in the exponent
    np.testing.assert_allclose(loss.item(), ref_loss, rtol=2e-6 * k)
    np.testing.assert_allclose(rs.cpu().numpy(), np.exp(S - abs(inv_t)).sum(1), rtol=2e-5 * k)
    np.testing.assert_allclose(cs.cpu().numpy(), np.exp(S - abs(inv_t)).sum(0), rtol=2e-5 * k)
    np.testing.assert_allclose(dg.cpu().numpy(), np.diagonal(S), rtol=1e-4, atol=2e-6 * k)      # f32 accumulation of D products
    assert abs(out8[1].item() - float(met["accuracy"])) <= 2.0 / B
    np.testing.assert_allclose(out8[2].item(), met["positive_similarity_mean"], rtol=1e-4, atol=1e-6 * k)
    np.testing.assert_allclose(out8[3].item(), met["negative_similarity_mean"], rtol=2e-3, atol=2e-6 * k)
    # gradients through the autograd node the task uses
    a, b = tn.clone().requires_grad_(True), tc.clone().requires_grad_(True)
    l2, _, _ = _ScoreCEFn.apply(a, b, inv_t, "fp8", False, False)
    assert l2.item() == loss.item()
    l2.backward()
    nb, cb = O.score_operands_bf16(n.astype(np.float64), c.astype(np.float64), T)
    rN, rC = O.score_ce_bwd(n8, c8, S, lse, T, q=O.q_bf16, prod_operands=(nb, cb), block_fp8=bool(fp8_grad))
    print(f"fp8 B={B} D={D} T={T} fp8_grad={fp8_grad}: dN {_rel(a.grad.cpu().numpy(), rN):.2e} dC {_rel(b.grad.cpu().numpy(), rC):.2e}")
    assert _rel(a.grad.cpu().numpy(), rN) <= 1e-4 * k and _rel(b.grad.cpu().numpy(), rC) <= 1e-4 * k
    # what the format costs, against the unrounded oracle
    f_loss, _, fS, flse = O.score_ce_fwd(n.astype(np.float64), c.astype(np.float64), T)
    fN, fC = O.score_ce_bwd(n.astype(np.float64), c.astype(np.float64), fS, flse, T)
    print(f"    vs unrounded: loss {abs(loss.item() - f_loss) / f_loss:.2e} dN {_rel(a.grad.cpu().numpy(), fN):.2e} dC {_rel(b.grad.cpu().numpy(), fC):.2e}")
    np.testing.assert_allclose(loss.item(), f_loss, rtol=2e-3 if T >= 0.5 else 2e-2)
    fmt = 8e-2 if T >= 0.5 else 2e-1                      # (1/T multiplies the operands' rounding error in the exponent)
    assert _rel(a.grad.cpu().numpy(), fN) <= fmt and _rel(b.grad.cpu().numpy(), fC) <= fmt


# ------------------------------------------------------------------ BASELINE configs[2] / [3] / [4] at their own sizes
# measured (round 4): T = 1: loss 5.818 vs 5.813, accuracy 0.093 vs 0.107, recall@10 0.525 vs 0.547; T = 0.05: loss 2.729 vs 2.748, accuracy
# 0.401 vs 0.391, recall@10 0.797 vs 0.797 (fp8_grad 1 vs 0; both from loss ln(1024) + ... = 6.93 / 7.66 at step 0)
FP8_GRAD_TRAJECTORY_BOUNDS = {"loss_rel": 0.02, "accuracy_abs": 0.03, "recall_abs": 0.04}


@pytest.mark.parametrize("T", [1.0, 0.05])
def test_fp8_grad_default_trains_like_bf16_gradient_products(tt, schema_real, tmp_path, ctx_option, T):
    """TT_OPT_FP8_GRAD defaults to 1 -- e4m3 softmax weights and e4m3 operands in the backward's gradient
    products -- and the single-step bounds (FP8_STEP_BOUNDS) say nothing about whether that default still TRAINS like the bf16
    gradient products (TT_OPT_FP8_GRAD 0).  200 captured steps of configs[4]'s arithmetic (final_embedding_dim 256, fp8 score
    operands, row-sparse FusedAdam) on a fixed pool of 8 batches (B = 1024: the model memorises them -- loss falls, in-batch
    accuracy and recall@10 rise), from the same seed, with fp8_grad 1 and 0, at T = 1 and T = 0.05 (1/T = 20 in the exponent: the
    regime where the e4m3 rounding of S's operands costs most): the two trajectories must end at the same loss, accuracy and
    recall@10 within the stated bounds, and both must have learned.  (The reference holds no fp8 fixture: fp8 parity is pinned by
    the rounded oracle per step and by this A/B per trajectory, not by reference outputs.)"""
    from jodalrob_twotower_amd import synthetic
    from jodalrob_twotower_amd.graph import GraphedTrainStep
    from jodalrob_twotower_amd.optim import FusedAdam
    kn, kc = schema_real["notice"]["categorical"][:6], schema_real["company"]["categorical"][:3]
    vn, vc = [3000, 500, 12, 12, 40, 7], [2000, 30, 9]
    meta = synthetic.write_metadata(tmp_path / "m.csv", {"notice": dict(zip(kn, vn)), "company": dict(zip(kc, vc))})
    B, steps = 1024, 200
    pool = [synthetic.make_batch(B, vn, vc, kn, kc, 64, 32, torch.device(DEV), seed=4000 + i) for i in range(8)]
    ends = {}
    for fp8_grad in (1, 0):
        ctx_option(_L.TT_OPT_FP8_GRAD, fp8_grad, 1)
        torch.manual_seed(99)
        task = tt.create_two_tower_train_task(kn, kc, metadata_path=str(meta), categorical_embedding_dim=32, notice_dense_input_dim=64,
                                              company_dense_input_dim=32, tower_hidden_dims=[128, 64], final_embedding_dim=256, dropout_rate=0.0,
                                              temperature=T, device=DEV, embedding_grad="sparse", score_dtype="fp8", mlp_dtype="bf16")
        task.train()
        task._pair_check_done = True
        opt = FusedAdam.for_task(task, lr=3e-3, weight_decay=1e-5)
        gs = GraphedTrainStep(task, opt, pool[0], warmup=1, accumulate_metrics=True)
        first = None
        for i in range(steps):
            r = gs.step(pool[i % len(pool)])
            if i == 0:
                first = float(r["loss"].item())
            if i == steps - 2 * len(pool):
                gs.metric_sums.zero_()                       # means over the last two passes through the pool
        sums = gs.metric_sums.cpu()
        gs.close()
        task.eval()
        rec = float(np.mean([float((task.diagonal_ranks(b) < 10).float().mean().item()) for b in pool]))
        ends[fp8_grad] = {"first_loss": first, "loss": float(sums[0]) / (2 * len(pool)), "accuracy": float(sums[1]) / (2 * len(pool)), "recall@10": rec}
        del task, opt, gs
        torch.cuda.empty_cache()
    print(f"\n[fp8_grad trajectory A/B, T = {T}]", json.dumps(ends))
    a, b = ends[1], ends[0]
    bd = FP8_GRAD_TRAJECTORY_BOUNDS
    assert a["first_loss"] == pytest.approx(b["first_loss"], rel=1e-6)                     # same start
    for e in (a, b):
        assert e["loss"] < e["first_loss"] - 0.5 and e["recall@10"] > 10.0 / B * 5, e      # both learned (recall@10 of a random model: 0.01)
    assert abs(a["loss"] - b["loss"]) <= bd["loss_rel"] * abs(b["loss"]), (a, b)
    assert abs(a["accuracy"] - b["accuracy"]) <= bd["accuracy_abs"], (a, b)
    assert abs(a["recall@10"] - b["recall@10"]) <= bd["recall_abs"], (a, b)


def test_configs4_whole_step_full_size(tt, schema_real, tmp_path):
    """BASELINE configs[4] as ONE WHOLE STEP at its own size -- batch 65536, final_embedding_dim 256, score_dtype "fp8", row-sparse
    table gradients, FusedAdam (fused sparse Adam on the looked-up rows), 1 M + 1 M-row tables -- eagerly and replayed from the
    captured graph: lookup of 2.5 M slots, the global radix duplicate-row plan (the keyed plan stops at B = 8192), segmented
    reduction, both towers, fp8 score forward / backward, fused Adam and the graph's copy hand-over (tt_copy_multi).  The oracle
    cannot hold B = 65536, so the step is checked through size-independent properties: the loss of a fresh initialisation is
    ln(B); two passes from the same state give the same bits (loss, touched rows, row gradients); the touched-row set IS the
    set of clamped ids of the batch (numpy on the ids: bit-exact); one optimiser step moves exactly those rows and every dense
    parameter; and the replayed steps reproduce the eager ones bit for bit over three different batches.  (The same step
    against the rounded f64 oracle, at B = 2048 / 1000: test_bf16_step_vs_rounded_oracle[...-256-fp8].)"""
    from jodalrob_twotower_amd import synthetic
    from jodalrob_twotower_amd.graph import GraphedTrainStep
    from jodalrob_twotower_amd.optim import FusedAdam
    B = 65536
    dev = torch.device(DEV)
    torch.manual_seed(17)
    task, (kn, kc, vn, vc) = _config_task(tt, schema_real, tmp_path, 1_000_000, 1_000_000, False, final_embedding_dim=256, score_dtype="fp8")
    state0 = {k: v.detach().clone() for k, v in task.state_dict().items()}
    batches = [synthetic.make_batch(B, vn, vc, kn, kc, 256, 128, dev, seed=700 + i) for i in range(3)]
    store = task.two_tower_model.embedding_store
    store = store() if callable(store) else store
    opt = FusedAdam.for_task(task, lr=1e-3, weight_decay=1e-5)
    runs = []
    for rep in range(2):
        opt.zero_grad()
        r = task(batches[0], return_metrics=True)
        r["loss"].backward()
        plan, grad_rows = store.sparse_grad
        U = int(plan.n_unique.item())
        runs.append((r["loss"].item(), plan.unique_rows[:U].clone(), grad_rows[:U].clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    assert abs(runs[0][0] - np.log(B)) < 0.25, runs[0][0]                 # ln(65536) = 11.09 at a random initialisation
    assert bool(torch.isfinite(runs[0][2]).all())
    # the touched rows are the batch's ids, clamped and offset into the fused row space (cat_embed.py:114-117): bit-exact
    want = []
    off = 0
    for side, keys, vocab in (("notice", kn, vn), ("company", kc, vc)):
        ids = batches[0][side]["kjt"].values().view(B, len(keys))
        offs = off + torch.tensor(np.concatenate([[0], np.cumsum(vocab)[:-1]]), device=dev)
        hi = torch.tensor(vocab, device=dev) - 1
        want.append((torch.minimum(ids.clamp(min=0), hi) + offs).reshape(-1))
        off += sum(vocab)
    want_rows = torch.unique(torch.cat(want)).int()
    assert torch.equal(runs[0][1].int(), want_rows)
    # back to the initial state (the two passes above moved the BatchNorm running statistics twice), then step 1 for real
    task.load_state_dict(state0)
    opt = FusedAdam.for_task(task, lr=1e-3, weight_decay=1e-5)
    opt.zero_grad()
    r = task(batches[0], return_metrics=True)
    r["loss"].backward()
    assert r["loss"].item() == runs[0][0]
    before_tab = store.weight.detach().clone()
    before_dense = {n: p.detach().clone() for n, p in task.named_parameters() if "categorical_embedder" not in n}
    opt.step()
    torch.cuda.synchronize()
    changed = (store.weight.detach() != before_tab).any(dim=1).nonzero().flatten().int()
    assert torch.equal(changed, want_rows)                                  # exactly the touched rows moved
    del before_tab
    for n, p in task.named_parameters():
        if "categorical_embedder" not in n:
            assert not torch.equal(p.detach(), before_dense[n]), n
    # eager steps 2, 3 on new batches, then the same three steps replayed from a captured graph on a task reset to the same state
    losses_e = [runs[0][0]]
    for b in batches[1:]:
        opt.zero_grad()
        r = task(b, return_metrics=True)
        r["loss"].backward()
        opt.step()
        losses_e.append(r["loss"].item())
    torch.cuda.synchronize()
    final_e = {k: v.detach().clone() for k, v in task.state_dict().items()}
    del opt, runs, r
    task.load_state_dict(state0)
    opt = FusedAdam.for_task(task, lr=1e-3, weight_decay=1e-5)
    gs = GraphedTrainStep(task, opt, batches[0], warmup=1, preserve_state=False)     # its one eager warm-up step IS step 1
    losses_g = [None] + [gs.step(b)["loss"].item() for b in batches[1:]]
    torch.cuda.synchronize()
    assert losses_g[1:] == losses_e[1:], (losses_g, losses_e)
    for k, v in final_e.items():
        assert torch.equal(task.state_dict()[k], v), k
    gs.close()
    del gs, opt, task, final_e, state0
    torch.cuda.empty_cache()


def _config_task(tt, schema_real, tmp_path, rows_n, rows_c, sharded, **kw):
    from jodalrob_twotower_amd import synthetic
    kn, kc = schema_real["notice"]["categorical"], schema_real["company"]["categorical"]
    vn = synthetic.scale_vocabs(schema_real["notice"]["vocab_sizes"], rows_n)
    vc = synthetic.scale_vocabs(schema_real["company"]["vocab_sizes"], rows_c)
    meta = synthetic.write_metadata(tmp_path / "m.csv", {"notice": dict(zip(kn, vn)), "company": dict(zip(kc, vc))})
    common = dict(metadata_path=str(meta), categorical_embedding_dim=32, notice_dense_input_dim=256, company_dense_input_dim=128,
                  tower_hidden_dims=[128, 64], dropout_rate=0.0, temperature=1.0, device=DEV, embedding_grad="sparse", mlp_dtype="bf16", **kw)
    if sharded:
        from jodalrob_twotower_amd.distributed import create_distributed_train_task
        task = create_distributed_train_task(kn, kc, **common)
    else:
        task = tt.create_two_tower_train_task(kn, kc, **common)
    task.train()
    task._pair_check_done = True
    return task, (kn, kc, vn, vc)


@pytest.mark.parametrize("zipf", [None, 1.2])
def test_configs2_3_sharded_100m_rows(tt, schema_real, tmp_path, zipf):
    """BASELINE configs[2] (100 M notice + 10 M company rows, row-wise sharded tables behind the fixed-capacity exchange, RCCL)
    and configs[3] (the same with Zipf(1.2) ids) at their OWN table sizes on the one GPU of the box (world 1: every kernel and
    collective of the sharded step runs, RCCL moves the data to itself; 42 GB with the Adam moments), through
    size-independent properties: the ids really range over the 100 M rows, no bucket overflows, the loss of a random
    initialisation is ln(B), two passes give the same bits (loss, touched rows, row gradients), Zipf ids touch far fewer
    distinct rows than slots, and one optimiser step moves exactly the touched rows of the shard.  (Equality with the
    unsharded single-GPU step: test_configs2_sharded_equals_unsharded_10m_rows, at sizes where both tasks fit side by side.)"""
    import os
    import torch.distributed as dist
    from jodalrob_twotower_amd import synthetic
    from jodalrob_twotower_amd.optim import FusedAdam
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29537")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        B = 8192
        torch.manual_seed(3)
        dtask, (kn, kc, vn, vc) = _config_task(tt, schema_real, tmp_path, 100_000_000, 10_000_000, True, final_embedding_dim=64, score_dtype="bf16")
        batch = synthetic.make_batch(B, vn, vc, kn, kc, 256, 128, torch.device(DEV), seed=41, zipf_alpha=zipf)
        assert int(batch["notice"]["kjt"].values().max()) > 10_000_000        # ids over the whole 100 M-row key
        opt = FusedAdam.for_task(dtask, lr=1e-3)
        runs = []
        for rep in range(2):
            opt.zero_grad()
            r = dtask(batch, return_metrics=True)
            r["loss"].backward()
            dtask.exchange.check_overflow()
            plan, grad_rows = dtask.sharded_store.sparse_grad
            U = int(plan.n_unique.item())
            runs.append((r["loss"].item(), plan.unique_rows[:U].clone(), grad_rows[:U].clone()))
        assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
        assert 8.0 < runs[0][0] < 10.0                            # ln(8192) = 9.01 at random init
        U = runs[0][1].numel()
        if zipf is None:
            assert U > 0.7 * B * 21                               # 17 of the 38 keys are binary; the others rarely repeat in 100 M rows
        else:
            assert U < 0.6 * B * 38                               # hot ids: far fewer distinct rows than slots
        before = dtask.embedding_shard.detach().clone()
        opt.step()
        torch.cuda.synchronize()
        changed = (dtask.embedding_shard.detach() != before).any(dim=1).nonzero().flatten().int()
        rows = runs[1][1]
        assert torch.equal(changed, rows[rows < dtask.sharded_store.local_rows].int())      # exactly the touched rows moved
        del before, dtask, opt
        torch.cuda.empty_cache()
    finally:
        dist.destroy_process_group()


def test_configs2_sharded_equals_unsharded_10m_rows(tt, schema_real, tmp_path):
    """the sharded step (world 1, padded exchange) == the single-GPU step on the same 10 M + 1 M-row tables and batch:
    loss bit for bit, touched rows and their gradients bit for bit (the sizes at which both tasks and a table copy fit
    comfortably; configs[2]'s own 100 M + 10 M rows are covered by test_configs2_3_sharded_100m_rows)."""
    import os
    import torch.distributed as dist
    from jodalrob_twotower_amd import synthetic
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29538")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        B = 8192
        torch.manual_seed(5)
        stask, (kn, kc, vn, vc) = _config_task(tt, schema_real, tmp_path, 10_000_000, 1_000_000, False, final_embedding_dim=64, score_dtype="bf16")
        dtask, _ = _config_task(tt, schema_real, tmp_path, 10_000_000, 1_000_000, True, final_embedding_dim=64, score_dtype="bf16")
        dtask.load_full_state_dict(stask.state_dict())
        batch = synthetic.make_batch(B, vn, vc, kn, kc, 256, 128, torch.device(DEV), seed=43, zipf_alpha=1.2)
        rs = stask(batch, return_metrics=True); rs["loss"].backward()
        rd = dtask(batch, return_metrics=True); rd["loss"].backward()
        dtask.exchange.check_overflow()
        assert rs["loss"].item() == rd["loss"].item()
        ps, gs = stask.two_tower_model.embedding_store.sparse_grad
        pd, gd = dtask.sharded_store.sparse_grad
        Us, Ud = int(ps.n_unique.item()), int(pd.n_unique.item())
        assert Us == Ud and torch.equal(ps.unique_rows[:Us], pd.unique_rows[:Ud]) and torch.equal(gs[:Us], gd[:Ud])
        for (n1, p1), (n2, p2) in zip(stask.named_parameters(), [(n, p) for n, p in dtask.named_parameters() if n != "embedding_shard"]):
            if "categorical_embedder" not in n1:
                assert torch.equal(p1.grad, p2.grad), n1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("score_dtype", ["fp8", "bf16"])
def test_configs4_full_size_properties(tt, score_dtype):
    """BASELINE configs[4] at its own size (final_embedding_dim 256, batch 65536; fp8 and bf16 score kernels) through
    size-independent properties: the loss of random unit rows is ln(B) to the operands' precision, swapping the towers leaves
    the loss untouched and swaps the gradients, every row of dN is orthogonal to nothing in particular but the gradient of
    a shifted loss... (linearity:) scaling d_loss scales the gradients exactly, and the step is bitwise reproducible."""
    from jodalrob_twotower_amd.two_tower_train_task import _ScoreCEFn
    B, D = 65536, 256
    g = torch.Generator(device=DEV).manual_seed(9)
    n = torch.nn.functional.normalize(torch.randn((B, D), generator=g, device=DEV), dim=1)
    c = torch.nn.functional.normalize(torch.randn((B, D), generator=g, device=DEV), dim=1)

    def run(a, b, seed_grad=1.0):
        a, b = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        loss, out8, _ = _ScoreCEFn.apply(a, b, 1.0, score_dtype, False, False)
        loss.backward(torch.full((), seed_grad, device=DEV))
        return loss.item(), a.grad, b.grad, out8

    l1, dn1, dc1, o1 = run(n, c)
    assert abs(l1 - np.log(B)) < 2e-2                              # scores of random unit rows in 256-d: |s| ~ 1/16
    l2, dn2, dc2, _ = run(n, c)
    assert l1 == l2 and torch.equal(dn1, dn2) and torch.equal(dc1, dc2)          # reproducible
    _, dn3, dc3, _ = run(n, c, 2.0)
    assert torch.equal(dn3, 2.0 * dn1) and torch.equal(dc3, 2.0 * dc1)           # linear in the incoming gradient (a power of two: exact)
    if score_dtype == "bf16":
        # swapping the towers: the same loss, the gradients swap (bf16: both images carry... only the notice one is scaled, so
        # the swap changes roundings: compare at the operands' precision)
        l4, dn4, dc4, _ = run(c, n)
        assert abs(l4 - l1) < 1e-4 * l1
        assert float((dn4 - dc1).norm() / dc1.norm()) < 2e-2 and float((dc4 - dn1).norm() / dn1.norm()) < 2e-2
    # the gradient of the loss w.r.t. a unit row is a combination of the other tower's rows: finite, no row exactly zero
    assert bool(torch.isfinite(dn1).all()) and bool(torch.isfinite(dc1).all()) and float(dn1.abs().sum(1).min()) > 0.0
    assert 0.0 <= o1[1].item() <= 1.0 and abs(o1[2].item()) < 0.01 and abs(o1[3].item()) < 1e-3
