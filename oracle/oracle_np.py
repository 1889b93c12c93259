"""CPU ORACLE (numpy) -- a from-scratch restatement of the reference's two-tower training step.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg as the CHECKER / reported baseline -- never by the product path
(jodalrob-twotower_amd/), which fails loudly without its HIP library.

Pinned (tests/test_oracle_golden.py) against golden vectors produced by the reference's own
PyTorch code in the build container (oracle/gen_golden.py -> tests/golden/*).

Each function cites the reference lines it follows (paths relative to /root/reference).
Everything is written by hand from the maths (forward AND backward), so that it is independent
both of torch.autograd and of the HIP kernels it checks.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5          # nn.BatchNorm1d default (src/towers/tower/base_tower.py:91)
BN_MOMENTUM = 0.1
NORM_EPS = 1e-12       # F.normalize default eps (base_tower.py:145)
LOG2E = 1.4426950408889634


# ------------------------------------------------------------------------------------------------
# operand rounding of the MEASURED mode (mlp_dtype="bf16", score_dtype="bf16"): the HIP kernels keep f32
# tensors and f32 accumulation but feed the matrix cores bf16 operands.  Passing `q=q_bf16` to the
# functions below rounds exactly those operands (and nothing else) in the restatement, so the f64 oracle
# and the kernels then differ by f32 accumulation order only.  q=None is the reference's arithmetic.
# ------------------------------------------------------------------------------------------------
def q_bf16(a):
    """round-to-nearest-even to bfloat16 (through f32, as the kernels do), returned in a's float dtype"""
    a = np.asarray(a)
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return r.astype(a.dtype if a.dtype.kind == "f" else np.float32)


def q_e4m3(a):
    """round-to-nearest-even to OCP fp8 e4m3fn (3 mantissa bits, exponent bias 7, subnormals down to 2^-9, largest finite
    448: saturating), returned in a's float dtype -- the operand rounding of the fp8 score kernels (score_dtype="fp8")"""
    a = np.asarray(a)
    x = np.asarray(a, dtype=np.float64)
    ax = np.abs(x)
    e = np.floor(np.log2(np.maximum(ax, 2.0 ** -20)))
    e = np.maximum(e, -6.0)                                  # below the smallest normal 2^-6 the step stays 2^-9
    step = 2.0 ** (e - 3)
    q = np.round(ax / step) * step                           # numpy rounds half to even
    q = np.minimum(q, 448.0)
    return (np.sign(x) * q).astype(a.dtype if a.dtype.kind == "f" else np.float32)


def _ident(a):
    return a


# ------------------------------------------------------------------------------------------------
# a2  id wire format  (src/towers/pairs/unified_bid_data_loader.py:827-841)
# ------------------------------------------------------------------------------------------------
def build_batch_kjt_values(ids_bk: np.ndarray):
    """[B,K] int64 -> (values [B*K] sample-major, lengths ones[B*K])."""
    ids_bk = np.asarray(ids_bk, dtype=np.int64)
    return ids_bk.reshape(-1).copy(), np.ones(ids_bk.size, dtype=np.int64)


# ------------------------------------------------------------------------------------------------
# a4  id unpack + clamp  (src/towers/cat_embed.py:88-123)
# ------------------------------------------------------------------------------------------------
def unpack_clamp_ids(values: np.ndarray, vocab_sizes) -> np.ndarray:
    """values i64 [B*K] sample-major; returns clamped ids [B,K] (v[b*K+k] clamped to [0,V_k-1])."""
    K = len(vocab_sizes)
    values = np.asarray(values, dtype=np.int64)
    B = values.size // K                                   # cat_embed.py:98 (floor division)
    ids = values[: B * K].reshape(B, K)
    hi = np.asarray(vocab_sizes, dtype=np.int64)[None, :] - 1
    return np.minimum(np.maximum(ids, 0), hi)              # torch.clamp(min=0,max=V-1): cat_embed.py:117


# ------------------------------------------------------------------------------------------------
# a5  per-key gather + concat  (src/towers/cat_embed.py:157-178)
# ------------------------------------------------------------------------------------------------
def embed_lookup(tables, ids_clamped: np.ndarray) -> np.ndarray:
    """tables: list of K arrays [V_k,E]; returns [B, K*E] in key order (exact copy of rows)."""
    return np.concatenate([tables[k][ids_clamped[:, k]] for k in range(len(tables))], axis=1)


def embed_grad_dense(d_concat: np.ndarray, ids_clamped: np.ndarray, vocab_sizes, E: int):
    """a16 embedding backward: dense [V_k,E] grads, duplicate ids summed (nn.Embedding sparse=False)."""
    grads = []
    for k, V in enumerate(vocab_sizes):
        g = np.zeros((V, E), dtype=d_concat.dtype)
        np.add.at(g, ids_clamped[:, k], d_concat[:, k * E:(k + 1) * E])
        grads.append(g)
    return grads


def embed_grad_sparse(d_concat: np.ndarray, ids_clamped: np.ndarray, row_offsets, E: int):
    """Sparse-unique form of the same gradient over the FUSED row space (row = offset_k + id):
    returns (unique_rows sorted ascending i64 [U], grad_rows [U,E]).  Sum order = ascending sample
    index b within each unique row (the deterministic order the HIP segment-reduce uses)."""
    B, K = ids_clamped.shape
    rows = (ids_clamped + np.asarray(row_offsets, dtype=np.int64)[None, :]).reshape(-1)
    vals = d_concat.reshape(B * K, E)
    order = np.argsort(rows, kind="stable")
    rs, vs = rows[order], vals[order]
    uniq, start = np.unique(rs, return_index=True)
    out = np.add.reduceat(vs, start, axis=0) if len(rs) else np.zeros((0, E), d_concat.dtype)
    return uniq.astype(np.int64), out.astype(d_concat.dtype)


# ------------------------------------------------------------------------------------------------
# a6/a7  tower  (src/towers/tower/base_tower.py:71-147)
# ------------------------------------------------------------------------------------------------
def tower_layout(state: dict, prefix: str, keys):
    """Reads the layer structure off the state dict: returns (n_hidden_blocks, final_idx)."""
    i = 0
    while f"{prefix}mlp.{4 * i + 2}.running_mean" in state:
        i += 1
    return i, 4 * i


def tower_fwd(state: dict, prefix: str, keys, vocab_sizes, dense, values, train: bool, dtype=np.float32, q=None):
    """BaseTower.forward.  Returns (emb [B,D], cache, bn_updates).
    x = cat[dense W0^T + b0 | embed concat]          base_tower.py:133-139
    per hidden block: BN(ReLU(x W^T + b)) (dropout p=0 / eval: identity)   base_tower.py:88-93
    y = h W_f^T + b_f ; y / max(||y||, 1e-12)        base_tower.py:97,145
    q: operand rounding of the bf16 mode (q_bf16) -- every Linear's two operands, and the stored tower input x.
    """
    f = lambda a: np.asarray(a, dtype=dtype)
    q = q or _ident
    ids = unpack_clamp_ids(values, vocab_sizes)
    # (rows are gathered before the dtype conversion: a 1 M-row table is never copied whole)
    rows = [f(np.asarray(state[f"{prefix}categorical_embedder.embeddings.{k}.weight"])[ids[:, i]]) for i, k in enumerate(keys)]
    W0, b0 = f(state[prefix + "dense_projection.weight"]), f(state[prefix + "dense_projection.bias"])
    dense = f(dense)
    x = q(np.concatenate([q(dense) @ q(W0).T + b0] + rows, axis=1))
    nblk, fin = tower_layout(state, prefix, keys)
    cache = {"ids": ids, "dense": dense, "x": x, "blocks": [], "E": rows[0].shape[1] if rows else 0,
             "H0": W0.shape[0], "q": q}
    bn_updates = {}
    h = x
    for i in range(nblk):
        W, b = f(state[f"{prefix}mlp.{4 * i}.weight"]), f(state[f"{prefix}mlp.{4 * i}.bias"])
        g, be = f(state[f"{prefix}mlp.{4 * i + 2}.weight"]), f(state[f"{prefix}mlp.{4 * i + 2}.bias"])
        pre = q(h) @ q(W).T + b
        a = np.maximum(pre, 0)
        if train:
            mean = a.mean(axis=0)
            var = a.var(axis=0)                                         # biased, used to normalise
            n = a.shape[0]
            rm, rv = f(state[f"{prefix}mlp.{4 * i + 2}.running_mean"]), f(state[f"{prefix}mlp.{4 * i + 2}.running_var"])
            bn_updates[f"{prefix}mlp.{4 * i + 2}.running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
            bn_updates[f"{prefix}mlp.{4 * i + 2}.running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * (n / max(n - 1, 1))
            bn_updates[f"{prefix}mlp.{4 * i + 2}.num_batches_tracked"] = \
                np.asarray(state[f"{prefix}mlp.{4 * i + 2}.num_batches_tracked"]) + 1
        else:
            mean, var = f(state[f"{prefix}mlp.{4 * i + 2}.running_mean"]), f(state[f"{prefix}mlp.{4 * i + 2}.running_var"])
        rstd = 1.0 / np.sqrt(var + dtype(BN_EPS))
        xhat = (a - mean) * rstd
        out = xhat * g + be
        cache["blocks"].append({"inp": h, "W": W, "pre": pre, "xhat": xhat, "rstd": rstd, "g": g})
        h = out
    Wf, bf = f(state[f"{prefix}mlp.{fin}.weight"]), f(state[f"{prefix}mlp.{fin}.bias"])
    y = q(h) @ q(Wf).T + bf
    nrm = np.sqrt((y * y).sum(axis=1, keepdims=True))
    den = np.maximum(nrm, dtype(NORM_EPS))
    emb = y / den
    cache.update({"h_last": h, "Wf": Wf, "y": y, "den": den, "nrm": nrm, "emb": emb, "train": train})
    return emb, cache, bn_updates


def tower_bwd(cache: dict, d_emb: np.ndarray, prefix: str, keys, vocab_sizes, table_grads: str = "dense",
              proj_grad: str = "direct"):
    """Backward of tower_fwd (train-mode BN, dropout p=0).  Returns {state_dict key: grad}.
    table_grads: "dense" = the reference's dense [V_k, E] arrays; "none" = skip them (the caller takes the
    slot gradients `_d_concat` and forms the sparse-unique rows itself: 1 M-row tables).
    proj_grad: how the dense projection's gradients are formed under operand rounding (without rounding the two are the
    same numbers): "direct" = d_proj^T . dense with d_proj = (d_pre . W)[:, :H0] rounded as a GEMM operand (autograd's
    order); "factored" = W[:, :H0]^T . (d_pre^T . dense), the one-launch first-block backward of the HIP path (edge-free
    shapes): d_pre, dense and W are rounded GEMM operands, the intermediate product is not (the kernel carries it as hi + lo
    bf16 pairs); the bias gradient W[:, :H0]^T . colsum(d_pre) is formed unrounded."""
    grads = {}
    q = cache.get("q") or _ident
    emb, den, nrm = cache["emb"], cache["den"], cache["nrm"]
    # y/max(||y||,eps): for ||y||>eps  dy = (d - emb*(emb.d))/||y|| ; else dy = d/eps
    dot = (emb * d_emb).sum(axis=1, keepdims=True)
    dy = np.where(nrm > NORM_EPS, (d_emb - emb * dot) / den, d_emb / den)
    nblk = len(cache["blocks"])
    fin = 4 * nblk
    grads[f"{prefix}mlp.{fin}.weight"] = q(dy).T @ q(cache["h_last"])
    grads[f"{prefix}mlp.{fin}.bias"] = dy.sum(axis=0)
    dh = q(dy) @ q(cache["Wf"])
    for i in reversed(range(nblk)):
        blk = cache["blocks"][i]
        xhat, rstd, g = blk["xhat"], blk["rstd"], blk["g"]
        grads[f"{prefix}mlp.{4 * i + 2}.weight"] = (dh * xhat).sum(axis=0)
        grads[f"{prefix}mlp.{4 * i + 2}.bias"] = dh.sum(axis=0)
        if cache["train"]:
            dxh = dh * g
            da = rstd * (dxh - dxh.mean(axis=0) - xhat * (dxh * xhat).mean(axis=0))
        else:
            da = dh * g * rstd
        dpre = da * (blk["pre"] > 0)
        grads[f"{prefix}mlp.{4 * i}.weight"] = q(dpre).T @ q(blk["inp"])
        grads[f"{prefix}mlp.{4 * i}.bias"] = dpre.sum(axis=0)
        dh = q(dpre) @ q(blk["W"])
    H0, E = cache["H0"], cache["E"]
    dproj, dcat = dh[:, :H0], dh[:, H0:]
    if proj_grad == "factored" and nblk >= 1:
        Wp = cache["blocks"][0]["W"][:, :H0]
        grads[prefix + "dense_projection.weight"] = q(Wp).T @ (q(dpre).T @ q(cache["dense"]))
        grads[prefix + "dense_projection.bias"] = Wp.T @ dpre.sum(axis=0)
    elif proj_grad in ("direct", "factored"):
        grads[prefix + "dense_projection.weight"] = q(dproj).T @ q(cache["dense"])
        grads[prefix + "dense_projection.bias"] = dproj.sum(axis=0)
    else:
        raise ValueError(f"proj_grad must be 'direct' or 'factored', got {proj_grad!r}")
    if table_grads == "dense":
        for k, g in zip(keys, embed_grad_dense(dcat, cache["ids"], vocab_sizes, E)):
            grads[f"{prefix}categorical_embedder.embeddings.{k}.weight"] = g
    grads["_d_concat"] = dcat
    return grads


# ------------------------------------------------------------------------------------------------
# a11-a13  score matrix, loss, metrics  (src/towers/two_tower_train_task.py:99-179)
# ------------------------------------------------------------------------------------------------
def _logsumexp(S, axis):
    m = S.max(axis=axis, keepdims=True)
    return (m + np.log(np.exp(S - m).sum(axis=axis, keepdims=True))).squeeze(axis)


def score_ce_fwd(N, C, temperature=1.0):
    """S = N C^T (/T iff T != 1)  :99-112 ;  loss = 0.5*(CE(S,diag)+CE(S^T,diag))  :114-134 ;
    metrics :162-179.  Returns (loss, metrics dict, S, (lse_row, lse_col))."""
    S = N @ C.T
    if temperature != 1.0:
        S = S / N.dtype.type(temperature)
    B = S.shape[0]
    diag = np.diagonal(S)
    lse_r, lse_c = _logsumexp(S, 1), _logsumexp(S, 0)
    loss = 0.5 * ((lse_r - diag).mean() + (lse_c - diag).mean())
    acc = (S.argmax(axis=1) == np.arange(B)).astype(np.float32).mean()
    pos = diag.mean()
    neg = (S.sum() - diag.sum()) / max(B * B - B, 1) if B > 1 else np.float32("nan")
    return loss, {"accuracy": acc, "positive_similarity_mean": pos, "negative_similarity_mean": neg,
                  "similarity_gap": pos - neg}, S, (lse_r, lse_c)


def q_block_e4m3(W, axis):
    """Block-scaled e4m3 rounding of the softmax weights along `axis` (score_bwd_rows8_kernel, include/twotower.h
    TT_OPT_FP8_GRAD): blocks of 32 consecutive indices (one 32-row tile of the summed operand = one scale block of
    v_mfma_scale_f32_32x32x64_f8f6f4); per block and per position on the other axis, scale = 2^(floor(log2 m) - 7) with m
    the block's largest entry (floored at 2^-103), and the entries become scale * e4m3_rne(w / scale).  W >= 0."""
    W = np.moveaxis(np.asarray(W, dtype=np.float64), axis, -1)
    A, B = W.shape
    Bp = (B + 31) // 32 * 32
    Wp = np.zeros((A, Bp))
    Wp[:, :B] = W
    Wb = Wp.reshape(A, Bp // 32, 32)
    m = np.maximum(Wb.max(axis=2, keepdims=True), 2.0 ** -103)
    scale = np.ldexp(1.0, np.frexp(m)[1] - 1 - 7)             # frexp: m = f * 2^e with f in [0.5, 1)
    Q = (q_e4m3(Wb / scale) * scale).reshape(A, Bp)[:, :B]
    return np.moveaxis(Q, -1, axis)


def score_ce_bwd(N, C, S, lse, temperature=1.0, dloss=1.0, q=None, prod_operands=None, block_fp8=False):
    """dS = (softmax_rows + softmax_cols - 2I)/(2B) ; dN = dS C / T ; dC = dS^T N / T.
    q: the bf16 score kernels round the weight matrix (softmax_rows + softmax_cols - 2I) once more before the
    gradient products (it is the second MFMA's operand).  prod_operands = (N', C'): the operands of the two gradient
    products when they differ from those S was formed from (fp8 score kernels with bf16 gradient products: S from e4m3
    operands, products from bf16 ones).
    block_fp8 (the fp8 score kernels' default, TT_OPT_FP8_GRAD 1): the gradient products take the e4m3 operands N, C that S was
    formed from and block-scaled e4m3 weights (q_block_e4m3: blocks run along the summed index, i.e. along b for dN and along
    a for dC); the diagonal's weight w_aa - 2 stays exact and multiplies prod_operands' (bf16) row."""
    B = S.shape[0]
    W = np.exp(S - lse[0][:, None])
    W += np.exp(S - lse[1][None, :])
    k = dloss / (2 * B)
    if temperature != 1.0:
        k = k / S.dtype.type(temperature)
    if block_fp8:
        N16, C16 = prod_operands
        wd = np.diagonal(W).copy() - 2
        W[np.arange(B), np.arange(B)] = 0
        dN = q_block_e4m3(W, 1) @ C + wd[:, None] * C16
        dC = q_block_e4m3(W, 0).T @ N + wd[:, None] * N16
        return dN * k, dC * k
    if prod_operands is not None:
        N, C = prod_operands
    W[np.arange(B), np.arange(B)] -= 2
    if q is not None:
        W = q(W)
    W *= k
    return W @ C, W.T @ N


def score_operands_fp8(N, C, temperature=1.0):
    """The S-product operands of the fp8 score kernels: fp8(64 * scale * x) / (64 * scale), scale = 1/T * log2(e) on the
    notice side and 1 on the company side (tt_score_pack2_fp8); the products are formed through f32 first, as the pack kernel does."""
    sn = np.float32(np.float32(1.0 / temperature) * np.float32(LOG2E))
    n32 = np.asarray(N, dtype=np.float32) * sn * np.float32(64.0)
    c32 = np.asarray(C, dtype=np.float32) * np.float32(64.0)
    return (q_e4m3(n32).astype(N.dtype) / N.dtype.type(float(sn) * 64.0), q_e4m3(c32).astype(C.dtype) / C.dtype.type(64.0))


def score_operands_bf16(N, C, temperature=1.0):
    """The operand images of the bf16 score kernels: the notice rows are packed times 1/T * log2(e) (the softmax's
    exponent scale rides in the MFMA: tt_score_unit_scale), so the operand is bf16(scale * n) / scale."""
    sn = float(np.float32(np.float32(1.0 / temperature) * np.float32(LOG2E)))
    return q_bf16(np.asarray(N) * N.dtype.type(sn)) / N.dtype.type(sn), q_bf16(C)


# ------------------------------------------------------------------------------------------------
# a9/a10  TwoTowerModel.forward + TwoTowerTrainTask.forward (+ a16 backward)
COS_EPS = 1e-12      # EPSILON of ATen's cosine_embedding_loss


def score_variant_fwd_bwd(N, C, temperature=1.0, loss_type="cross_entropy", label_smoothing=0.0, dloss=1.0):
    """The loss variants of two_tower_train_task.py:114-160 on the materialised S = N C^T / T, with their gradients:
      cross_entropy + label_smoothing e (:118-133, F.cross_entropy(label_smoothing=e) on S and S^T):
          row term = lse_a - (1 - e) s_aa - (e / B) sum_b s_ab ; loss = (mean rows + mean columns) / 2
      cosine_embedding (:135-158): F.cosine_embedding_loss([s], [1], +-1) = 1 - cos on the diagonal, max(0, cos) off it,
          cos = s / sqrt((s^2 + 1e-12)(1 + 1e-12)), mean over all B^2 entries.
    Returns (loss, metrics, S, dN, dC)."""
    _, metrics, S, lse = score_ce_fwd(N, C, temperature)
    B = S.shape[0]
    eye = np.eye(B, dtype=bool)
    if loss_type == "cross_entropy":
        e = S.dtype.type(label_smoothing)
        d = np.diagonal(S)
        rows = lse[0] - (1 - e) * d - e / B * S.sum(axis=1)
        cols = lse[1] - (1 - e) * d - e / B * S.sum(axis=0)
        loss = 0.5 * (rows.mean() + cols.mean())
        onehot = np.where(eye, 1 - e, 0).astype(S.dtype)
        dS = 0.5 / B * ((np.exp(S - lse[0][:, None]) - onehot - e / B) + (np.exp(S - lse[1][None, :]) - onehot - e / B))
    elif loss_type == "cosine_embedding":
        q = S * S + COS_EPS
        cos = S / np.sqrt(q * (1 + COS_EPS))
        loss = np.where(eye, 1 - cos, np.maximum(cos, 0)).mean()
        dcos = COS_EPS / (q * np.sqrt(q) * np.sqrt(1 + COS_EPS))
        dS = np.where(eye, -dcos, np.where(cos > 0, dcos, 0)) / (B * B)
    else:
        raise ValueError(f"Unsupported loss_type: {loss_type}")
    dS = dS * S.dtype.type(dloss) / S.dtype.type(temperature)
    return loss, metrics, S, dS @ C, dS.T @ N


# ------------------------------------------------------------------------------------------------
NT, CT = "two_tower_model.notice_tower.", "two_tower_model.company_tower."


def task_step(state, batch, keys_n, keys_c, vocab_n, vocab_c, temperature=1.0, train=True, backward=True,
              dtype=np.float32, rounding=None, table_grads="dense", keep_sim=True, proj_grad="direct",
              loss_type="cross_entropy", label_smoothing=0.0, score_rounding=None):
    """One forward (+backward) of the task.  batch: dict with notice_ids/company_ids [B,K] (or flat
    values) and notice_dense/company_dense.  Returns dict(loss, metrics, sim, notice_emb, company_emb,
    grads{state key: array}, bn_updates).
    rounding="bf16": the operand rounding of the measured mode (mlp_dtype = score_dtype = "bf16"), see q_bf16.
    score_rounding="fp8" (with rounding="bf16"; BASELINE configs[4], score_dtype="fp8"): the score matrix AND the two gradient
    products are formed from e4m3 operands (score_operands_fp8), the softmax weights block-scaled e4m3 (q_block_e4m3), the
    diagonal's weight exact on the bf16 operand's row.  score_rounding="fp8_s" (TT_OPT_FP8_GRAD 0): only the score matrix from
    e4m3 operands; gradient products from the bf16 ones with bf16 weights."""
    q = q_bf16 if rounding == "bf16" else None
    if rounding not in (None, "bf16"):
        raise ValueError(f"rounding must be None or 'bf16', got {rounding!r}")
    if score_rounding not in (None, "fp8", "fp8_s") or (score_rounding is not None and q is None):
        raise ValueError("score_rounding must be None, or 'fp8' / 'fp8_s' together with rounding='bf16'")
    vals_n = np.asarray(batch["notice_ids"]).reshape(-1)
    vals_c = np.asarray(batch["company_ids"]).reshape(-1)
    if batch["notice_dense"].shape[0] != batch["company_dense"].shape[0]:
        raise ValueError("notice/company batch size mismatch")          # two_tower_train_task.py:64-67
    ne, cn, bn_n = tower_fwd(state, NT, keys_n, vocab_n, batch["notice_dense"], vals_n, train, dtype, q)
    ce, cc, bn_c = tower_fwd(state, CT, keys_c, vocab_c, batch["company_dense"], vals_c, train, dtype, q)
    sn, sc = score_operands_bf16(ne, ce, temperature) if q is not None else (ne, ce)
    prod = None
    if score_rounding is not None:
        prod = (sn, sc)
        sn, sc = score_operands_fp8(ne, ce, temperature)
    variant = loss_type != "cross_entropy" or label_smoothing != 0.0          # the dense loss path (f32 only)
    if variant:
        if q is not None:
            raise ValueError("the loss variants run in f32 (no operand rounding)")
        loss, metrics, S, dN, dC = score_variant_fwd_bwd(sn, sc, temperature, loss_type, label_smoothing)
    else:
        loss, metrics, S, lse = score_ce_fwd(sn, sc, temperature)
    out = {"loss": loss, **metrics, "sim": S if keep_sim else None, "notice_emb": ne, "company_emb": ce,
           "bn_updates": {**bn_n, **bn_c}}
    if backward:
        if not variant:
            dN, dC = score_ce_bwd(sn, sc, S, lse, temperature, q=q, prod_operands=prod, block_fp8=score_rounding == "fp8")
        del S
        g = tower_bwd(cn, dN, NT, keys_n, vocab_n, table_grads, proj_grad)
        out["d_concat_notice"] = g.pop("_d_concat")
        g2 = tower_bwd(cc, dC, CT, keys_c, vocab_c, table_grads, proj_grad)
        out["d_concat_company"] = g2.pop("_d_concat")
        out["grads"] = {**g, **g2}
        out["ids_notice"], out["ids_company"] = cn["ids"], cc["ids"]
    return out


# ------------------------------------------------------------------------------------------------
# a15  predict_batch top-k  (two_tower_train_task.py:181-207) and evaluator (src/evaluation/evaluator.py:20-71)
# ------------------------------------------------------------------------------------------------
def topk_rows(S, k):
    """values/indices of the k largest per row, descending; ties -> lower column index first."""
    idx = np.argsort(-S, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(S, idx, axis=1), idx


def diag_rank(S):
    """0-based rank of S[i,i] in row i under a descending sort (count of strictly greater entries;
    equals the evaluator's argsort position when there are no ties with the diagonal)."""
    d = np.diagonal(S)[:, None]
    return (S > d).sum(axis=1)


def diag_rank_stable(S):
    """Position of column i in a STABLE descending sort of row i: ties with the diagonal count only when they sit at a lower
    column index.  The reference ranks with torch.topk / torch.argsort (evaluator.py:34, :58), whose order among equal scores is
    unspecified; the HIP path (tt_diag_rank_rows, the rank output of the score sweeps) fixes it to this rule.  Equal scores
    are real in this workload: two pairs of a batch that share a company have identical company embeddings."""
    order = np.argsort(-S, axis=1, kind="stable")
    return (order == np.arange(S.shape[0])[:, None]).argmax(axis=1)


def recall_at_k(S, k):
    return (diag_rank(S) < min(k, S.shape[1])).astype(np.float32).mean()          # evaluator.py:20-43


def mrr(S):
    return (1.0 / (diag_rank(S) + 1.0)).astype(np.float32).mean()                 # evaluator.py:45-71


# ------------------------------------------------------------------------------------------------
# a17  optimiser + schedule  (scripts/train.py:231-242)
# ------------------------------------------------------------------------------------------------
def warmup_lr(base_lr, step, warmup_steps):
    """LambdaLR: lr used by optimiser step number `step` (0-based); step 0 has lr 0."""
    return base_lr * (step / warmup_steps if step < warmup_steps else 1.0)


def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.Adam (coupled L2 weight decay), t = 1-based step count.  In-place on p, m, v."""
    if wd:
        g = g + wd * p
    m *= b1
    m += (1 - b1) * g
    v *= b2
    v += (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    p -= (lr / bc1) * m / (np.sqrt(v) / np.sqrt(bc2) + eps)


def sparse_adam_rows(table, m, v, step_rows, rows, grad_rows, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """Row-wise sparse Adam of this build (NOT in the reference; DESIGN.md 'optimiser semantics'):
    only the looked-up rows are touched; each row keeps its own step count for bias correction."""
    for r, g in zip(rows, grad_rows):
        step_rows[r] += 1
        adam_step(table[r], g.astype(table.dtype), m[r], v[r], int(step_rows[r]), lr, b1, b2, eps, wd)


# ------------------------------------------------------------------------------------------------
# a18  FeatureProjector + projection concat  (src/torchrec_preprocess/feature_projector.py:20-28,
#      feature_preprocessor.py:150-233 / unified_bid_data_loader.py:1380-1448)
# ------------------------------------------------------------------------------------------------
def mlp2(x, w0, b0, w1, b1):
    return np.maximum(x @ w0.T + b0, 0) @ w1.T + b1


def project_features(pstate, numeric, text_dict, text_cols=None):
    """dense_projected = cat[ num_proj(numeric) | text_proj(text[col]) for col in text_cols ]."""
    parts = []
    if numeric is not None:
        parts.append(mlp2(numeric, pstate["num_proj.0.weight"], pstate["num_proj.0.bias"],
                          pstate["num_proj.2.weight"], pstate["num_proj.2.bias"]))
    if numeric is None or text_dict:
        for col in (text_cols or list(text_dict)):
            if col in text_dict:
                parts.append(mlp2(text_dict[col], pstate["text_proj.0.weight"], pstate["text_proj.0.bias"],
                                  pstate["text_proj.2.weight"], pstate["text_proj.2.bias"]))
    return np.concatenate(parts, axis=1) if parts else None


def build_id_mappings(stores):
    """feature_preprocessor.py:235-268."""
    n2i, c2i = {}, {}
    if "notice" in stores:
        n2i = {tuple(p): i for i, p in enumerate(stores["notice"].get("ids", []))}
    if "company" in stores:
        cids = stores["company"].get("ids", [])
        if len(cids):
            if isinstance(cids[0], (tuple, list)):
                c2i = {str(t[0]): i for i, t in enumerate(cids)}
            else:
                c2i = {str(c): i for i, c in enumerate(cids)}
    return n2i, c2i
