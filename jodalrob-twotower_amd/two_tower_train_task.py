"""TwoTowerTrainTask -- drop-in for src/towers/two_tower_train_task.py:9-293 on the HIP path.

forward(batch, return_metrics) -> loss, or a dict with loss / accuracy / positive_similarity_mean /
negative_similarity_mean / similarity_gap / similarity_matrix (same keys as :89-95).

The B x B score matrix is never materialised on the training path: tt_score_dir_fwd sweeps it tile
by tile on MFMA for both softmax directions, tt_score_loss_finish forms loss and metrics, and
tt_score_dir_bwd recomputes the tiles for the gradients.  `similarity_matrix` is produced on demand
(eagerly for small batches, lazily above LAZY_SIM_BATCH) by tt_score_matrix.
"""
from __future__ import annotations

from typing import Dict, Union

import torch
from ._lib import no_dynamo as _no_dynamo
import torch.nn as nn

from . import ops
from .config import settings
from .two_tower_model import TwoTowerModel, create_two_tower_model

LAZY_SIM_BATCH = 2048


class _DenseLossFn(torch.autograd.Function):
    """The loss variants the fused kernels do not cover, on the materialised score matrix (tt_score_dense_fwd / _bwd):
    loss_type 0 = cross-entropy with label smoothing (:114-133), 1 = cosine-embedding loss (:135-158).  O(B^2) memory."""

    @staticmethod
    def forward(ctx, n, c, inv_t, loss_type, label_smoothing):
        n, c = n.contiguous().float(), c.contiguous().float()
        S, stats, out8, loss = ops.score_dense_fwd(n, c, inv_t, loss_type, label_smoothing)
        ctx.save_for_backward(n, c, S, stats)
        ctx.cfg = (inv_t, loss_type, label_smoothing)
        ctx.mark_non_differentiable(out8)
        return loss.reshape(()), out8

    @staticmethod
    def backward(ctx, d_loss, _d_out8):
        n, c, S, stats = ctx.saved_tensors
        inv_t, loss_type, label_smoothing = ctx.cfg
        # tt_score_dense_bwd overwrites its S argument with dS through a raw pointer (no version counter sees it): work on a copy,
        # so a second backward over the same graph (retain_graph=True) starts from the scores again
        dN, dC = ops.score_dense_bwd(n, c, inv_t, loss_type, label_smoothing, S.clone(), stats, d_loss.reshape(1).contiguous().float())
        return dN, dC, None, None, None


class _ScoreCEFn(torch.autograd.Function):
    """loss = 0.5 * [CE(S, diag) + CE(S^T, diag)],  S = N C^T / T   (:99-134); out8 carries the metrics.
    score_dtype 'fp32': exact-f32 MFMA path (parity); 'bf16': bf16-operand MFMA fast path."""

    @staticmethod
    def forward(ctx, n, c, inv_t, score_dtype, want_col_rank=True, full_rank=True, packed_n=None, packed_c=None, scale_n=None):
        n, c = n.contiguous().float(), c.contiguous().float()
        B, D = n.shape
        shift = abs(inv_t)                                   # unit rows: |s| <= 1/T
        if score_dtype == "fp8":
            # e4m3 operands for the S products (v_mfma_scale_f32_32x32x64_f8f6f4: twice the bf16 MFMA rate), everything else as
            # the bf16 path; always the single-pass forward and the workgroup-staged backward (BASELINE configs[4])
            scale_n = ops.score_unit_scale(inv_t)
            Np, Cp = ops.score_pack2_fp8(n, c, scale_n, 1.0)
            rowsum, colsum, diag, row_rank, inv, out8, loss = ops.score_fwd_sym(Np, Cp, B, D, inv_t, shift, scale_n, True, fp8=True)
            ctx.packed = (Np, Cp, scale_n, inv)
            ctx.fp8 = True
            ctx.save_for_backward(n, c, rowsum, colsum)
            ctx.inv_t, ctx.shift = inv_t, shift
            ctx.mark_non_differentiable(out8, row_rank)
            ctx.set_materialize_grads(False)
            return loss, out8, row_rank
        ctx.fp8 = False
        if score_dtype == "bf16":
            # scale_n: the notice image holds bf16(scale_n * n) -- with scale_n = inv_t * log2(e) the exponent scale of the
            # softmax rides in the MFMA and the kernels skip a multiply-add per score (results are scale-free).
            # The towers' fused tail can emit the packed operand images itself (tt_tower_acts.emb_packed): one launch fewer.
            if packed_n is not None and packed_c is not None:
                Np, Cp = packed_n, packed_c
                scale_n = 1.0 if scale_n is None else scale_n
            else:
                scale_n = ops.score_unit_scale(inv_t) if scale_n is None else scale_n
                Np, Cp = ops.score_pack2_bf16(n, c, scale_n, 1.0)
            if not want_col_rank and not full_rank:
                # steady state of training: ONE sweep of the B x B tiles serves both softmax directions (tt_score_fwd_sym_bf16)
                rowsum, colsum, diag, row_rank, inv, out8, loss = ops.score_fwd_sym(Np, Cp, B, D, inv_t, shift, scale_n, True)
                ctx.packed = (Np, Cp, scale_n, inv)
                ctx.save_for_backward(n, c, rowsum, colsum)
                ctx.inv_t, ctx.shift = inv_t, shift
                ctx.mark_non_differentiable(out8, row_rank)
                ctx.set_materialize_grads(False)
                return loss, out8, row_rank
            rowsum, colsum, diag, row_rank, col_rank, sumscore, inv = ops.score_fwd_bf16(Np, Cp, B, D, inv_t, shift, want_col_rank,
                                                                                         full_rank, scale_n, with_inv=True)
            if not want_col_rank:
                col_rank = row_rank                      # placeholder: column top-1 rate is only a first-call diagnostic
            ctx.packed = (Np, Cp, scale_n, inv)
        else:
            rowsum, diag, row_rank, sumscore = ops.score_dir_fwd(n, c, inv_t, shift, 0, True)
            colsum, _, col_rank, _ = ops.score_dir_fwd(c, n, inv_t, shift, 0, False)
            ctx.packed = None
        out8, loss = ops.score_loss_finish(B, shift, rowsum, colsum, diag, row_rank, col_rank, sumscore)
        ctx.save_for_backward(n, c, rowsum, colsum)
        ctx.inv_t, ctx.shift = inv_t, shift
        ctx.mark_non_differentiable(out8, row_rank)
        ctx.set_materialize_grads(False)                     # else autograd zero-fills grads for out8 / row_rank every step
        return loss, out8, row_rank

    @staticmethod
    def backward(ctx, d_loss, _d_out8, _d_rank):
        n, c, rowsum, colsum = ctx.saved_tensors
        B, D = n.shape
        if d_loss is None:
            return None, None, None, None, None, None, None, None, None
        if d_loss.dtype != torch.float32 or not d_loss.is_contiguous():
            d_loss = d_loss.contiguous().float()
        scale = ctx.inv_t / (2.0 * B)
        if ctx.packed is not None:
            dN, dC = ops.score_bwd_bf16(ctx.packed[0], ctx.packed[1], B, D, ctx.inv_t, ctx.shift, rowsum, colsum, d_loss, scale,
                                        ctx.packed[2], ctx.packed[3], fp8=ctx.fp8)
        else:
            dN = ops.score_dir_bwd(n, c, ctx.inv_t, ctx.shift, 0, rowsum, colsum, d_loss, scale)
            dC = ops.score_dir_bwd(c, n, ctx.inv_t, ctx.shift, 0, colsum, rowsum, d_loss, scale)
        return dN, dC, None, None, None, None, None, None, None


class _Result(dict):
    """Result dict whose 'similarity_matrix' entry is computed on first access (large batches)."""

    def __init__(self, *a, sim_thunk=None, out8=None, **k):
        super().__init__(*a, **k)
        self._sim_thunk = sim_thunk
        self.out8 = out8          # the finish kernel's 8 floats (loss, accuracy, pos, neg, gap, ...): the entries above are views of it

    def _materialise(self):
        if self._sim_thunk is not None and not dict.__contains__(self, "similarity_matrix"):
            dict.__setitem__(self, "similarity_matrix", self._sim_thunk())
            self._sim_thunk = None

    def __missing__(self, key):
        if key == "similarity_matrix" and self._sim_thunk is not None:
            self._materialise()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or (key == "similarity_matrix" and self._sim_thunk is not None)

    def get(self, key, default=None):
        if key == "similarity_matrix":
            self._materialise()
        return dict.get(self, key, default)

    def keys(self):
        self._materialise()
        return dict.keys(self)

    def items(self):
        self._materialise()
        return dict.items(self)

    def values(self):
        self._materialise()
        return dict.values(self)


class TwoTowerTrainTask(nn.Module):
    def __init__(self, two_tower_model: TwoTowerModel, temperature: float = 1.0, loss_type: str = "cross_entropy",
                 label_smoothing: float = 0.0, score_dtype: str = None):
        super().__init__()
        self.score_dtype = score_dtype or settings.score_dtype
        if self.score_dtype not in ("fp32", "bf16", "fp8"):
            raise ValueError(f"score_dtype must be 'fp32', 'bf16' or 'fp8', got {self.score_dtype!r}")
        if self.score_dtype == "bf16" and settings.tower_pack:       # (settings.tower_pack = False: separate pack launch; tests)
            for tw in (two_tower_model.notice_tower, two_tower_model.company_tower):
                tw.pack_for_score = True
            # the notice image carries the softmax's exponent scale (see _ScoreCEFn)
            two_tower_model.notice_tower.pack_scale = ops.score_unit_scale(1.0 / float(temperature))
        self.two_tower_model = two_tower_model
        self.temperature = temperature
        self.loss_type = loss_type
        self.label_smoothing = label_smoothing
        if loss_type not in ["cross_entropy", "cosine_embedding"]:
            raise ValueError(f"Unsupported loss_type: {loss_type}")                         # :37-38
        if not 0.0 <= float(label_smoothing) <= 1.0:
            raise ValueError(f"label_smoothing must be in [0, 1], got {label_smoothing}")
        # cosine-embedding loss (:135-158) and label smoothing (never set by the reference's factory or driver) run on the
        # dense path: the score matrix is materialised (O(B^2) memory), f32 throughout -- see _DenseLossFn
        self._dense_loss = loss_type != "cross_entropy" or float(label_smoothing) != 0.0

    # ---- forward: :40-97 ----------------------------------------------------------------------------
    @_no_dynamo
    def forward(self, batch, return_metrics: bool = False):
        notice_input, company_input = batch["notice"], batch["company"]
        nb, cb = notice_input["dense"].size(0), company_input["dense"].size(0)
        if nb != cb:                                                                         # :64-67
            raise ValueError(f"Notice와 Company 배치 크기가 다릅니다: {nb} vs {cb}")
        notice_embeddings, company_embeddings = self.two_tower_model(notice_input, company_input)
        loss, out8, _ = self._score_ce(notice_embeddings, company_embeddings, 1.0 / float(self.temperature),
                                       not hasattr(self, "_pair_check_done"))
        if not hasattr(self, "_pair_check_done"):                                            # :82-84
            self._verify_positive_pair_alignment(out8)
            self._pair_check_done = True
        if not return_metrics:
            return loss
        n_det, c_det, inv_t = notice_embeddings.detach(), company_embeddings.detach(), 1.0 / float(self.temperature)
        res = _Result({"loss": loss, "accuracy": out8[1], "positive_similarity_mean": out8[2],
                       "negative_similarity_mean": out8[3], "similarity_gap": out8[4]},
                      sim_thunk=lambda: ops.score_matrix(n_det, c_det, inv_t), out8=out8)
        if nb <= LAZY_SIM_BATCH:
            res._materialise()
        return res

    def forward_with_ranks(self, batch):
        """(result dict as forward(batch, return_metrics=True), 0-based rank of each positive in its row) from ONE pass
        through the towers -- what the evaluator needs per batch (src/evaluation/evaluator.py:123-155 calls the model once
        and ranks its similarity matrix).  Ranks come from the exact-f32 score sweep, as diagonal_ranks()."""
        with torch.no_grad():
            n, c = self.two_tower_model(batch["notice"], batch["company"])
            if n.size(0) != c.size(0):
                raise ValueError(f"Notice와 Company 배치 크기가 다릅니다: {n.size(0)} vs {c.size(0)}")
            inv_t = 1.0 / float(self.temperature)
            loss, out8, _ = self._score_ce(n, c, inv_t, False)
            _, _, rank, _ = ops.score_dir_fwd(n.contiguous(), c.contiguous(), inv_t, abs(inv_t), 0, False)
            n_det, c_det = n.detach(), c.detach()
            res = _Result({"loss": loss, "accuracy": out8[1], "positive_similarity_mean": out8[2],
                           "negative_similarity_mean": out8[3], "similarity_gap": out8[4]},
                          sim_thunk=lambda: ops.score_matrix(n_det, c_det, inv_t))
            return res, rank

    def _score_ce(self, n, c, inv_t, first_call):
        """(loss, out8, row_rank) of the symmetric in-batch-negative softmax-CE (:99-134); the sharded task overrides it."""
        if self._dense_loss:
            if n.shape != c.shape:
                raise ValueError("the dense loss path needs as many notice rows as company rows")
            loss, out8 = _DenseLossFn.apply(n, c, inv_t, 0 if self.loss_type == "cross_entropy" else 1, float(self.label_smoothing))
            return loss, out8, None
        pn, pc = getattr(n, "_tt_packed", None), getattr(c, "_tt_packed", None)     # (buffer, scale) emitted by the towers
        if self.score_dtype != "bf16" or n.shape != c.shape or pn is None or pc is None or pc[1] != 1.0:
            return _ScoreCEFn.apply(n, c, inv_t, self.score_dtype, first_call, False)
        return _ScoreCEFn.apply(n, c, inv_t, self.score_dtype, first_call, False, pn[0], pc[0], pn[1])

    def _compute_similarity_matrix(self, notice_embeddings, company_embeddings):                # :99-112
        return self.two_tower_model.compute_similarity(notice_embeddings, company_embeddings, self.temperature)

    def _verify_positive_pair_alignment(self, out8: torch.Tensor):                               # :253-290
        row_hit, col_hit = float(out8[1]), float(out8[5])
        print("🔍 [Positive Pair Alignment Check]")
        print(f"   📊 Notice→Company Top-1 정확도: {row_hit:.3f}")
        print(f"   📊 Company→Notice Top-1 정확도: {col_hit:.3f}")
        print(f"   🎯 대각선이 row-max인 비율: {row_hit:.3f}")
        if row_hit < 0.05 and col_hit < 0.05:
            print("   🚨 CRITICAL: Positive pair 정합성 실패!")
            print("   → DataLoader에서 notice/company 순서가 어긋남")
            print("   → 동일한 샘플러/인덱스로 페어를 구성하세요")
        elif row_hit < 0.3 or col_hit < 0.3:
            print("   ⚠️  WARNING: Positive pair 정합성 부족")
            print("   → 배치 내 positive 비율 확인 필요")
        else:
            print("   ✅ Positive pair 정합성 양호")
        print()

    # ---- predict_batch: :181-207 ----------------------------------------------------------------------
    @_no_dynamo
    def predict_batch(self, batch, top_k: int = 10) -> Dict[str, torch.Tensor]:
        self.eval()
        with torch.no_grad():
            n, c = self.two_tower_model(batch["notice"], batch["company"])
            sim = ops.score_matrix(n.contiguous(), c.contiguous(), 1.0 / float(self.temperature))
            vals, idx = ops.topk_rows(sim, top_k)
            return {"top_similarities": vals, "top_indices": idx, "all_similarities": sim}

    def diagonal_ranks(self, batch):
        """0-based rank of each positive in its row (count of strictly larger scores, ties before the
        diagonal counted) -- the quantity Recall@K / MRR need, without the B x B matrix."""
        with torch.no_grad():
            n, c = self.two_tower_model(batch["notice"], batch["company"])
            inv_t = 1.0 / float(self.temperature)
            _, _, rank, _ = ops.score_dir_fwd(n.contiguous(), c.contiguous(), inv_t, abs(inv_t), 0, False)
            return rank


def create_two_tower_train_task(notice_categorical_keys, company_categorical_keys, metadata_path: str = "meta/metadata.csv",
                                categorical_embedding_dim: int = 64, notice_dense_input_dim: int = 256,
                                company_dense_input_dim: int = 128, tower_hidden_dims=None, final_embedding_dim: int = 128,
                                dropout_rate: float = 0.2, temperature: float = 1.0, loss_type: str = "cross_entropy",
                                device="cuda:0", embedding_grad=None, score_dtype=None, mlp_dtype=None) -> TwoTowerTrainTask:
    model = create_two_tower_model(notice_categorical_keys=notice_categorical_keys,
                                   company_categorical_keys=company_categorical_keys, metadata_path=metadata_path,
                                   categorical_embedding_dim=categorical_embedding_dim,
                                   notice_dense_input_dim=notice_dense_input_dim,
                                   company_dense_input_dim=company_dense_input_dim, tower_hidden_dims=tower_hidden_dims,
                                   final_embedding_dim=final_embedding_dim, dropout_rate=dropout_rate, device=device,
                                   embedding_grad=embedding_grad, mlp_dtype=mlp_dtype)
    return TwoTowerTrainTask(two_tower_model=model, temperature=temperature, loss_type=loss_type, score_dtype=score_dtype)
