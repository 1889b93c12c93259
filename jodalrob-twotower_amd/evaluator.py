"""TwoTowerEvaluator -- drop-in for src/evaluation/evaluator.py:6-283 (Recall@K, MRR, aggregate metrics, the
result printers and the prediction demo that scripts/train.py:444-452 calls before its final checkpoint).

The reference ranks with torch.topk / a full argsort plus a Python loop per row (:58-68).  Here both
metrics come from the RANK OF THE POSITIVE in its row -- tt_diag_rank_rows on a given similarity matrix,
or TwoTowerTrainTask.diagonal_ranks(batch) straight from the embeddings without the B x B matrix:
  recall@k = mean(rank < k),  MRR = mean(1 / (rank + 1)).
Every method the reference's callers touch is pinned by tests/golden/api_surface.json (tests/test_api_surface.py).
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops


class TwoTowerEvaluator:
    def __init__(self, device: str = "cuda:0"):
        self.device = device

    @staticmethod
    def _ranks(similarity_matrix: torch.Tensor) -> torch.Tensor:
        s = similarity_matrix.detach().float()
        return ops.diag_rank_rows(s if s.stride(1) == 1 else s.contiguous())

    def compute_recall_at_k(self, similarity_matrix: torch.Tensor, k: int) -> torch.Tensor:        # :20-43
        k = min(k, similarity_matrix.size(1))
        return (self._ranks(similarity_matrix) < k).float().mean()

    def compute_mrr(self, similarity_matrix: torch.Tensor) -> torch.Tensor:                         # :45-71
        return (1.0 / (self._ranks(similarity_matrix).float() + 1.0)).mean()

    def metrics_from_ranks(self, ranks: torch.Tensor, basic: Dict) -> Dict[str, float]:
        b = ranks.numel()
        r5, r10 = (ranks < min(5, b)).float().mean().item(), (ranks < min(10, b)).float().mean().item()
        mrr = (1.0 / (ranks.float() + 1.0)).mean().item()
        ra, rr5, rr10 = 1.0 / b, min(5.0 / b, 1.0), min(10.0 / b, 1.0)
        g = lambda k: float(basic.get(k, 0.0))
        return {"loss": g("loss"), "accuracy": g("accuracy"), "similarity_gap": g("similarity_gap"),
                "positive_similarity_mean": g("positive_similarity_mean"), "negative_similarity_mean": g("negative_similarity_mean"),
                "recall@5": r5, "recall@10": r10, "mrr": mrr, "batch_size": b, "random_accuracy": ra, "random_recall@5": rr5,
                "random_recall@10": rr10, "accuracy_improvement": g("accuracy") > ra, "recall@5_improvement": r5 > rr5,
                "recall@10_improvement": r10 > rr10}

    def compute_comprehensive_metrics(self, similarity_matrix: torch.Tensor, basic_metrics: Dict) -> Dict[str, float]:   # :73-121
        return self.metrics_from_ranks(self._ranks(similarity_matrix), basic_metrics)

    @torch.no_grad()
    def evaluate_single_batch(self, model, batch: Dict, verbose: bool = True) -> Dict[str, float]:  # :123-155
        model.eval()
        if hasattr(model, "forward_with_ranks"):                       # one pass through the towers serves loss, metrics and ranks
            res, ranks = model.forward_with_ranks(batch)
        else:
            res = model(batch, return_metrics=True)
            ranks = self._ranks(res["similarity_matrix"])
        basic = {k: (v.item() if torch.is_tensor(v) and v.numel() == 1 else v) for k, v in dict.items(res) if k != "similarity_matrix"}
        out = self.metrics_from_ranks(ranks, basic)
        if verbose:
            self.print_single_batch_results(out)
        return out

    @torch.no_grad()
    def evaluate_comprehensive(self, model, dataloader, verbose: bool = True, max_batches: int = None) -> Dict[str, float]:   # :157-209
        """Mean of the per-batch metrics over `dataloader` (every key of the reference's result -- loss, accuracy, recall@5,
        recall@10, mrr, similarity_gap, num_batches -- plus the means of the other per-batch figures).  `max_batches` (not in
        the reference) stops early; `num_batches` is the number of batches evaluated."""
        fast = self._fast_eval(model, dataloader)
        if fast is not None:
            fast.reset()
            n = dataloader.eval_batches(fast, lambda b: self.evaluate_single_batch(model, b, verbose=False), max_batches)
            out = fast.means() if n else {}
            if n:                                            # the figures every batch shares (evaluate_single_batch's other keys)
                b = dataloader.batch_size
                out.update({"batch_size": float(b), "random_accuracy": 1.0 / b, "random_recall@5": min(5.0 / b, 1.0),
                            "random_recall@10": min(10.0 / b, 1.0)})
            out["num_batches"] = n
            model.eval()
            if verbose and n:
                self.print_comprehensive_results(out)
            return out
        agg, n = {}, 0
        for i, batch in enumerate(dataloader):
            if max_batches is not None and i >= max_batches:
                break
            m = self.evaluate_single_batch(model, batch, verbose=False)
            for k, v in m.items():
                if isinstance(v, (int, float)) and not isinstance(v, bool):
                    agg[k] = agg.get(k, 0.0) + float(v)
            n += 1
        out = {k: v / max(n, 1) for k, v in agg.items()}
        out["num_batches"] = n
        if verbose and n:
            self.print_comprehensive_results(out)
        return out

    def _fast_eval(self, model, dataloader):
        """A GraphedEvalStep for (model, loader batch size) when the loader is a device-resident DevicePairLoader with at least one
        full batch and the model is this package's task on a GPU; cached on the evaluator.  None: the batch-by-batch loop."""
        if not (hasattr(dataloader, "eval_batches") and hasattr(model, "forward_with_ranks") and getattr(model, "exchange", None) is None):
            return None
        B = dataloader.batch_size
        if dataloader.pairs.shape[0] < B or dataloader.shuffle or not dataloader.pairs.is_cuda:
            return None
        key = (id(model), B)
        cache = self.__dict__.setdefault("_eval_graphs", {})
        if key not in cache:
            from .graph import GraphedEvalStep
            cache[key] = GraphedEvalStep(model, dataloader.batch(None, 0))
        return cache[key]

    def close(self):
        """Drops the captured evaluation graphs (and their private memory pools)."""
        for g in self.__dict__.pop("_eval_graphs", {}).values():
            g.close()

    # ---- printers (:211-267): same lines on stdout as the reference, driven by tables ---------------------------------
    _ROWS_SINGLE = (("Loss", "loss", 4), ("Top-1 Accuracy", "accuracy", 3), ("Recall@5", "recall@5", 3), ("Recall@10", "recall@10", 3),
                    ("MRR", "mrr", 3), ("Similarity Gap", "similarity_gap", 3), ("Positive Similarity (평균)", "positive_similarity_mean", 3),
                    ("Negative Similarity (평균)", "negative_similarity_mean", 3))
    _ROWS_BASELINE = (("Top-1 정확도", "random_accuracy", "accuracy", "accuracy_improvement"),
                      ("Recall@5", "random_recall@5", "recall@5", "recall@5_improvement"),
                      ("Recall@10", "random_recall@10", "recall@10", "recall@10_improvement"))
    _ROWS_MEAN = _ROWS_SINGLE[:6]
    # (metric, label, ((threshold, verdict), ...), verdict below every threshold)
    _GRADES = (("accuracy", "Top-1 정확도", ((0.3, "우수 (0.3 이상)"), (0.15, "보통 (0.15~0.3)")), "미흡 (0.15 미만)"),
               ("recall@10", "Recall@10", ((0.6, "실용적 수준 (0.6 이상)"), (0.4, "개선 필요 (0.4~0.6)")), "부족 (0.4 미만)"),
               ("similarity_gap", "유사도 구분", ((0.5, "양호 (0.5 이상)"),), "개선 필요 (0.5 미만)"))

    def print_single_batch_results(self, metrics: Dict[str, float]):                                 # :211-227
        print(f"배치 크기: {int(metrics['batch_size'])}")
        for label, key, digits in self._ROWS_SINGLE:
            print(f"{label}: {metrics[key]:.{digits}f}")
        print("\n--- 랜덤 기준선과 비교 ---")
        for label, rand, cur, better in self._ROWS_BASELINE:
            print(f"랜덤 {label}: {metrics[rand]:.3f} | 현재: {metrics[cur]:.3f} ({'개선' if metrics[better] else '미흡'})")

    def print_comprehensive_results(self, metrics: Dict[str, float]):                                # :229-240
        print(f"테스트 배치 수: {int(metrics['num_batches'])}")
        for label, key, digits in self._ROWS_MEAN:
            print(f"평균 {label}: {metrics[key]:.{digits}f}")
        print("\n--- 성능 평가 ---")
        self.print_performance_assessment(metrics)

    def print_performance_assessment(self, metrics: Dict[str, float]):                               # :242-267
        for key, label, steps, floor in self._GRADES:
            verdict = next((text for bound, text in steps if metrics[key] > bound), floor)
            print(f"{label}: {verdict}")

    def demonstrate_predictions(self, model, batch: Dict, top_k: int = 10):                          # :269-283
        """Top-k retrieval demo on one batch (model.predict_batch -> tt_score_matrix + tt_topk_rows); called by the reference
        driver right before its final checkpoint (scripts/train.py:452)."""
        model.eval()
        with torch.no_grad():
            pred = model.predict_batch(batch, top_k=top_k)
            sim = pred["all_similarities"]
            print("--- 추론 예제 ---")
            print(f"배치 크기: {batch['notice']['dense'].shape[0]}")
            print(f"첫 번째 공고의 Top-5 유사도: {pred['top_similarities'][0][:5]}")
            print(f"첫 번째 공고의 Top-5 업체 인덱스: {pred['top_indices'][0][:5]}")
            print(f"유사도 행렬 크기: {sim.shape}")
            print(f"대각선 유사도 (positive pairs): {torch.diag(sim)[:5]}")
            print(f"첫 번째 행 유사도 범위: {sim[0].min():.3f} ~ {sim[0].max():.3f}")
