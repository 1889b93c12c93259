"""TwoTowerEvaluator -- drop-in for src/evaluation/evaluator.py:6-283 (Recall@K, MRR, aggregate metrics).

The reference ranks with torch.topk / a full argsort plus a Python loop per row (:58-68).  Here both
metrics come from the RANK OF THE POSITIVE in its row -- tt_diag_rank_rows on a given similarity matrix,
or TwoTowerTrainTask.diagonal_ranks(batch) straight from the embeddings without the B x B matrix:
  recall@k = mean(rank < k),  MRR = mean(1 / (rank + 1)).
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
            print(f"[eval] loss {out['loss']:.4f} acc {out['accuracy']:.4f} R@5 {out['recall@5']:.4f} "
                  f"R@10 {out['recall@10']:.4f} MRR {out['mrr']:.4f}")
        return out

    @torch.no_grad()
    def evaluate_comprehensive(self, model, data_loader, max_batches: int = None, verbose: bool = True) -> Dict[str, float]:   # :157-209
        agg, n = {}, 0
        for i, batch in enumerate(data_loader):
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
            print(f"[eval] {n} batches: loss {out['loss']:.4f} acc {out['accuracy']:.4f} R@5 {out['recall@5']:.4f} "
                  f"R@10 {out['recall@10']:.4f} MRR {out['mrr']:.4f}")
        return out
