"""TwoTowerModel -- drop-in for src/towers/two_tower_model.py:10-166 on the HIP path.

Both towers' tables are fused into one row space at construction, so one tt_embed_lookup_fwd launch
serves both towers and one duplicate-row plan / segmented reduction serves both backward passes.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from ._lib import no_dynamo as _no_dynamo
import torch.nn as nn

from . import ops
from .cat_embed import EmbeddingStore
from .towers import CompanyTower, NoticeTower, run_towers


class TwoTowerModel(nn.Module):
    def __init__(self, notice_tower_config: Dict, company_tower_config: Dict, final_embedding_dim: int = 128,
                 device="cuda:0", check_norms: bool = False):
        super().__init__()
        self.device = device
        self.final_embedding_dim = final_embedding_dim
        # the reference checks ||emb|| == 1 with two host syncs per step (two_tower_model.py:81-87);
        # here that is an opt-in debug flag
        self.check_norms = check_norms
        self.notice_tower = NoticeTower(**notice_tower_config)
        self.company_tower = CompanyTower(**company_tower_config)
        assert notice_tower_config.get("final_embedding_dim", 128) == final_embedding_dim      # :38-39
        assert company_tower_config.get("final_embedding_dim", 128) == final_embedding_dim
        ne, ce = self.notice_tower.categorical_embedder, self.company_tower.categorical_embedder
        if ne.materialize and ce.materialize and ne.embedding_dim == ce.embedding_dim:
            EmbeddingStore.fuse([self.notice_tower.categorical_embedder.store, self.company_tower.categorical_embedder.store])
        self.to(torch.device(self.device))

    @property
    def embedding_store(self) -> EmbeddingStore:
        return self.notice_tower.categorical_embedder.store

    @_no_dynamo
    def forward(self, notice_input, company_input, return_similarity: bool = False, temperature: float = 1.0):
        notice_embeddings, company_embeddings = run_towers([self.notice_tower, self.company_tower],
                                                           [notice_input, company_input])
        if notice_embeddings.size(0) != company_embeddings.size(0):                              # :74-78
            raise ValueError(f"Notice와 Company 배치 크기가 다릅니다: {notice_embeddings.size(0)} vs {company_embeddings.size(0)}")
        if self.check_norms:
            for name, e in (("Notice", notice_embeddings), ("Company", company_embeddings)):
                n = e.detach().norm(p=2, dim=1)
                if not torch.allclose(n, torch.ones_like(n), atol=1e-4):
                    print(f"Warning: {name} embeddings are not L2 normalized")
        if return_similarity:                                                                    # :90-98
            return {"notice_embeddings": notice_embeddings, "company_embeddings": company_embeddings,
                    "similarity_matrix": self.compute_similarity(notice_embeddings, company_embeddings, temperature)}
        return notice_embeddings, company_embeddings

    def get_notice_embeddings(self, notice_input) -> torch.Tensor:
        return self.notice_tower(notice_input)

    def get_company_embeddings(self, company_input) -> torch.Tensor:
        return self.company_tower(company_input)

    def compute_similarity(self, notice_emb: torch.Tensor, company_emb: torch.Tensor, temperature: float = 1.0) -> torch.Tensor:
        return _SimilarityFn.apply(notice_emb, company_emb, 1.0 / float(temperature))


class _SimilarityFn(torch.autograd.Function):
    """Dense S = N C^T / T (torch.mm of two_tower_model.py:92,117), differentiable."""

    @staticmethod
    def forward(ctx, n, c, inv_t):
        n, c = n.contiguous().float(), c.contiguous().float()
        ctx.save_for_backward(n, c)
        ctx.inv_t = inv_t
        return ops.score_matrix(n, c, inv_t)

    @staticmethod
    def backward(ctx, dS):
        n, c = ctx.saved_tensors
        dS = dS.contiguous()
        # dN = dS C / T ; dC = dS^T N / T  -- Linear-shaped products on the same MFMA tiles
        dN = ops.linear_fwd(dS, c.t().contiguous(), None) * ctx.inv_t
        dC = ops.linear_fwd(dS.t().contiguous(), n.t().contiguous(), None) * ctx.inv_t
        return dN, dC, None


def create_two_tower_model(notice_categorical_keys: List[str], company_categorical_keys: List[str],
                           metadata_path: str = "meta/metadata.csv", categorical_embedding_dim: int = 64,
                           notice_dense_input_dim: int = 256, company_dense_input_dim: int = 128,
                           tower_hidden_dims: Optional[List[int]] = None, final_embedding_dim: int = 128,
                           dropout_rate: float = 0.2, device="cuda:0", embedding_grad: Optional[str] = None,
                           mlp_dtype: Optional[str] = None) -> TwoTowerModel:
    if tower_hidden_dims is None:
        tower_hidden_dims = [256, 128]
    common = dict(metadata_path=metadata_path, categorical_embedding_dim=categorical_embedding_dim,
                  tower_hidden_dims=tower_hidden_dims, final_embedding_dim=final_embedding_dim, dropout_rate=dropout_rate,
                  device=device, embedding_grad=embedding_grad, mlp_dtype=mlp_dtype)
    return TwoTowerModel(
        notice_tower_config=dict(categorical_keys=notice_categorical_keys, dense_input_dim=notice_dense_input_dim, **common),
        company_tower_config=dict(categorical_keys=company_categorical_keys, dense_input_dim=company_dense_input_dim, **common),
        final_embedding_dim=final_embedding_dim, device=device)
