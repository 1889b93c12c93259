"""FeatureProjector -- drop-in for src/torchrec_preprocess/feature_projector.py:4-28: two frozen 2-layer
MLPs (numeric: Linear(n,128)-ReLU-Linear(128,128); text: Linear(768,128)-ReLU-Linear(128,128), shared by
all text columns).  Same sub-module names (`num_proj.{0,2}`, `text_proj.{0,2}`) => same state-dict keys.
The forward runs tt_linear_fwd (f32 MFMA) and is inference-only, as in the reference (it is only ever
called under torch.no_grad(): feature_preprocessor.py:170, unified_bid_data_loader.py:1380-1448)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops


class FeatureProjector(nn.Module):
    def __init__(self, num_dim: int, text_dim: int, num_proj_dim: int = 128, text_proj_dim: int = 128):
        super().__init__()
        self.num_proj = nn.Sequential(nn.Linear(num_dim, num_proj_dim), nn.ReLU(), nn.Linear(num_proj_dim, num_proj_dim))
        self.text_proj = nn.Sequential(nn.Linear(text_dim, text_proj_dim), nn.ReLU(), nn.Linear(text_proj_dim, text_proj_dim))

    @staticmethod
    def _mlp(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
        x = x.to(dtype=torch.float32).contiguous()
        h = ops.linear_fwd(x, seq[0].weight.detach().contiguous(), seq[0].bias.detach(), relu=True)
        return ops.linear_fwd(h, seq[2].weight.detach().contiguous(), seq[2].bias.detach(), relu=False)

    @torch.no_grad()
    def forward(self, dense: Optional[torch.Tensor], text_dict: Dict[str, torch.Tensor]):
        dense_proj = self._mlp(self.num_proj, dense) if dense is not None else None
        text_proj = {col: self._mlp(self.text_proj, x) for col, x in text_dict.items()}
        return dense_proj, text_proj
