"""Minimal id container with the slice of the torchrec.KeyedJaggedTensor interface the reference's
hot path uses (src/towers/cat_embed.py:92-94, src/towers/tower/base_tower.py:130,
scripts/train.py:269-272): keys() / values() / lengths() / to() / device() / pin_memory().

values(): int64 [B*K], SAMPLE-major (categorical_data.flatten() of a [B,K] matrix --
src/towers/pairs/unified_bid_data_loader.py:834); lengths(): ones [B*K] (:835).
A real torchrec KJT is accepted everywhere this class is: only these methods are called.
"""
from __future__ import annotations

from typing import List

import torch


class KeyedJaggedTensor:
    def __init__(self, keys: List[str], values: torch.Tensor, lengths: torch.Tensor | None = None):
        self._keys = list(keys)
        self._values = values
        self._lengths = lengths

    @classmethod
    def from_lengths_sync(cls, keys, values, lengths):
        return cls(keys, values, lengths)

    def keys(self) -> List[str]:
        return self._keys

    def values(self) -> torch.Tensor:
        return self._values

    def lengths(self) -> torch.Tensor:
        if self._lengths is None:      # all bags have length 1 on this path
            self._lengths = torch.ones(self._values.numel(), dtype=torch.long, device=self._values.device)
        return self._lengths

    def to(self, device, non_blocking: bool = False) -> "KeyedJaggedTensor":
        lengths = None if self._lengths is None else self._lengths.to(device, non_blocking=non_blocking)
        return KeyedJaggedTensor(self._keys, self._values.to(device, non_blocking=non_blocking), lengths)

    def device(self) -> torch.device:
        return self._values.device

    def pin_memory(self) -> "KeyedJaggedTensor":
        lengths = None if self._lengths is None else self._lengths.pin_memory()
        return KeyedJaggedTensor(self._keys, self._values.pin_memory(), lengths)

    def __repr__(self):
        return f"KeyedJaggedTensor(keys={len(self._keys)}, values={tuple(self._values.shape)}, device={self._values.device})"


def build_batch_kjt(categorical_data: torch.Tensor, categorical_keys: List[str]) -> KeyedJaggedTensor:
    """[B,K] ids -> KJT, the wire format of _build_batch_kjt (unified_bid_data_loader.py:827-841)."""
    b, k = categorical_data.shape
    values = categorical_data.reshape(-1).to(torch.long)
    return KeyedJaggedTensor(categorical_keys, values,
                             torch.ones(b * k, dtype=torch.long, device=categorical_data.device))
