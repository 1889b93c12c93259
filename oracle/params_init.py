"""Deterministic, torch-RNG-independent parameter and batch recipes shared by the golden
generator (oracle/gen_golden.py) and the tests.  TEST INFRASTRUCTURE ONLY.

State-dict key conventions follow the reference (SURVEY.md §8c "facts verified"):
  two_tower_model.{notice,company}_tower.categorical_embedder.embeddings.<key>.weight [V_k+10, E]
  ....dense_projection.{weight,bias}; ....mlp.{0,4,..}.{weight,bias}; ....mlp.{2,6,..}.{weight,bias,
  running_mean,running_var,num_batches_tracked}
"""
from __future__ import annotations

import zlib

import numpy as np


def _rng_for(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode("utf-8"))])


def init_state_numpy(shapes: dict, seed: int) -> dict:
    """shapes: {state_dict key: shape tuple}.  Every entry gets its own stream keyed by name."""
    out = {}
    for name, shape in shapes.items():
        r = _rng_for(seed, name)
        shape = tuple(shape)
        if name.endswith("num_batches_tracked"):
            out[name] = np.asarray(0, dtype=np.int64)
        elif name.endswith("running_mean"):
            out[name] = (0.1 * r.standard_normal(shape)).astype(np.float32)
        elif name.endswith("running_var"):
            out[name] = (1.0 + 0.25 * r.random(shape)).astype(np.float32)
        elif "embeddings." in name:
            out[name] = r.standard_normal(shape).astype(np.float32)            # nn.Embedding init N(0,1)
        elif len(shape) == 2:
            out[name] = (r.standard_normal(shape) / np.sqrt(shape[1])).astype(np.float32)
        elif name.endswith("bias"):
            out[name] = (0.1 * r.standard_normal(shape)).astype(np.float32)
        else:                                                                    # BatchNorm weight
            out[name] = (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
    return out


def synth_batch_numpy(B, vocab_n, vocab_c, din_n, din_c, seed, oob=False) -> dict:
    """ids i64 [B,K] sample-major (a2 wire format after flatten), dense f32.  With oob=True a few ids
    are pushed below 0 / above V-1 to pin the clamp (src/towers/cat_embed.py:114-117)."""
    r = np.random.default_rng(seed)
    ids_n = np.stack([r.integers(0, v, B) for v in vocab_n], axis=1).astype(np.int64)
    ids_c = np.stack([r.integers(0, v, B) for v in vocab_c], axis=1).astype(np.int64)
    if oob:
        ids_n[0, 0] = -5
        ids_n[1 % B, -1] = vocab_n[-1] + 12345
        ids_n[2 % B, 1 % len(vocab_n)] = vocab_n[1 % len(vocab_n)]          # exactly V -> V-1
        ids_c[3 % B, 0] = -1
        ids_c[4 % B, -1] = 2 ** 40
    # duplicates inside the batch so the scatter-add path sums rows
    if B >= 8:
        ids_n[5] = ids_n[6]
        ids_c[5] = ids_c[7]
    return {
        "notice_ids": ids_n, "company_ids": ids_c,
        "notice_dense": r.standard_normal((B, din_n)).astype(np.float32),
        "company_dense": r.standard_normal((B, din_c)).astype(np.float32),
    }
