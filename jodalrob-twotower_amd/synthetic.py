"""Synthetic notice/company workloads (the reference's data source is a PostgreSQL database that
does not exist here: data/database_connector.py, src/torchrec_preprocess/feature_store.py -- out of
scope).  Shapes follow SURVEY.md §8(d): real key lists (32 notice + 6 company categorical keys), per-key
vocabularies scaled so that each tower's tables sum to a requested number of rows.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import torch

from .kjt import KeyedJaggedTensor

META_HEADER = "테이블명,컬럼명,타입,사용 여부,NULL 전략,범주형 여부,범주 갯수,NULL 갯수,길이,PK,NN,국문 설명,비고"


def scale_vocabs(vocabs: Sequence[int], total_rows: int) -> List[int]:
    """V_k' = max(2, round(V_k * f)), f = total/sum(V); the largest key absorbs the remainder."""
    f = total_rows / float(sum(vocabs))
    out = [max(2, int(round(v * f))) for v in vocabs]
    big = max(range(len(out)), key=lambda i: out[i])
    out[big] += total_rows - sum(out)
    if out[big] < 2:
        raise ValueError("total_rows too small for this key list")
    return out


def write_metadata(path, tables: Dict[str, Dict[str, int]], safety_margin: int = 10) -> Path:
    """Metadata CSV (reference column headers) whose category counts give exactly the requested vocab
    sizes under the reference rule vocab = count + 10 (src/towers/cat_embed.py:76)."""
    lines = [META_HEADER]
    for table, cols in tables.items():
        for col, vocab in cols.items():
            lines.append(f"{table},{col},text,Y,,Y,{vocab - safety_margin},0,,,,,")
    path = Path(path)
    path.write_text("\n".join(lines) + "\n", encoding="utf-8")
    return path


def load_real_schema(path) -> dict:
    return json.loads(Path(path).read_text())


def make_batch(B: int, vocab_n: Sequence[int], vocab_c: Sequence[int], keys_n, keys_c, din_n: int, din_c: int, device,
               seed: int, zipf_alpha: Optional[float] = None) -> dict:
    """ids ~ U[0, V_k) per key (or Zipf(alpha) ranks mapped through a fixed multiplicative hash), dense
    ~ N(0,1); generated on the device with a seeded generator (plumbing only)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def ids_for(vocabs):
        cols = []
        for v in vocabs:
            if zipf_alpha is None:
                cols.append(torch.randint(0, v, (B,), generator=g, device=device, dtype=torch.int64))
            else:
                # inverse-CDF sampling of a truncated Zipf(alpha) rank, then a fixed permutation-like hash
                u = torch.rand(B, generator=g, device=device, dtype=torch.float64)
                a = 1.0 - zipf_alpha
                rank = ((u * (float(v) ** a - 1.0) + 1.0) ** (1.0 / a)).floor().clamp_(1, v).to(torch.int64) - 1
                cols.append((rank * 2654435761 + 12345) % v)
        return torch.stack(cols, dim=1).reshape(-1).contiguous()

    def dense(d):
        return torch.randn((B, d), generator=g, device=device, dtype=torch.float32)

    return {"notice": {"dense": dense(din_n), "kjt": KeyedJaggedTensor(list(keys_n), ids_for(vocab_n))},
            "company": {"dense": dense(din_c), "kjt": KeyedJaggedTensor(list(keys_c), ids_for(vocab_c))}}


class SyntheticSource:
    """In-memory stand-in for the PostgreSQL engine of the reference (data/database_connector.py -- out of
    scope): produces feature stores in the layout of src/torchrec_preprocess/feature_store.py:148-153 and
    positive (notice, company) pairs."""

    def __init__(self, n_notice: int, n_company: int, n_pairs: int, vocab_notice, vocab_company, seed: int = 0,
                 text_dim: int = 768):
        self.n = {"notice": n_notice, "company": n_company}
        self.vocab = {"notice": list(vocab_notice), "company": list(vocab_company)}
        self.n_pairs, self.seed, self.text_dim = n_pairs, seed, text_dim

    def build_feature_store(self, table: str, side_schema, chunksize: int = 5000, limit=None):
        import numpy as np
        n = self.n[table] if limit is None else min(self.n[table], limit)
        rng = np.random.default_rng([self.seed, 1 if table == "notice" else 2])
        vocab = self.vocab[table]
        if len(vocab) != len(side_schema.categorical):
            raise ValueError(f"{table}: {len(side_schema.categorical)} categorical keys but {len(vocab)} vocab sizes")
        if table == "notice":
            ids = [(f"N{i:09d}", "00") for i in range(n)]
        else:
            ids = [f"{1000000000 + i}" for i in range(n)]
        return {
            "ids": ids,
            "numeric": rng.standard_normal((n, len(side_schema.numeric))).astype(np.float32) if side_schema.numeric else None,
            "categorical": np.stack([rng.integers(0, v, n) for v in vocab], axis=1).astype(np.int64),
            "text": {c: rng.standard_normal((n, self.text_dim)).astype(np.float32) for c in (side_schema.text or [])},
            "categorical_keys": list(side_schema.categorical),
        }

    def load_pairs(self, pair_schema, limit=None):
        import numpy as np
        rng = np.random.default_rng([self.seed, 3])
        n = self.n_pairs if limit is None else min(self.n_pairs, limit)
        ni, ci = rng.integers(0, self.n["notice"], n), rng.integers(0, self.n["company"], n)
        return [((f"N{a:09d}", "00"), f"{1000000000 + b}") for a, b in zip(ni, ci)]
