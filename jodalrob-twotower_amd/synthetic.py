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
