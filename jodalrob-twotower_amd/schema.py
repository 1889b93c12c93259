"""Column schema of the two feature tables, derived from the metadata CSV exactly as the reference
does (src/torchrec_preprocess/schema.py:14-88 + data/column_classifier.py:67-130): use == Y rows
only; PK columns set aside; bigint / double precision / numeric / integer -> numeric; text-like or
character(1) columns -> categorical if flagged Y, else text; everything else ignored.
"""
from __future__ import annotations

import csv
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional

_ALIASES = {
    "table": ["테이블명", "table", "TABLE"],
    "column": ["컬럼명", "컬럼", "column", "COLUMN", "필드명"],
    "use": ["사용 여부", "사용여부", "use", "USE"],
    "pk": ["PK", "pk", "Pk"],
    "dtype": ["타입", "데이터타입", "type", "TYPE", "data_type"],
    "is_categorical": ["범주형 여부", "범주형여부", "categorical", "IS_CATEGORICAL"],
    "n_categories": ["범주 갯수"],
}
_NUMERIC = {"bigint", "double precision", "numeric", "integer"}


@dataclass
class SideSchema:
    table: str
    pk_cols: List[str]
    numeric: List[str] = field(default_factory=list)
    categorical: List[str] = field(default_factory=list)
    text: List[str] = field(default_factory=list)
    text_embed_prefix: Optional[str] = None
    text_embed_dims: Optional[int] = 768


@dataclass
class PairSchema:
    table: str = "bid_two_tower"
    notice_id_cols: List[str] = field(default_factory=list)
    company_id_cols: List[str] = field(default_factory=list)


@dataclass
class TorchRecSchema:
    notice: SideSchema
    company: SideSchema
    pair: PairSchema


def _header(fieldnames, what: str) -> str:
    names = list(fieldnames)
    for cand in _ALIASES[what]:
        if cand in names:
            return cand
    loose = {re.sub(r"\s+", "", n).lower(): n for n in names}
    for cand in _ALIASES[what]:
        key = re.sub(r"\s+", "", cand).lower()
        if key in loose:
            return loose[key]
    raise KeyError(f"metadata CSV lacks a '{what}' column (tried {_ALIASES[what]})")


def _yes(v) -> bool:
    return v is not None and str(v).strip().lower() in {"y", "yes", "true", "1", "t"}


def read_metadata(path) -> List[Dict[str, str]]:
    with open(Path(path), newline="", encoding="utf-8-sig") as f:
        return list(csv.DictReader(f))


def classify_columns(table_name: str, metadata_path="meta/metadata.csv") -> Dict[str, List[str]]:
    rows = read_metadata(metadata_path)
    if not rows:
        return {"total": 0, "pk": [], "numeric": [], "categorical": [], "text": []}
    h = {k: _header(rows[0].keys(), k) for k in ("table", "column", "use", "pk", "dtype", "is_categorical")}
    mine = [r for r in rows if str(r[h["table"]]).strip() == str(table_name).strip() and _yes(r[h["use"]])]
    out = {"total": len(mine), "pk": [], "numeric": [], "categorical": [], "text": []}
    for r in mine:
        name = str(r[h["column"]]).strip()
        if _yes(r[h["pk"]]):
            out["pk"].append(name)
            continue
        dt = str(r[h["dtype"]]).strip().lower()
        if dt in _NUMERIC:
            out["numeric"].append(name)
        elif dt.startswith("text") or dt == "varchar" or dt.startswith("character varying") or \
                re.fullmatch(r"(character|char|character varying|varchar)\s*\(\s*1\s*\)", dt):
            out["categorical" if _yes(r[h["is_categorical"]]) else "text"].append(name)
    return out


def category_counts(table_name: str, metadata_path) -> Dict[str, Optional[int]]:
    """column -> category count (None where the CSV cell is empty) for one table."""
    rows = read_metadata(metadata_path)
    if not rows:
        return {}
    ht, hc, hn = _header(rows[0].keys(), "table"), _header(rows[0].keys(), "column"), _header(rows[0].keys(), "n_categories")
    out: Dict[str, Optional[int]] = {}
    for r in rows:
        if str(r[ht]) == table_name:                       # reference compares without stripping (cat_embed.py:59)
            name = str(r[hc])
            if name not in out:
                cell = (r[hn] or "").strip()
                out[name] = int(float(cell)) if cell else None
    return out


def build_side_schema_from_meta(table_name: str, metadata_path="meta/metadata.csv") -> SideSchema:
    md = classify_columns(table_name, metadata_path)
    pk = set(md["pk"])
    return SideSchema(table=table_name, pk_cols=md["pk"],
                      numeric=[c for c in md["numeric"] if c not in pk],
                      categorical=[c for c in md["categorical"] if c not in pk],
                      text=[c for c in md["text"] if c not in pk],
                      text_embed_prefix=None, text_embed_dims=None)


def build_torchrec_schema_from_meta(*, notice_table: str, company_table: str, pair_table: str,
                                    pair_notice_id_cols: List[str], pair_company_id_cols: List[str],
                                    metadata_path="meta/metadata.csv") -> TorchRecSchema:
    notice = build_side_schema_from_meta(notice_table, metadata_path)
    company = build_side_schema_from_meta(company_table, metadata_path)
    return TorchRecSchema(
        notice=notice, company=company,
        pair=PairSchema(table=pair_table,
                        notice_id_cols=pair_notice_id_cols if pair_notice_id_cols is not None else notice.pk_cols,
                        company_id_cols=pair_company_id_cols if pair_company_id_cols is not None else company.pk_cols))
