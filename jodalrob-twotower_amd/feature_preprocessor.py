"""FeaturePreprocessor -- drop-in for src/torchrec_preprocess/feature_preprocessor.py:15-268.

Same constructor and methods; `db_engine` is any object with `build_feature_store(table, side_schema,
chunksize=, limit=)` returning the reference's store dict {ids, numeric f32 [N,n], categorical i64 [N,K],
text {col: f32 [N,768]}, categorical_keys} (feature_store.py:148-153).  The reference's SQLAlchemy engine
path needs PostgreSQL and is out of scope; jodalrob_twotower_amd.synthetic.SyntheticSource is the
in-memory stand-in.  The projection itself (the arithmetic) runs on the MI355X in chunks of `batch_size`
rows and lands, as in the reference, in store['dense_projected'] on the host.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .feature_projector import FeatureProjector
from .schema import SideSchema, TorchRecSchema


class FeaturePreprocessor:
    def __init__(self, schema: TorchRecSchema, device: str = "cuda:0", num_proj_dim: int = 128, text_proj_dim: int = 128,
                 batch_size: int = 1024):
        self.schema = schema
        self.device = torch.device(device)
        self.num_proj_dim, self.text_proj_dim, self.batch_size = num_proj_dim, text_proj_dim, batch_size
        self._setup_projectors()

    def _setup_projectors(self):                                                    # :41-61
        self.projectors = {}
        for side in ("notice", "company"):
            if hasattr(self.schema, side):
                self.projectors[side] = FeatureProjector(num_dim=len(getattr(self.schema, side).numeric), text_dim=768,
                                                         num_proj_dim=self.num_proj_dim,
                                                         text_proj_dim=self.text_proj_dim).to(self.device)

    def preprocess_all(self, db_engine, feature_chunksize: int = 5000, feature_limit: Optional[int] = None,
                       show_progress: bool = True) -> Dict[str, Dict]:                # :63-107
        out = {}
        for side in ("notice", "company"):
            if side in self.projectors:
                out[side] = self._preprocess_tower(db_engine, getattr(self.schema, side), side, feature_chunksize, feature_limit)
        return out

    def _preprocess_tower(self, db_engine, tower_schema: SideSchema, tower_name: str, chunksize: int, limit: Optional[int]):
        store = db_engine.build_feature_store(tower_schema.table, tower_schema, chunksize=chunksize, limit=limit)
        store = self._apply_projection(store, self.projectors[tower_name], tower_schema, tower_name)
        store["categorical_keys"] = list(tower_schema.categorical)
        return store

    def _apply_projection(self, store: Dict, projector: FeatureProjector, tower_schema: SideSchema, tower_name: str) -> Dict:
        """dense_projected = cat[num_proj(numeric) | text_proj(text[col]) for col in schema.text]   (:150-233)"""
        np_numeric = store.get("numeric")
        np_text = store.get("text") or {}
        n = len(np_numeric) if np_numeric is not None else (len(next(iter(np_text.values()))) if np_text else 0)
        text_cols = [c for c in (tower_schema.text or list(np_text)) if c in np_text]
        parts_all = []
        for lo in range(0, n, self.batch_size):
            hi = min(lo + self.batch_size, n)
            num = torch.from_numpy(np.ascontiguousarray(np_numeric[lo:hi])).float().to(self.device) if np_numeric is not None else None
            txt = {c: torch.from_numpy(np.ascontiguousarray(np_text[c][lo:hi])).float().to(self.device) for c in text_cols}
            pn, pt = projector(num, txt)
            if num is not None and txt:
                parts = [pn] + [pt[c] for c in text_cols]
            elif num is not None:
                parts = [pn]
            else:
                parts = [pt[c] for c in text_cols]
            if parts:
                parts_all.append(torch.cat(parts, dim=1).cpu().numpy())
        result = dict(store)
        result["dense_projected"] = np.concatenate(parts_all, axis=0) if parts_all else None
        return result

    def build_id_mappings(self, stores: Dict[str, Dict]) -> Tuple[Dict, Dict]:           # :235-268
        n2i, c2i = {}, {}
        if "notice" in stores:
            n2i = {tuple(p): i for i, p in enumerate(stores["notice"].get("ids", []))}
        if "company" in stores:
            cids = stores["company"].get("ids", [])
            if len(cids):
                if isinstance(cids[0], (tuple, list)):
                    c2i = {str(t[0]): i for i, t in enumerate(cids)}
                else:
                    c2i = {str(c): i for i, c in enumerate(cids)}
        return n2i, c2i
