"""Process-wide defaults of the drop-in, read from the environment ONCE when the package is imported.

The reference's driver builds its task through `create_two_tower_train_task(...)` with the reference's own arguments
(scripts/train.py:207-220), which have no notion of operand precision or gradient layout.  A deployment that runs that driver
unchanged selects the fast path with three environment variables -- nothing else in the package or the library reads the
environment:

    TT_SCORE_DTYPE      fp32 (default: exact-f32 parity path) | bf16 | fp8     score-matrix operands
    TT_MLP_DTYPE        fp32 (default) | bf16                                   tower Linear operands
    TT_EMBEDDING_GRAD   dense (default: `.grad` of every table, as nn.Embedding) | sparse (row lists for FusedAdam)
    TT_SYNC_DEBUG       1: synchronise after every C-ABI call (localises an asynchronous fault to its entry point)

Every explicit constructor argument (`score_dtype=`, `mlp_dtype=`, `embedding_grad=`) wins over these.  The remaining
fields are structural switches that exist so tests can compare a fused launch with the separate launches it replaced,
bit for bit; tests set them on `settings` directly (`monkeypatch.setattr(config.settings, ...)`), objects read them when
they are constructed.
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass
class Settings:
    score_dtype: str = "fp32"
    mlp_dtype: str = "fp32"
    embedding_grad: str = "dense"
    sync_debug: bool = False
    # structural switches (tests only; no environment variable)
    tower_io_dtype: str = "x"            # bf16 towers: which of the MLP input `x` / its gradient are stored as bf16: x | both | none
    tower_unfused_tail: bool = False     # BN / output Linear / L2-normalise tail as the separate kernels
    tower_unfused_front: bool = False    # projection GEMM, block GEMM and slab / statistics pass as separate launches
    tower_unfused_back: bool = False     # first-block / projection gradient GEMMs as separate launches
    tower_pack: bool = True              # the towers' fused tail emits the score kernels' operand images
    graph_ingest: bool = True            # GraphedTrainStep hands the batch over with tt_batch_ingest (key-major rows)
    graph_weight_shadows: bool = True    # ... and refreshes bf16 shadows of the towers' projection / block weights, which the one-launch front reads
    graph_ingest_rows: bool = True       # ... which also leaves the rows in slot order: the captured lookup reads those (tt_embed_lookup_rows_fwd)
    graph_ingest_lookup: bool = False    # the hand-over launch ALSO does the lookup (tt_batch_ingest_lookup): one launch fewer, measured neutral
    #                                      (both halves are bandwidth-bound: 0.2332 vs 0.2339 ms per step, profiles/NOTES.md round 4) -- off
    grad_planned: bool = True            # the keyed duplicate-row plan also prepares embed_grad's long-row list (one launch fewer)

    @classmethod
    def from_env(cls) -> "Settings":
        e = os.environ
        return cls(score_dtype=e.get("TT_SCORE_DTYPE", "fp32"), mlp_dtype=e.get("TT_MLP_DTYPE", "fp32"),
                   embedding_grad=e.get("TT_EMBEDDING_GRAD", "dense"), sync_debug=e.get("TT_SYNC_DEBUG", "0") not in ("", "0"))


settings = Settings.from_env()
