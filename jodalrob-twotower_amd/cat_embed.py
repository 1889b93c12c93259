"""CategoricalEmbedder on one fused HBM-resident table.

Drop-in for the reference class of the same name (src/towers/cat_embed.py:11-190): same constructor,
same vocab rule (metadata category count + 10, 1000 when unknown: :50-85), same forward contract
(kjt -> dict of [B,E] or concatenated [B,K*E]; clamp to [0,V-1]; kjt None -> zeros of batch 1), same
state-dict keys (`embeddings.<key>.weight`, shape [V_k, E]).

Layout: all per-key tables of one store are rows of ONE [R, E] f32 tensor (key k owns rows
[off_k, off_k+V_k)); the per-key nn.Parameters are views into it, so optimisers, state_dict and
checkpoints see the reference's tensors while the lookup / gradient kernels see one row space.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Union

import torch
from ._lib import no_dynamo as _no_dynamo
import torch.nn as nn

from . import ops
from .config import settings
from .schema import category_counts


class EmbeddingStore:
    """One fused row space [R, E] (+ its gradient / optimiser state), shared by >= 1 embedders."""

    def __init__(self, E: int, device, grad_mode: str = "dense"):
        if grad_mode not in ("dense", "sparse"):
            raise ValueError(f"embedding_grad must be 'dense' or 'sparse', got {grad_mode!r}")
        self.E = E
        self.device = torch.device(device)
        self.grad_mode = grad_mode
        self.weight: Optional[torch.Tensor] = None      # [R, E] f32
        self.grad: Optional[torch.Tensor] = None        # dense mode: [R, E]
        self.sparse_grad = None                         # sparse mode: (DedupPlan, grad_rows [M, E])
        self.defer_long_finish = False                  # GraphedTrainStep: the optimiser's launch finishes the long rows
        self.members: List["CategoricalEmbedder"] = []
        self.version = 0
        self._grad_counters = None                      # 4 int32 words the gradient reduction keeps zero between calls
        # set by GraphedTrainStep: (static id tensors per side, their _version at the last hand-over, rows_km[, x buffers]) -- the
        # fused rows of exactly those id tensors in key-major order, written by ops.batch_ingest.  A pass over other tensors, or
        # over these after anybody else wrote to them, does not see it (rows_km_for).  With the fused hand-over + lookup launch the
        # fourth entry holds, per side, the tower input x whose embedding columns that launch has ALREADY filled (x_for): valid for
        # one pass (`ingest_x_fresh`), because an optimiser step in between changes the table rows they were copied from.
        self.ingest = None
        self.ingest_x_fresh = False

    def rows_km_for(self, id_tensors) -> Optional[torch.Tensor]:
        """The key-major rows of ops.batch_ingest if they describe exactly `id_tensors` as they are now, else None."""
        reg = self.ingest
        if reg is None or len(reg[0]) != len(id_tensors):
            return None
        for t, r, v in zip(id_tensors, reg[0], reg[1]):
            if t.data_ptr() != r.data_ptr() or t.numel() != r.numel() or r._version != v:
                return None
        return reg[2]

    def rows_sm_for(self, id_tensors) -> Optional[torch.Tensor]:
        """The slot-order fused rows the hand-over launch left for exactly `id_tensors` (as they are now), or None."""
        reg = self.ingest
        if reg is None or len(reg) < 5 or reg[4] is None or self.rows_km_for(id_tensors) is None:
            return None
        return reg[4]

    def x_for(self, id_tensors, shapes) -> Optional[list]:
        """The towers' input buffers whose embedding columns the hand-over launch has already filled for exactly `id_tensors` (as
        they are now), or None: then the lookup runs as its own launch.  shapes: per side (B, x_width, dtype).  One pass only."""
        reg = self.ingest
        if reg is None or len(reg) < 4 or reg[3] is None or not self.ingest_x_fresh or self.rows_km_for(id_tensors) is None:
            return None
        xs = reg[3]
        if len(xs) != len(shapes) or any(tuple(x.shape) != (b, w) or x.dtype != dt for x, (b, w, dt) in zip(xs, shapes)):
            return None
        self.ingest_x_fresh = False
        return xs

    def grad_counters(self) -> Optional[torch.Tensor]:
        """Allocated on first use OUTSIDE a graph capture (a zero-fill inside one would become a memset node: DESIGN.md section 6);
        while capturing without them, the reduction falls back to its own zeroing launch."""
        if self._grad_counters is None or self._grad_counters.device != self.device:
            if self.device.type != "cuda" or torch.cuda.is_current_stream_capturing():
                return None
            self._grad_counters = torch.zeros(4, dtype=torch.int32, device=self.device)
        return self._grad_counters

    @property
    def rows(self) -> int:
        return 0 if self.weight is None else self.weight.shape[0]

    def append_rows(self, n: int) -> int:
        new = torch.empty((n, self.E), dtype=torch.float32, device=self.device)
        nn.init.normal_(new)                             # nn.Embedding default init N(0,1)
        base = self.rows
        self.weight = new if self.weight is None else torch.cat([self.weight, new])
        self.grad, self.sparse_grad = None, None
        self.version += 1
        return base

    def apply(self, fn):
        new = fn(self.weight)
        if new is not self.weight:
            if new.dtype != torch.float32:
                raise TypeError("embedding tables are kept in float32")
            self.weight = new
            self.device = new.device
            self.grad = None if self.grad is None else fn(self.grad)
            self.sparse_grad = None
            self.version += 1

    @staticmethod
    def fuse(stores: List["EmbeddingStore"]) -> "EmbeddingStore":
        """Concatenate several stores into one row space and rebind their embedders."""
        uniq = []
        for s in stores:
            if all(s is not u for u in uniq):
                uniq.append(s)
        if len(uniq) == 1:
            return uniq[0]
        first = uniq[0]
        if any(s.E != first.E for s in uniq):
            raise ValueError("cannot fuse embedding stores of different embedding_dim")
        fused = EmbeddingStore(first.E, first.device, first.grad_mode)
        fused.weight = torch.cat([s.weight.to(first.device) for s in uniq])
        base = 0
        for s in uniq:
            for m in s.members:
                m._rebind(fused, m.row_base + base)
            base += s.rows
        return fused

    def optim_parameters(self):
        """The nn.Parameters an optimiser sees for this row space (per-key views)."""
        return [p for m in self.members for p in m.table_parameters()]

    def bind_grads(self):
        for m in self.members:
            m._bind_grads()

    # ---- gradient bookkeeping (called from the autograd backward) ------------------------------
    def accumulate_grad(self, plan: ops.DedupPlan, srcs, B: int, short_segments: bool = False):
        """srcs: [(d_out view [B, K*E], K)] in slot order of `plan`."""
        if self.grad_mode == "sparse":
            grad_rows = torch.empty((max(plan.M, 1), self.E), dtype=torch.float32, device=self.device)
            ops.embed_grad(plan, srcs, B, self.E, ops.TT_GRAD_SPARSE, grad_rows, short_segments, self.grad_counters(),
                           defer_finish=self.defer_long_finish)
            self.sparse_grad = (plan, grad_rows)
            return
        params = self.optim_parameters()
        fresh = any(p.grad is None for p in params) or self.grad is None
        if self.grad is None:
            self.grad = torch.empty_like(self.weight)
        if fresh:
            self.grad.zero_()                            # reference semantics: dense [V_k, E] grads
            ops.embed_grad(plan, srcs, B, self.E, ops.TT_GRAD_DENSE_SET, self.grad, counters=self.grad_counters())
            self.bind_grads()
        else:
            ops.embed_grad(plan, srcs, B, self.E, ops.TT_GRAD_DENSE_ACC, self.grad, counters=self.grad_counters())


class _Table(nn.Module):
    """Holds the per-key Parameter (a view into the fused store); name parity with nn.Embedding."""

    def __init__(self, view: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(view)

    @property
    def num_embeddings(self):
        return self.weight.shape[0]

    @property
    def embedding_dim(self):
        return self.weight.shape[1]


class CategoricalEmbedder(nn.Module):
    def __init__(self, keys: List[str], metadata_path: Union[str, Path], table_name: str, embedding_dim: int = 64,
                 device: Optional[str] = "cuda:0", safety_margin: int = 10, embedding_grad: Optional[str] = None,
                 materialize: bool = True):
        super().__init__()
        self.keys = list(keys)
        self.materialize = materialize
        self.embedding_dim = embedding_dim
        self.device_str = str(device or "cuda:0")
        self.safety_margin = safety_margin
        self.vocab_sizes = self._extract_vocab_sizes(metadata_path, table_name, self.keys)
        print(f"[CategoricalEmbedder] Initializing with {len(self.keys)} features")
        mode = embedding_grad or settings.embedding_grad
        self.store = EmbeddingStore(embedding_dim, torch.device(self.device_str), mode)
        self.embeddings = nn.ModuleDict()
        offs = []
        if materialize:
            self.row_base = self.store.append_rows(sum(self.vocab_sizes[k] for k in self.keys)) if self.keys else 0
            self.store.members.append(self)
            off = self.row_base
            for k in self.keys:
                v = self.vocab_sizes[k]
                self.embeddings[k] = _Table(self.store.weight[off:off + v])
                offs.append(off)
                off += v
        else:       # rows live in a sharded store owned by the distributed task; only the key directory is kept
            self.row_base, off = 0, 0
            for k in self.keys:
                offs.append(off)
                off += self.vocab_sizes[k]
        dev = self.store.device
        self.register_buffer("_key_row_offset", torch.tensor(offs, dtype=torch.int64, device=dev), persistent=False)
        self.register_buffer("_key_vocab", torch.tensor([self.vocab_sizes[k] for k in self.keys], dtype=torch.int64,
                                                        device=dev), persistent=False)
        self._bound_version = self.store.version

    # ---- vocab sizes: src/towers/cat_embed.py:50-85 -------------------------------------------
    def _extract_vocab_sizes(self, metadata_path, table_name: str, keys: List[str]) -> Dict[str, int]:
        try:
            counts = category_counts(table_name, metadata_path)
        except Exception as e:                                            # whole-file failure: :82-85
            print(f"[ERROR] Failed to extract vocab_sizes from metadata: {e}")
            print("[FALLBACK] Using default vocab_size=1000 for all keys")
            return {k: 1000 for k in keys}
        out = {}
        for k in keys:
            if k not in counts:
                print(f"[WARNING] Column '{k}' not found in metadata, using default vocab_size=1000")
                out[k] = 1000
            elif counts[k] is None:
                print(f"[WARNING] No category count info for '{k}', using default vocab_size=1000")
                out[k] = 1000
            else:
                out[k] = int(counts[k]) + self.safety_margin
        return out

    # ---- store plumbing -------------------------------------------------------------------------
    def table_parameters(self):
        return [self.embeddings[k].weight for k in self.keys] if self.materialize else []

    @property
    def total_rows(self) -> int:
        return sum(self.vocab_sizes[k] for k in self.keys)

    def set_row_base(self, base: int):
        """(non-materialised embedders) place this embedder's keys at `base` of a global row space."""
        off, offs = base, []
        for k in self.keys:
            offs.append(off)
            off += self.vocab_sizes[k]
        self.row_base = base
        self._key_row_offset = torch.tensor(offs, dtype=torch.int64, device=self._key_row_offset.device)

    def _rebind(self, store: Optional[EmbeddingStore] = None, row_base: Optional[int] = None):
        if store is not None and store is not self.store:
            self.store = store
            if self not in store.members:
                store.members.append(self)
        if row_base is not None:
            self.row_base = row_base
        off = self.row_base
        offs = []
        for k in self.keys:
            v = self.vocab_sizes[k]
            self.embeddings[k].weight.data = self.store.weight[off:off + v]
            self.embeddings[k].weight.grad = None
            offs.append(off)
            off += v
        dev = self.store.device
        self._key_row_offset = torch.tensor(offs, dtype=torch.int64, device=dev)
        self._key_vocab = self._key_vocab.to(dev)
        self._bound_version = self.store.version
        if self.store.grad is not None:
            self._bind_grads()

    def _bind_grads(self):
        off = self.row_base
        for k in self.keys:
            v = self.vocab_sizes[k]
            self.embeddings[k].weight.grad = self.store.grad[off:off + v]
            off += v

    def _apply(self, fn, recurse=True):
        if not self.materialize:
            dev = fn(torch.empty(0, device=self._key_vocab.device)).device
            self._key_row_offset, self._key_vocab = self._key_row_offset.to(dev), self._key_vocab.to(dev)
            self.store.device = dev
            return self
        # move the fused storage ONCE, then re-point the per-key views (keeps Parameter identity, so
        # optimisers built before .to(device) stay valid)
        self.store.apply(fn)
        for s_member in self.store.members:
            if s_member._bound_version != s_member.store.version:
                s_member._rebind()
        self._key_row_offset = self._key_row_offset.to(self.store.device)
        self._key_vocab = self._key_vocab.to(self.store.device)
        return self

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        return super().load_state_dict(state_dict, strict=strict, assign=False)   # copy through the views

    # ---- forward: src/towers/cat_embed.py:126-190 ------------------------------------------------
    def lookup_side(self, values: torch.Tensor, out_view: torch.Tensor) -> ops.LookupSide:
        return ops.LookupSide(values, self._key_row_offset, self._key_vocab, out_view, len(self.keys))

    @_no_dynamo
    def forward(self, kjt, return_dict: bool = True):
        K, E = len(self.keys), self.embedding_dim
        if kjt is None:                                                    # :143-150
            dev = self.store.device
            if return_dict:
                return {k: torch.zeros(1, E, device=dev) for k in self.keys}
            return torch.zeros(1, K * E, device=dev)
        values = kjt.values()
        cat = _LookupFn.apply(self, values, *self.table_parameters())
        if not return_dict:
            return cat
        return {k: cat[:, i * E:(i + 1) * E] for i, k in enumerate(self.keys)}


class _LookupFn(torch.autograd.Function):
    """Stand-alone differentiable lookup (used when the embedder is called outside a tower)."""

    @staticmethod
    def forward(ctx, emb: CategoricalEmbedder, values: torch.Tensor, *tables):
        K, E = len(emb.keys), emb.embedding_dim
        values = values.to(device=emb.store.device, dtype=torch.int64).contiguous()
        B = values.numel() // max(K, 1)                                   # :98
        out = torch.empty((B, K * E), dtype=torch.float32, device=emb.store.device)
        need_grad = any(ctx.needs_input_grad)
        rows = ops.embed_lookup(emb.store.weight, [emb.lookup_side(values[:B * K], out)], B, want_rows=need_grad) \
            if K and B else None
        ctx.emb, ctx.B = emb, B
        ctx.plan = ops.dedup_plan(rows, emb.store.rows) if rows is not None else None
        return out

    @staticmethod
    def backward(ctx, d_out):
        emb = ctx.emb
        if ctx.plan is not None:
            d_out = d_out.contiguous()
            emb.store.accumulate_grad(ctx.plan, [(d_out, len(emb.keys))], ctx.B)
        return (None, None) + (None,) * len(emb.keys)


def create_categorical_embedder(keys: List[str], metadata_path="meta/metadata.csv", table_name: str = "notice",
                                embedding_dim: int = 64, device: Optional[str] = "cuda:0") -> CategoricalEmbedder:
    return CategoricalEmbedder(keys=keys, metadata_path=metadata_path, table_name=table_name,
                               embedding_dim=embedding_dim, device=device)
