"""FusedAdam -- torch.optim.Adam semantics (scripts/train.py:231: lr, betas=(0.9,0.999), eps=1e-8,
coupled weight_decay) on the HIP path, usable wherever the reference builds `optim.Adam(
train_task.parameters(), ...)` (works with LambdaLR warm-up: lr is read from param_groups each step).

  * tower weights            : one tt_adam_multi_step launch per 32 tensors (exact dense Adam)
  * embedding tables, dense  : ONE tt_adam_dense_step over the fused [R, E] store (exact: identical to
    grad mode                  per-key Adam since the update is elementwise)
  * embedding tables, sparse : tt_sparse_adam_step over the rows looked up in this step only.  Rows that
    grad mode                  were not looked up keep weight / exp_avg / exp_avg_sq untouched (the
                               reference's dense Adam would decay their moments and apply weight decay);
                               bias correction uses the global step.  See DESIGN.md "optimiser semantics".

State layout matches torch.optim.Adam ('step', 'exp_avg', 'exp_avg_sq' per parameter; for table
parameters these are views into store-level buffers), so optimizer.state_dict() round-trips through
the reference's checkpoint format (scripts/train.py:506-511).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from . import ops
from .cat_embed import EmbeddingStore


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, stores: List[EmbeddingStore] = ()):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._stores: List[EmbeddingStore] = list(stores)
        self._store_state: Dict[int, dict] = {}
        self._hp_dev = None             # [n_groups, 8] device floats while a captured graph owns the step
        self._step_cache = None         # current_step() as of the last eager change (peek_step)

    @classmethod
    def for_task(cls, task, **kw):
        """Collect the embedding stores of a TwoTowerTrainTask / TwoTowerModel / tower automatically."""
        return cls(task.parameters(), stores=find_stores(task), **kw)

    # ---- store-level state --------------------------------------------------------------------------
    def _state_of(self, store: EmbeddingStore) -> dict:
        st = self._store_state.get(id(store))
        if st is None or st["m"].shape != store.weight.shape or st["m"].device != store.weight.device:
            st = {"m": torch.zeros_like(store.weight), "v": torch.zeros_like(store.weight), "step": 0}
            self._store_state[id(store)] = st
            shard = getattr(store, "shard_param", None)
            if shard is not None:                            # sharded store: one parameter = the local rows
                self.state[shard] = {"step": torch.tensor(0.0), "exp_avg": st["m"], "exp_avg_sq": st["v"]}
            for emb in ([] if shard is not None else store.members):   # per-parameter views, torch.optim.Adam layout
                off = emb.row_base
                for k in emb.keys:
                    n = emb.vocab_sizes[k]
                    p = emb.embeddings[k].weight
                    self.state[p] = {"step": torch.tensor(0.0), "exp_avg": st["m"][off:off + n],
                                     "exp_avg_sq": st["v"][off:off + n]}
                    off += n
        return st

    def _table_param_ids(self):
        return {id(p) for s in self._stores for p in s.optim_parameters()}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._flush_pending()
        self._step_cache = None
        table_ids = self._table_param_ids()
        group_of = {}
        for group in self.param_groups:
            for p in group["params"]:
                group_of[id(p)] = group
        # ---- tower weights ----
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            hp = None if self._hp_dev is None else self._hp_dev[gi]
            items = []
            step_no = None
            for p in group["params"]:
                if id(p) in table_ids or p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                step_no = int(st["step"].item())
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                items.append((p, g, st["exp_avg"], st["exp_avg_sq"], step_no))
            by_step: Dict[int, list] = {}
            for it in items:
                by_step.setdefault(it[4], []).append(it[:4])
            for s_no, its in by_step.items():
                fuse = self._fusable_store(group, s_no, len(its), len(by_step))
                if fuse is not None:      # tower weights + looked-up table rows: one launch
                    store, st = fuse
                    plan, grad_rows = store.sparse_grad
                    st["step"] += 1
                    ops.adam_fused(its, store.weight, st["m"], st["v"], plan, grad_rows, s_no, group["lr"], b1, b2,
                                   group["eps"], group["weight_decay"], hp)
                    store.sparse_grad = None
                    for p in store.optim_parameters():
                        self.state[p]["step"] = torch.tensor(float(st["step"]))
                    continue
                ops.adam_multi(its, s_no, group["lr"], b1, b2, group["eps"], group["weight_decay"], hp)
        # ---- embedding stores ----
        for store in self._stores:
            members = store.optim_parameters()
            if not members:
                continue
            group = group_of.get(id(members[0]))
            if group is None:
                continue                                     # tables not handed to this optimiser
            b1, b2 = group["betas"]
            hp = None if self._hp_dev is None else self._hp_dev[self.param_groups.index(group)]
            st = self._state_of(store)
            if store.grad_mode == "sparse":
                if store.sparse_grad is None:
                    continue
                plan, grad_rows = store.sparse_grad
                st["step"] += 1
                ops.adam_sparse(store.weight, st["m"], st["v"], plan, grad_rows, st["step"], group["lr"], b1, b2,
                                group["eps"], group["weight_decay"], hp)
                store.sparse_grad = None
            else:
                if store.grad is None or any(p.grad is None for p in members):
                    continue
                st["step"] += 1
                ops.adam_dense(store.weight, store.grad, st["m"], st["v"], st["step"], group["lr"], b1, b2,
                               group["eps"], group["weight_decay"], hp)
            for p in members:
                self.state[p]["step"] = torch.tensor(float(st["step"]))
        return loss

    def _fusable_store(self, group, s_no: int, n_items: int, n_buckets: int):
        """The one sparse-gradient store of `group` whose next step number is s_no (else None)."""
        if n_buckets != 1 or not (1 <= n_items <= 32):
            return None
        cands = []
        for store in self._stores:
            members = store.optim_parameters()
            if members and any(members[0] is p for p in group["params"]):
                cands.append(store)
        if len(cands) != 1:
            return None
        store = cands[0]
        if store.grad_mode != "sparse" or store.sparse_grad is None or store.sparse_grad[0].M < 1:
            return None
        st = self._state_of(store)
        return (store, st) if st["step"] + 1 == s_no else None

    def advance_steps(self, n: int):
        """Account for `n` optimiser steps executed by graph replays (step counters live on the host;
        folded into the per-parameter state lazily)."""
        self._pending_steps = getattr(self, "_pending_steps", 0) + n

    def _flush_pending(self):
        n = getattr(self, "_pending_steps", 0)
        if n:
            self._pending_steps = 0
            if self._step_cache is not None:
                self._step_cache += n
            for st in self._store_state.values():
                st["step"] += n
            for st in self.state.values():
                if "step" in st:
                    st["step"] = st["step"] + float(n)

    def current_step(self) -> int:
        self._flush_pending()
        steps = [int(float(st["step"])) for st in self.state.values() if "step" in st]
        return max(steps) if steps else 0

    def peek_step(self) -> int:
        """Number of optimiser steps taken so far -- eager step() calls AND graph replays accounted through advance_steps() --
        without walking the per-parameter state on every call: a captured step asks for it at every hand-over (its bias
        corrections must follow the optimiser's own count when eager steps and replays interleave: GraphedTrainStep._fill_slot)."""
        if self._step_cache is None:
            self._step_cache = self.current_step()                 # (flushes the pending replays into the state)
        return self._step_cache + getattr(self, "_pending_steps", 0)

    def state_dict(self):
        self._flush_pending()
        return super().state_dict()

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        for store in self._stores:
            store.sparse_grad = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._step_cache = None
        # re-point the table parameters' moments at store-level buffers (the kernels update those; self.state holds views)
        for store in self._stores:
            shard = getattr(store, "shard_param", None)
            if shard is not None:
                # row-wise sharded store: ONE parameter = this rank's rows, its moments are the store-level buffers themselves
                old = dict(self.state[shard]) if shard in self.state and "exp_avg" in self.state[shard] else None
                self._store_state.pop(id(store), None)
                if old is None:
                    continue
                st = self._state_of(store)                     # fresh zero buffers, aliased into self.state[shard]
                st["m"].copy_(old["exp_avg"].to(st["m"].device))
                st["v"].copy_(old["exp_avg_sq"].to(st["v"].device))
                st["step"] = int(float(old["step"]))
                self.state[shard]["step"] = torch.tensor(float(st["step"]))
                continue
            loaded = {}
            for emb in store.members:
                for k in emb.keys:
                    p = emb.embeddings[k].weight
                    if p in self.state and "exp_avg" in self.state[p]:
                        loaded[id(p)] = dict(self.state[p])
            self._store_state.pop(id(store), None)
            if not loaded:
                continue
            st = self._state_of(store)
            for emb in store.members:
                for k in emb.keys:
                    p = emb.embeddings[k].weight
                    old = loaded.get(id(p))
                    if old is not None:
                        self.state[p]["exp_avg"].copy_(old["exp_avg"])
                        self.state[p]["exp_avg_sq"].copy_(old["exp_avg_sq"])
                        st["step"] = max(st["step"], int(float(old["step"])))
                        self.state[p]["step"] = torch.tensor(float(st["step"]))


def find_stores(module) -> List[EmbeddingStore]:
    from .cat_embed import CategoricalEmbedder
    out = []
    for m in module.modules():
        if hasattr(m, "embedding_stores"):
            out += [s for s in m.embedding_stores() if all(s is not o for o in out)]
        if isinstance(m, CategoricalEmbedder) and m.materialize and all(m.store is not s for s in out):
            out.append(m.store)
    return out
