"""BaseTower / NoticeTower / CompanyTower on the HIP path.

Drop-in for src/towers/tower/{base_tower,notice_tower,company_tower}.py: same constructor arguments,
same sub-module names and therefore the same state-dict keys and shapes
(`categorical_embedder.embeddings.<key>.weight`, `dense_projection.{weight,bias}`,
`mlp.{4i}.{weight,bias}`, `mlp.{4i+2}.{weight,bias,running_mean,running_var,num_batches_tracked}`,
`mlp.{4n}.{weight,bias}`), same forward contract: {"dense": [B,Din] f32, "kjt": KJT} -> unit rows [B,D].

The torch modules below (nn.Linear / nn.BatchNorm1d / ...) are PARAMETER CONTAINERS only -- they give
the reference's initialisation and key names; their forward is never called.  The arithmetic runs in
libtwotower_hip.so: tt_embed_lookup_fwd writes the embedding rows straight into the MLP input buffer
(no torch.cat), tt_tower_mlp_fwd/_bwd run the MLP, tt_dedup_plan + tt_embed_grad_bwd produce the
table gradients.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch
from ._lib import no_dynamo as _no_dynamo
import torch.nn as nn

from . import _lib as L
from . import ops
from .config import settings
from .cat_embed import CategoricalEmbedder, EmbeddingStore


_DEBUG_KEEP = None   # set to a list by debugging tools


def _sync_comm(sides):
    """The communicator for SyncBN if it applies to this pass: every tower asks for it, trains, shares one batch size and
    dropout setting (the C side then checks the shape conditions of the fused tail), else None."""
    if not sides:
        return None
    comms = [getattr(s.tower, "sync_comm", None) for s in sides]
    if any(c is None for c in comms) or not all(s.train for s in sides):
        return None
    if len({s.B for s in sides}) != 1 or len({s.p_drop for s in sides}) != 1:
        raise ValueError("SyncBN needs one batch size and one dropout rate for all towers of a pass")
    return comms[0]


def _al(n: int) -> int:
    return (n + 63) // 64 * 64


class BaseTower(nn.Module):
    def __init__(self, categorical_keys: List[str], metadata_path: str = "meta/metadata.csv", table_name: str = "notice",
                 categorical_embedding_dim: int = 64, dense_input_dim: int = 256,
                 tower_hidden_dims: Optional[List[int]] = None, final_embedding_dim: int = 128, dropout_rate: float = 0.2,
                 device="cuda:0", embedding_grad: Optional[str] = None, materialize_tables: bool = True,
                 mlp_dtype: Optional[str] = None):
        super().__init__()
        if tower_hidden_dims is None:
            tower_hidden_dims = [256, 128]
        self.mlp_dtype = mlp_dtype or settings.mlp_dtype
        if self.mlp_dtype not in ("fp32", "bf16"):
            raise ValueError(f"mlp_dtype must be 'fp32' or 'bf16', got {self.mlp_dtype!r}")
        # bf16 MLP: the tower input x = [projection | embedding rows] lives in bf16 -- bit-identical results (the GEMMs
        # that read x round it to bf16 anyway) at half the bytes: lookup 15.7 -> 9.2 us, step -7 us at B = 8192.
        # A bf16 d_x (settings.tower_io_dtype = dx | both) is implemented but measures no faster (0.354 vs 0.352 ms per step: the
        # segmented reduction is bound by its dependent trips, not by bytes, and the data-gradient GEMM stores 2-byte
        # pieces) and rounds the per-slot row gradients, so it stays off.
        io = settings.tower_io_dtype if self.mlp_dtype == "bf16" else "none"
        if io not in ("none", "x", "dx", "both"):
            raise ValueError(f"settings.tower_io_dtype must be none|x|dx|both, got {io!r}")
        self.x_dtype = torch.bfloat16 if io in ("x", "both") else torch.float32
        self.dx_dtype = torch.bfloat16 if io in ("dx", "both") else torch.float32
        # the fused launches against the separate kernels they replaced (tests): config.Settings
        self.unfused_tail, self.unfused_front, self.unfused_back = settings.tower_unfused_tail, settings.tower_unfused_front, settings.tower_unfused_back
        self.categorical_keys = list(categorical_keys)
        self.exchange = None            # set by the distributed task: sharded-table row exchange
        self.sync_comm = None           # set by the distributed task (sync_bn=True): BN statistics over all ranks' rows
        self.pack_for_score = False     # set by the train task (score_dtype='bf16'): also emit the score kernels' operand images
        self._last_packed = None
        self.pack_scale = 1.0           # the images hold bf16(pack_scale * emb)
        self._seed_dev = None           # set by GraphedTrainStep: device word added to the dropout seed
        self._w16 = None                # set by GraphedTrainStep while its step runs: (bf16 shadow of w_proj, [bf16 shadows of the blocks' weights])
        self._seed_override = None      # tests: a fixed dropout seed instead of one drawn from torch's CPU generator
        self.device = device
        self.tower_hidden_dims = list(tower_hidden_dims)
        self.final_embedding_dim = final_embedding_dim
        self.dropout_rate = float(dropout_rate)
        self.dense_input_dim = dense_input_dim
        dev = torch.device(device)
        self.categorical_embedder = CategoricalEmbedder(keys=self.categorical_keys, metadata_path=metadata_path,
                                                        table_name=table_name, embedding_dim=categorical_embedding_dim,
                                                        device=str(dev), embedding_grad=embedding_grad,
                                                        materialize=materialize_tables)
        self.dense_projection = nn.Linear(dense_input_dim, tower_hidden_dims[0])
        self._build_mlp(categorical_embedding_dim, tower_hidden_dims, final_embedding_dim, dropout_rate)
        self._struct_key = None
        self._params_struct = None
        self.to(dev)                    # (the reference hard-codes cuda:0 here: base_tower.py:69)

    def _build_mlp(self, categorical_embedding_dim, tower_hidden_dims, final_embedding_dim, dropout_rate):
        in_dim = tower_hidden_dims[0] + len(self.categorical_keys) * categorical_embedding_dim
        layers = []
        for h in tower_hidden_dims[1:]:
            layers += [nn.Linear(in_dim, h), nn.ReLU(), nn.BatchNorm1d(h), nn.Dropout(dropout_rate)]
            in_dim = h
        layers.append(nn.Linear(in_dim, final_embedding_dim))
        self.mlp = nn.Sequential(*layers)

    # ---- parameter plumbing ----------------------------------------------------------------------
    @property
    def n_hidden(self) -> int:
        return len(self.tower_hidden_dims) - 1

    def dense_parameters(self) -> List[torch.Tensor]:
        """Order = order of the gradients tt_tower_mlp_bwd returns."""
        ps = [self.dense_projection.weight, self.dense_projection.bias]
        for i in range(self.n_hidden):
            lin, bn = self.mlp[4 * i], self.mlp[4 * i + 2]
            ps += [lin.weight, lin.bias, bn.weight, bn.bias]
        ps += [self.mlp[4 * self.n_hidden].weight, self.mlp[4 * self.n_hidden].bias]
        return ps

    def _params(self):
        nh = self.n_hidden
        lins = [self.mlp[4 * i] for i in range(nh)]
        bns = [self.mlp[4 * i + 2] for i in range(nh)]
        out = self.mlp[4 * nh]
        tensors = [self.dense_projection.weight, self.dense_projection.bias, out.weight, out.bias]
        for lin, bn in zip(lins, bns):
            tensors += [lin.weight, lin.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]
        w16 = self._w16
        key = tuple(t.data_ptr() for t in tensors) + tuple(b.num_batches_tracked.data_ptr() for b in bns) + (self.mlp_dtype, self.x_dtype, self.dx_dtype, self.unfused_tail, self.unfused_front, self.unfused_back) + \
            ((w16[0].data_ptr(),) + tuple(w.data_ptr() for w in w16[1]) if w16 is not None else ())
        if key != self._struct_key:
            for t in tensors:
                if t.dtype != torch.float32 or not t.is_contiguous():
                    raise TypeError("tower parameters must be contiguous float32")
            E = self.categorical_embedder.embedding_dim
            self._params_struct = ops.tower_params_struct(
                self.dense_input_dim, self.tower_hidden_dims[0], len(self.categorical_keys) * E, self.tower_hidden_dims[1:],
                self.final_embedding_dim, self.dense_projection.weight, self.dense_projection.bias,
                [l.weight for l in lins], [l.bias for l in lins], [b.weight for b in bns], [b.bias for b in bns],
                [b.running_mean for b in bns], [b.running_var for b in bns], out.weight, out.bias,
                bn_nbt=[b.num_batches_tracked for b in bns],
                compute_dtype=ops.TT_BF16 if self.mlp_dtype == "bf16" else ops.TT_F32,
                x_dtype=ops.TT_BF16 if self.x_dtype == torch.bfloat16 else ops.TT_F32,
                dx_dtype=ops.TT_BF16 if self.dx_dtype == torch.bfloat16 else ops.TT_F32,
                flags=(ops.L.TT_TOWER_UNFUSED_TAIL if self.unfused_tail else 0) | (ops.L.TT_TOWER_UNFUSED_FRONT if self.unfused_front else 0) | (ops.L.TT_TOWER_UNFUSED_BACK if self.unfused_back else 0),
                w_proj_bf16=None if w16 is None else w16[0], w_bf16=() if w16 is None else w16[1])
            self._struct_key = key
        return self._params_struct

    @property
    def x_width(self) -> int:
        return self.tower_hidden_dims[0] + len(self.categorical_keys) * self.categorical_embedder.embedding_dim

    # ---- forward: src/towers/tower/base_tower.py:101-147 ------------------------------------------
    @_no_dynamo
    def forward(self, tower_input: Dict[str, torch.Tensor]) -> torch.Tensor:
        return run_towers([self], [tower_input])[0]


class NoticeTower(BaseTower):
    def __init__(self, categorical_keys: List[str], metadata_path: str = "meta/metadata.csv",
                 categorical_embedding_dim: int = 64, dense_input_dim: int = 256,
                 tower_hidden_dims: Optional[List[int]] = None, final_embedding_dim: int = 128, dropout_rate: float = 0.2,
                 device="cuda:0", embedding_grad: Optional[str] = None, materialize_tables: bool = True,
                 mlp_dtype: Optional[str] = None):
        super().__init__(categorical_keys=categorical_keys, metadata_path=metadata_path, table_name="notice",
                         categorical_embedding_dim=categorical_embedding_dim, dense_input_dim=dense_input_dim,
                         tower_hidden_dims=tower_hidden_dims, final_embedding_dim=final_embedding_dim,
                         dropout_rate=dropout_rate, device=device, embedding_grad=embedding_grad,
                         materialize_tables=materialize_tables, mlp_dtype=mlp_dtype)


class CompanyTower(BaseTower):
    def __init__(self, categorical_keys: List[str], metadata_path: str = "meta/metadata.csv",
                 categorical_embedding_dim: int = 64, dense_input_dim: int = 128,
                 tower_hidden_dims: Optional[List[int]] = None, final_embedding_dim: int = 128, dropout_rate: float = 0.2,
                 device="cuda:0", embedding_grad: Optional[str] = None, materialize_tables: bool = True,
                 mlp_dtype: Optional[str] = None):
        super().__init__(categorical_keys=categorical_keys, metadata_path=metadata_path, table_name="company",
                         categorical_embedding_dim=categorical_embedding_dim, dense_input_dim=dense_input_dim,
                         tower_hidden_dims=tower_hidden_dims, final_embedding_dim=final_embedding_dim,
                         dropout_rate=dropout_rate, device=device, embedding_grad=embedding_grad,
                         materialize_tables=materialize_tables, mlp_dtype=mlp_dtype)


# --------------------------------------------------------------------------------------------------
# one autograd node for >= 1 towers (lookup and table gradient fused across towers that share a store)
# --------------------------------------------------------------------------------------------------
def run_towers(towers: Sequence[BaseTower], inputs: Sequence[Dict]) -> List[torch.Tensor]:
    flat = []
    for tw, inp in zip(towers, inputs):
        dev = tw.categorical_embedder.store.device
        dense = inp["dense"].to(dev)
        kjt = inp["kjt"]
        values = (kjt.to(dev) if hasattr(kjt, "to") else kjt).values()
        flat += [dense, values] + tw.dense_parameters() + tw.categorical_embedder.table_parameters()
    outs = _TowersFn.apply(tuple(towers), *flat)
    outs = list(outs) if isinstance(outs, tuple) else [outs]
    for tw, o in zip(towers, outs):                      # hand the packed images to the score step with the rows they belong to
        if tw._last_packed is not None:
            o._tt_packed, tw._last_packed = tw._last_packed, None
    return outs


class _Side:
    __slots__ = ("tower", "B", "acts", "acts_struct", "buf", "x", "emb", "train", "seed", "p_drop", "sync_keep", "packed", "values")


class _TowersFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, towers, *flat):
        sides: List[_Side] = []
        pos = 0
        grad_on = any(ctx.needs_input_grad)       # (grad mode is always off inside Function.forward)
        spans = []
        lookups: Dict[int, list] = {}
        for tw in towers:
            emb = tw.categorical_embedder
            nd, nt = len(tw.dense_parameters()), len(emb.table_parameters())
            dense, values = flat[pos], flat[pos + 1]
            spans.append((pos, nd, nt))
            pos += 2 + nd + nt
            dev = emb.store.device
            if dense.dim() != 2 or dense.shape[1] != tw.dense_input_dim:
                raise ValueError(f"dense input must be [B, {tw.dense_input_dim}], got {tuple(dense.shape)}")
            dense = dense.to(dtype=torch.float32).contiguous()
            values = values.to(dtype=torch.int64).contiguous()
            B, K, E = dense.shape[0], len(emb.keys), emb.embedding_dim
            if values.numel() != B * K:
                raise ValueError(f"kjt carries {values.numel()} ids but the batch needs B*K = {B}*{K}")
            s = _Side()
            s.tower, s.B, s.train = tw, B, tw.training
            s.p_drop = tw.dropout_rate if tw.training else 0.0
            s.seed = 0 if s.p_drop <= 0 else (tw._seed_override if tw._seed_override is not None else
                                              int(torch.empty((), dtype=torch.int64).random_().item()))
            s.values = values
            s.acts = (dense,)
            sides.append(s)
        # a graph-replayed step's hand-over launch may ALREADY have looked the rows up into persistent input buffers
        # (ops.batch_ingest(table=...): tt_batch_ingest_lookup): then x is that buffer and no lookup launch follows
        pre = {}
        if towers[0].exchange is None and grad_on:
            by_store: Dict[int, list] = {}
            for s in sides:
                emb = s.tower.categorical_embedder
                if s.B and len(emb.keys):
                    by_store.setdefault(id(emb.store), []).append(s)
            for group in by_store.values():
                st = group[0].tower.categorical_embedder.store
                xs = st.x_for([g.values for g in group], [(g.B, g.tower.x_width, g.tower.x_dtype) for g in group])
                if xs is not None:
                    for g, x in zip(group, xs):
                        pre[id(g)] = x
        for s in sides:
            tw, B, dense, values = s.tower, s.B, s.acts[0], s.values
            emb = tw.categorical_embedder
            K, E = len(emb.keys), emb.embedding_dim
            dev = emb.store.device
            # one flat activation buffer: x | (pre_i, act_i)* | (mean_i, rstd_i)* | y
            hid = tw.tower_hidden_dims[1:]
            px = pre.get(id(s))
            x_f32 = tw.x_dtype == torch.float32 and px is None
            sizes = [_al(B * tw.x_width) if x_f32 else 0] + [_al(B * h) for h in hid for _ in (0, 1)] + \
                    [_al(h) for h in hid for _ in (0, 1)] + [_al(B * tw.final_embedding_dim)]
            s.buf = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
            offs = [0]
            for z in sizes:
                offs.append(offs[-1] + z)
            s.x = px if px is not None else (s.buf[:B * tw.x_width].view(B, tw.x_width) if x_f32 else
                                             torch.empty((B, tw.x_width), dtype=tw.x_dtype, device=dev))
            s.emb = torch.empty((B, tw.final_embedding_dim), dtype=torch.float32, device=dev)
            a = L.TowerActs()
            base, esz = s.buf.data_ptr(), 4
            a.dense, a.x, a.emb = dense.data_ptr(), s.x.data_ptr(), s.emb.data_ptr()
            nh = len(hid)
            for i in range(nh):
                a.pre[i] = base + esz * offs[1 + 2 * i]
                a.act[i] = base + esz * offs[2 + 2 * i]
                a.mean[i] = base + esz * offs[1 + 2 * nh + 2 * i]
                a.rstd[i] = base + esz * offs[2 + 2 * nh + 2 * i]
            a.y = base + esz * offs[1 + 4 * nh]
            s.packed = None
            if tw.pack_for_score and B:
                # the score kernels' bf16 operand images of the unit rows, written by the tower pass itself (tt_tower_acts.emb_packed)
                s.packed = torch.empty(L.load().tt_score_pack_bytes(B, tw.final_embedding_dim), dtype=torch.uint8, device=dev)
                a.emb_packed, a.emb_pack_scale = s.packed.data_ptr(), float(tw.pack_scale)
            s.acts_struct = a
            if K and B:
                lookups.setdefault(id(emb.store), []).append((s, emb.lookup_side(values, s.x[:, tw.tower_hidden_dims[0]:]), px is not None))
        # fused lookup (+ duplicate-row plan when a backward will follow) per store
        plans = []
        exch = towers[0].exchange
        ctx.exch, ctx.exch_state = exch, None
        if exch is not None:            # sharded table: ids -> owners, pooled rows <- owners (distributed.py)
            group = [g for grp in lookups.values() for g in grp]
            if group:
                ctx.exch_state = exch.forward([g[1] for g in group], group[0][0].B, grad_on)
                ctx.exch_sides = [g[0] for g in group]
            lookups = {}
        for group in lookups.values():
            store: EmbeddingStore = group[0][0].tower.categorical_embedder.store
            B = group[0][0].B
            if any(g[0].B != B for g in group):
                # different batch sizes cannot share one launch: fall back to one launch per side
                for g in group:
                    rows = ops.embed_lookup(store.weight, [g[1]], g[0].B, want_rows=grad_on)
                    plans.append([store, [g[0]], ops.dedup_plan(rows, store.rows) if grad_on else None])
                continue
            # a graph-replayed step hands the batch over with ops.batch_ingest, which leaves the fused rows of exactly these id
            # tensors in key-major order: the plan sorts those, the lookup need not write its slot-major copy
            km = store.rows_km_for([g[1].ids for g in group]) if (grad_on and 0 < B <= ops.KEYED_MAX_B) else None
            sm = store.rows_sm_for([g[1].ids for g in group]) if km is not None else None
            if all(g[2] for g in group) and km is not None:
                rows = None                             # the hand-over launch has filled x already (tt_batch_ingest_lookup)
            elif sm is not None and store.weight.shape[1] % 4 == 0 and all(g[1].out.stride(0) % 4 == 0 for g in group):
                rows = None                             # ... or left the clamped fused rows in slot order: no id decoding here
                ops.embed_lookup_rows(store.weight, sm, [g[1] for g in group], B)
            else:
                rows = ops.embed_lookup(store.weight, [g[1] for g in group], B, want_rows=grad_on and km is None)
            plans.append([store, [g[0] for g in group], None, rows if km is None else km, grad_on, km is not None])
        # duplicate-row plans: depend on ids only, first needed in the backward.  In line on the launch stream (a side stream
        # overlapped the sort with the score kernels, but a captured graph with two streams is replayed node by node with
        # cross-queue signals: 4-6 us gaps in front of eight kernels -- as much as the overlap saved), and BEFORE the towers'
        # forward: with TT_OPT_DEFER_RIDERS the plan's compaction then rides in the towers' tail launch
        for pl in plans:
            if len(pl) == 6:
                store, psides, _, rows, want_plan, key_major = pl
                plan = None
                if want_plan:
                    Bs = psides[0].B
                    if 0 < Bs <= ops.KEYED_MAX_B:       # per-key LDS sorts (2 launches)
                        plan = ops.dedup_plan_keyed(rows, [len(q.tower.categorical_embedder.keys) for q in psides], Bs, key_major,
                                                    E=int(store.E))
                    else:
                        plan = ops.dedup_plan(rows, store.rows)
                    plan.keep = rows                    # keep the sort input alive until it has run
                pl[:] = [store, psides, plan]
        live = [s for s in sides if s.B]
        fused = len(live) > 1 and len({s.B for s in live}) == 1 and len({s.tower.n_hidden for s in live}) == 1 and \
            len({(s.train, s.p_drop) for s in live}) == 1
        comm = _sync_comm(live)
        if comm is not None:
            # SyncBN: the pass stops at the local BN statistics, ONE all-gather carries every tower's (n, mean, M2) triples,
            # and the second half merges them in rank order (twotower.h: sync_phase)
            s0 = live[0]
            widths = [3 * s.tower.tower_hidden_dims[1] for s in live]
            loc = torch.empty(sum(widths), dtype=torch.float32, device=s0.emb.device)
            params = [s.tower._params() for s in live]
            off = 0
            for s, p, w in zip(live, params, widths):
                s.acts_struct.bn_sync_local = loc.data_ptr() + 4 * off
                p.sync_phase, p.sync_ranks, p.rng_row_offset = 1, comm.world, comm.rank * s.B
                off += w
            try:
                ops.towers_fwd(params, [s.acts_struct for s in live], s0.B, s0.train, s0.p_drop, s0.seed, s0.emb.device, s0.tower._seed_dev)
                allg = comm.all_gather(loc)
                off = 0
                for s, p, w in zip(live, params, widths):
                    s.acts_struct.bn_sync_all, s.acts_struct.bn_sync_stride = allg.data_ptr() + 4 * off, loc.numel()
                    p.sync_phase = 2
                    off += w
                ops.towers_fwd(params, [s.acts_struct for s in live], s0.B, s0.train, s0.p_drop, s0.seed, s0.emb.device, s0.tower._seed_dev)
            finally:
                for p in params:
                    p.sync_phase = 0
            for s in live:
                s.seed, s.sync_keep = s0.seed, (loc, allg)
        elif fused:           # horizontal fusion: one launch per layer step covers every tower
            s0 = live[0]
            ops.towers_fwd([s.tower._params() for s in live], [s.acts_struct for s in live], s0.B, s0.train, s0.p_drop, s0.seed,
                           s0.emb.device, s0.tower._seed_dev)
            for s in live:
                s.seed = s0.seed
        else:
            for s in live:
                tw = s.tower
                ops.tower_fwd(tw._params(), s.acts_struct, s.B, s.train, s.p_drop, s.seed, s.emb.device, tw._seed_dev)
        for s in sides:
            s.tower._last_packed = None if s.packed is None else (s.packed, float(s.tower.pack_scale))
        ctx.sides, ctx.plans, ctx.spans, ctx.n_flat = sides, plans, spans, len(flat)
        # hand out aliases: keeping the returned objects themselves on ctx would form a reference cycle
        outs = tuple(s.emb.view(s.emb.shape) for s in sides)
        ctx.mark_non_differentiable(*[o for o, s in zip(outs, sides) if s.B == 0])
        return outs

    @staticmethod
    def backward(ctx, *d_embs):
        grads = [None] * ctx.n_flat
        dxs = {}
        exch = ctx.exch
        if exch is not None and exch.world > 1:            # global objective = mean over ranks of the local losses
            d_embs = [None if d is None else d * (1.0 / exch.world) for d in d_embs]
        flat_grads = []
        work = [(s, d, sp) for s, d, sp in zip(ctx.sides, d_embs, ctx.spans) if s.B and d is not None]
        prepared = []
        # one buffer for the whole pass: [dense gradients of every tower | per-tower scratch] -- the dense part is ONE
        # contiguous range, so the data-parallel sum is a single all-reduce however many towers there are
        side_sizes = []
        for s, _, _ in work:
            tw = s.tower
            side_sizes.append(([_al(p.numel()) for p in tw.dense_parameters()],
                               [_al(s.B * tw.x_width) if tw.dx_dtype == torch.float32 else 0] +
                               [_al(s.B * h) for h in tw.tower_hidden_dims[1:]] + [_al(s.B * tw.final_embedding_dim)]))
        dense_total = sum(sum(d) for d, _ in side_sizes)
        buf = torch.empty(dense_total + sum(sum(r) for _, r in side_sizes), dtype=torch.float32, device=work[0][0].emb.device) if work else None
        dense_at, scratch_at = 0, dense_total
        for (s, d_emb, (pos, nd, nt)), (dsz, rsz) in zip(work, side_sizes):
            tw = s.tower
            dev = s.emb.device
            d_emb = d_emb.to(dtype=torch.float32).contiguous()
            dps = tw.dense_parameters()
            hid = tw.tower_hidden_dims[1:]
            B = s.B
            dx_f32 = tw.dx_dtype == torch.float32
            offs = [dense_at]                                   # absolute offsets in buf: dense entries, then this tower's scratch
            for z in dsz[:-1]:
                offs.append(offs[-1] + z)
            dense_at += sum(dsz)
            offs.append(scratch_at)
            for z in rsz:
                offs.append(offs[-1] + z)
            scratch_at += sum(rsz)
            base = buf.data_ptr()
            views = [buf[offs[i]:offs[i] + p.numel()].view(p.shape) for i, p in enumerate(dps)]
            g = L.TowerGrads()
            g.w_proj, g.b_proj = base + 4 * offs[0], base + 4 * offs[1]
            for i in range(len(hid)):
                g.w[i], g.b[i] = base + 4 * offs[2 + 4 * i], base + 4 * offs[3 + 4 * i]
                g.bn_w[i], g.bn_b[i] = base + 4 * offs[4 + 4 * i], base + 4 * offs[5 + 4 * i]
                g.scratch[i] = base + 4 * offs[len(dps) + 1 + i]
            g.w_out, g.b_out = base + 4 * offs[len(dps) - 2], base + 4 * offs[len(dps) - 1]
            d_x = buf[offs[len(dps)]:offs[len(dps)] + B * tw.x_width].view(B, tw.x_width) if dx_f32 else \
                torch.empty((B, tw.x_width), dtype=tw.dx_dtype, device=dev)
            g.d_x = d_x.data_ptr()
            g.d_y = base + 4 * offs[len(dps) + 1 + len(hid)]
            prepared.append((s, d_emb, g))
            if _DEBUG_KEEP is not None:      # tests/debug/tail_diff.py: keep the backward scratch for inspection
                _DEBUG_KEEP.append({"buf": buf, "offs": offs, "n_dense": len(dps), "hidden": hid, "B": B, "acts": s.buf, "emb": s.emb,
                                    "d_emb": d_emb})
            for i, v in enumerate(views):
                grads[pos + 2 + i] = v
            dxs[id(s)] = d_x[:, tw.tower_hidden_dims[0]:]
        fused = len(prepared) > 1 and len({s.B for s, _, _ in prepared}) == 1 and \
            len({s.tower.n_hidden for s, _, _ in prepared}) == 1 and len({(s.train, s.p_drop, s.seed) for s, _, _ in prepared}) == 1
        comm = _sync_comm([s for s, _, _ in prepared])
        if comm is not None:
            s0 = prepared[0][0]
            widths = [2 * s.tower.tower_hidden_dims[1] for s, _, _ in prepared]
            loc = torch.empty(sum(widths), dtype=torch.float32, device=s0.emb.device)
            params = [s.tower._params() for s, _, _ in prepared]
            args = ([s.acts_struct for s, _, _ in prepared], [d for _, d, _ in prepared], [g for _, _, g in prepared], s0.B, s0.train,
                    s0.p_drop, s0.seed, s0.emb.device, s0.tower._seed_dev)
            off = 0
            for (s, _, g), p, w in zip(prepared, params, widths):
                g.s_sync_local = loc.data_ptr() + 4 * off
                p.sync_phase, p.sync_ranks, p.rng_row_offset = 1, comm.world, comm.rank * s.B
                off += w
            try:
                ops.towers_bwd(params, *args)
                allg = comm.all_gather(loc)                         # every rank's (S1, S2), added in rank order
                off = 0
                for (s, _, g), p, w in zip(prepared, params, widths):
                    g.s_sync_all, g.s_sync_stride = allg.data_ptr() + 4 * off, loc.numel()
                    p.sync_phase = 2
                    off += w
                ops.towers_bwd(params, *args)
            finally:
                for p in params:
                    p.sync_phase = 0
        elif fused:
            s0 = prepared[0][0]
            ops.towers_bwd([s.tower._params() for s, _, _ in prepared], [s.acts_struct for s, _, _ in prepared],
                           [d for _, d, _ in prepared], [g for _, _, g in prepared], s0.B, s0.train, s0.p_drop, s0.seed,
                           s0.emb.device, s0.tower._seed_dev)
        else:
            for s, d_emb, g in prepared:
                ops.tower_bwd(s.tower._params(), s.acts_struct, d_emb, g, s.B, s.train, s.p_drop, s.seed, s.emb.device,
                              s.tower._seed_dev)
        if exch is not None:
            # the exchange's backward first: its local reduction is the launch that hosts a queued (deferred) slab reduction
            if ctx.exch_state is not None:
                srcs = []
                for s in ctx.exch_sides:
                    K = len(s.tower.categorical_embedder.keys)
                    d = dxs.get(id(s))
                    if d is None:
                        d = torch.zeros((s.B, K * exch.E), dtype=s.tower.dx_dtype, device=s.emb.device)
                    srcs.append((d, K))
                exch.backward(ctx.exch_state, srcs, ctx.exch_sides[0].B)
            if buf is not None and dense_total:
                flat_grads.append(buf[:dense_total])
                # a slab reduction that is STILL queued owes w[0] / b[0] / w_proj / b_proj of this buffer: it must have run before
                # anything outside the library reads the dense gradients (GraphedTrainStep sets the deferral)
                L.flush_deferred(buf.device)
            exch.all_reduce_dense(flat_grads)
        # table gradients: one fused segmented reduction per store
        for store, plan_sides, plan in ctx.plans:
            if plan is None:
                continue
            srcs = []
            for s in plan_sides:
                K = len(s.tower.categorical_embedder.keys)
                d = dxs.get(id(s))
                if d is None:       # this tower received no gradient: contribute zeros
                    d = torch.zeros((s.B, K * store.E), dtype=s.tower.dx_dtype, device=store.device)
                srcs.append((d, K))
            store.accumulate_grad(plan, srcs, plan_sides[0].B)
        return (None, *grads)


def tower_dense_param_names(n_hidden: int) -> List[str]:
    names = ["dense_projection.weight", "dense_projection.bias"]
    for i in range(n_hidden):
        names += [f"mlp.{4 * i}.weight", f"mlp.{4 * i}.bias", f"mlp.{4 * i + 2}.weight", f"mlp.{4 * i + 2}.bias"]
    names += [f"mlp.{4 * n_hidden}.weight", f"mlp.{4 * n_hidden}.bias"]
    return names
