"""Multi-GPU two-tower step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-GPU (no collective anywhere: SURVEY.md §2.1); this is new design.

  * tables   : ONE global fused row space (notice keys then company keys), sharded ROW-WISE:
               owner(row) = row % world, local index = row // world  -> every key, including the two
               29 k-row (x scale) keys, is balanced over the GPUs.
  * batch    : each rank trains on its own B_local pairs; in-batch negatives are the rank's own
               (standard data-parallel semantics: global objective = mean over ranks of local losses).
  * exchange : per step  (1) bucket the rank's B_local*(K_n+K_c) row ids by owner (stable LSD sort on
               the device), (2) all-to-all of the bucket sizes, (3) all-to-all of the ids, (4) owners
               gather their rows (tt_embed_lookup_fwd on the local shard), (5) all-to-all of the pooled
               rows back, un-permuted straight into the towers' MLP input buffers; the backward mirrors
               it: all-to-all of the row gradients to the owners, duplicate-row plan + segmented
               reduction there, then the (sparse) Adam update on the local shard.
  * towers   : dense weights replicated; gradients summed with one all-reduce per tower over the flat
               gradient buffer tt_tower_mlp_bwd fills (0.3 MB: latency-bound, a single call).
               BatchNorm statistics are per rank (as torch DDP without SyncBatchNorm).

xGMI is point-to-point (7 links per GPU), so the pooled-row all-to-all places each peer's chunk on its
own link; ring-shaped collectives are avoided on the data path.

`RowExchange` holds the routing logic and is device-agnostic: its compute steps come from a backend
object.  `HipBackend` (the product path) calls the C ABI; the CPU/gloo tests inject a checker backend
(tests/test_distributed_gloo.py) -- there is no CPU fallback in the product path.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
from ._lib import no_dynamo as _no_dynamo
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .cat_embed import EmbeddingStore
from .two_tower_model import TwoTowerModel
from .two_tower_train_task import TwoTowerTrainTask


class ShardedStore(EmbeddingStore):
    """Local shard of a row-wise sharded global table: global row g lives on rank g % world at g // world."""

    def __init__(self, E: int, global_rows: int, rank: int, world: int, device, grad_mode: str = "sparse", seed: int = 0):
        super().__init__(E, device, grad_mode)
        self.global_rows, self.rank, self.world = global_rows, rank, world
        self.local_rows = (global_rows - rank + world - 1) // world if global_rows > rank else 0
        # nn.Embedding init N(0,1), drawn where the shard lives (a 12.5 M-row shard of the 100 M-row tables is 1.6 GB: the
        # host generator would take tens of seconds per rank); a stream per (seed, rank)
        g = torch.Generator(device=self.device)
        g.manual_seed(seed * 1000003 + rank)
        self.weight = torch.empty((max(self.local_rows, 1), E), dtype=torch.float32, device=self.device).normal_(generator=g)
        self.shard_param = nn.Parameter(self.weight)             # what optimisers / checkpoints see

    def optim_parameters(self):
        return [self.shard_param]

    def bind_grads(self):
        self.shard_param.grad = self.grad

    def load_global(self, global_weight: torch.Tensor):
        """global_weight [global_rows, E] (any device): keep this rank's rows."""
        mine = global_weight[self.rank::self.world]
        self.weight[:mine.shape[0]].copy_(mine.to(self.weight.device))

    def gather_global(self, group=None, comm=None) -> torch.Tensor:
        """all-gather the shards back into the [global_rows, E] table (checkpoint / parity use).  comm: an object with
        DistComm's all_gather instead of torch.distributed on `group` (tests)."""
        per = (self.global_rows + self.world - 1) // self.world
        pad = torch.zeros((per, self.E), dtype=torch.float32, device=self.weight.device)
        pad[:self.local_rows] = self.weight[:self.local_rows]
        if comm is not None:
            parts = list(comm.all_gather(pad).view(self.world, per, self.E))
        else:
            parts = [torch.empty_like(pad) for _ in range(self.world)]
            dist.all_gather(parts, pad, group=group)
        out = torch.empty((per * self.world, self.E), dtype=torch.float32, device=self.weight.device)
        for r, p in enumerate(parts):
            out[r::self.world] = p
        return out[:self.global_rows]


_CONSTS: dict = {}


def _const_i64(dev, n: int, value: int) -> torch.Tensor:
    key = (str(dev), n, value)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.full((n,), value, dtype=torch.int64, device=dev)
    return t


class HipBackend:
    """Compute steps of the exchange on the MI355X (C ABI)."""

    def global_rows(self, sides: Sequence[ops.LookupSide], B: int, E: int, table_rows: int) -> torch.Tensor:
        return ops.embed_lookup(None, [ops.LookupSide(s.ids, s.key_row_offset, s.key_vocab, None, s.K) for s in sides], B,
                                want_rows=True, E=E, table_rows=table_rows, tag="[route]")

    def bucket_by_owner(self, rows: torch.Tensor, world: int):
        owners = torch.remainder(rows, world)                              # int32 [M]
        plan = ops.dedup_plan(owners, world)                               # stable sort by owner
        counts = torch.bincount(owners.long(), minlength=world)            # int64 [world]
        return plan.sorted_src, counts

    def owner_lookup(self, weight: torch.Tensor, local_ids: torch.Tensor, want_plan: bool):
        n, E, dev = local_ids.numel(), weight.shape[1], weight.device
        out = torch.empty((n, E), dtype=torch.float32, device=dev)
        if n == 0:
            return out, None
        off = torch.zeros(1, dtype=torch.int64, device=dev)
        voc = torch.full((1,), weight.shape[0], dtype=torch.int64, device=dev)
        rows = ops.embed_lookup(weight, [ops.LookupSide(local_ids, off, voc, out, 1)], n, want_rows=want_plan)
        return out, (ops.dedup_plan(rows, weight.shape[0]) if want_plan else None)

    def place_rows(self, pooled: torch.Tensor, inv: torch.Tensor, sides: Sequence[ops.LookupSide], B: int):
        """out_side[b, k*E:(k+1)*E] = pooled[inv[slot]]  -- the lookup kernel with `pooled` as the table."""
        M = pooled.shape[0]
        dev = pooled.device
        if pooled.dtype == torch.bfloat16:
            # bf16 rows into bf16 tower inputs: a bit copy -- pairs of bf16 move as one f32 word (E/2 "floats" per row)
            if any(s.out.dtype != torch.bfloat16 for s in sides) or pooled.shape[1] % 2:
                raise ValueError("bf16 pooled rows need bf16 tower inputs and an even embedding dim")
            pooled = pooled.view(torch.float32)
            sides = [ops.LookupSide(s.ids, s.key_row_offset, s.key_vocab, s.out.view(torch.float32), s.K) for s in sides]
        lsides, base = [], 0
        for s in sides:
            n = B * s.K
            off, voc = _const_i64(dev, s.K, 0), _const_i64(dev, s.K, M)      # cached: no fill kernels per step
            lsides.append(ops.LookupSide(inv[base:base + n], off, voc, s.out, s.K))
            base += n
        ops.embed_lookup(pooled, lsides, B, want_rows=False, tag="[place]")

    def collect_grads(self, srcs, order: torch.Tensor, B: int, E: int) -> torch.Tensor:
        """d_bucket[i] = gradient of slot order[i]  (identity-segment plan through tt_embed_grad_bwd)."""
        M, dev = order.numel(), order.device
        ar = torch.arange(M + 1, dtype=torch.int32, device=dev)
        plan = ops.DedupPlan(order, ar[:M], ar, torch.full((1,), M, dtype=torch.int32, device=dev), M)
        out = torch.empty((max(M, 1), E), dtype=torch.float32, device=dev)
        ops.embed_grad(plan, srcs, B, E, ops.TT_GRAD_SPARSE, out)
        return out[:M]

    def owner_accumulate(self, store: ShardedStore, plan, d_rows: torch.Tensor, max_per_row: int = 0):
        # max_per_row: the most contributions a row can get (padded exchange: one per rank); 0 = unknown
        store.accumulate_grad(plan, [(d_rows, 1)], d_rows.shape[0], short_segments=0 < max_per_row <= 64)

    # ---- steps of the fixed-capacity exchange (PaddedRowExchange) --------------------------------------------
    def local_plan(self, rows: torch.Tensor, side_K: Sequence[int], B: int, table_rows: int = 0, key_major: bool = False, E: int = 0):
        """duplicate-row plan of this rank's slots over the GLOBAL row space (table_rows = its size: sets the sort's digit
        count; nothing is read back from the device).  key_major: `rows` are batch_ingest's [key][sample] rows; E > 0: the plan
        also prepares the local gradient reduction's long-row list (row and chunk passes then run as one launch)"""
        if 0 < B <= ops.KEYED_MAX_B:
            return ops.dedup_plan_keyed(rows, list(side_K), B, key_major, E=E)
        if table_rows <= 0:
            raise ValueError("local_plan: table_rows (the global row count) is required above the keyed plan's batch limit")
        return ops.dedup_plan(rows, table_rows)

    def route_bucket(self, plan, G: int, C: int, pad_id: Sequence[int], pad_u: int, overflow: torch.Tensor, expand: bool = False):
        return ops.route_bucket(plan, G, C, pad_id, pad_u, overflow, expand)

    def new_flag(self, device) -> torch.Tensor:
        return torch.zeros(1, dtype=torch.int32, device=device)

    def route_expand(self, plan, pos_u: torch.Tensor) -> torch.Tensor:
        return ops.route_expand(plan, pos_u)

    def gather_rows(self, table: torch.Tensor, idx: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
        """out[i] = table[idx[i]] (idx int32; above the table: last row; negative: a zero row)"""
        return ops.gather_rows(table, idx, out_dtype)

    def owner_plan(self, recv_ids: torch.Tensor, local_rows: int, G: int = 1):
        """plan over the received local row ids -- G ascending runs (every source sends its distinct rows in ascending
        order); the pad value `local_rows` groups into one (last) row that is left out of the plan's row count"""
        return ops.dedup_plan_runs(recv_ids, G, recv_ids.numel() // G, local_rows)

    def reduce_into_buckets(self, plan, pos_u: torch.Tensor, srcs, B: int, E: int, n_rows: int, counters=None, out=None) -> torch.Tensor:
        """[n_rows, E] f32: row pos_u[u] = summed gradient of plan row u; the other rows are not written (`out`: a caller-owned
        buffer whose unwritten rows therefore keep what they held)"""
        import dataclasses
        if out is None:
            out = torch.empty((n_rows, E), dtype=torch.float32, device=pos_u.device)
        ops.embed_grad(dataclasses.replace(plan, unique_rows=pos_u), srcs, B, E, ops.TT_GRAD_DENSE_SET, out, counters=counters)
        return out

    def reduce_local(self, plan, srcs, B: int, E: int, counters=None) -> torch.Tensor:
        """[M, E]: row u = summed gradient of plan row u (unused bucket entries carry u = -1: gather_rows gives them zeros)"""
        out = torch.empty((max(plan.M, 1), E), dtype=torch.float32, device=plan.unique_rows.device)
        ops.embed_grad(plan, srcs, B, E, ops.TT_GRAD_SPARSE, out, counters=counters)
        return out


class DistComm:
    """The collectives the exchanges need, on torch.distributed (RCCL on the GPUs, gloo in the CPU tests).  Tests may pass
    another object with the same four members (an in-process simulator of several ranks on one GPU)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_to_all_equal(self, send: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        out = torch.empty_like(send) if out is None else out
        dist.all_to_all_single(out, send.contiguous(), group=self.group)
        return out

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """[world * n, ...] in rank order"""
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


class HostStagedComm:
    """DistComm's interface over a gloo group with the device tensors staged through the host -- for REHEARSALS in which
    several ranks share one GPU (RCCL refuses two ranks on one device): every kernel of the step is the product path,
    only the wire is not xGMI.  Eager only (a host round trip cannot be captured)."""

    def __init__(self, group=None):
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def all_to_all_equal(self, send):
        s = send.detach().cpu().contiguous()
        out = torch.empty_like(s)
        dist.all_to_all_single(out, s, group=self.group)
        return out.to(send.device)

    def all_reduce_max(self, t):
        c = t.detach().cpu()
        dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
        t.copy_(c)
        return t

    def all_gather(self, t):
        c = t.detach().cpu().contiguous()
        outs = [torch.empty_like(c) for _ in range(self.world)]
        dist.all_gather(outs, c, group=self.group)
        return torch.cat(outs).to(t.device)

    def all_reduce_sum(self, t):
        c = t.detach().cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
        t.copy_(c)
        return t


class RowExchange:
    """Routing of row ids / pooled rows / row gradients between ranks (device-agnostic)."""

    def __init__(self, store: ShardedStore, group=None, backend=None, comm=None):
        self.store, self.group = store, group
        self.comm = comm or DistComm(group)
        self.world, self.rank = self.comm.world, self.comm.rank
        self.backend = backend or HipBackend()
        self.E = store.E

    def _a2a(self, send: torch.Tensor, send_splits: List[int], recv_splits: List[int]) -> torch.Tensor:
        out = torch.empty((sum(recv_splits),) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        dist.all_to_all_single(out, send.contiguous(), output_split_sizes=recv_splits, input_split_sizes=send_splits,
                               group=self.group)
        return out

    def forward(self, sides: Sequence[ops.LookupSide], B: int, want_grad: bool):
        G, be = self.world, self.backend
        rows = be.global_rows(sides, B, self.E, self.store.global_rows)          # int32 [M], slot order
        M = rows.numel()
        order, counts = be.bucket_by_owner(rows, G)                              # slots grouped by owner
        send_ids = torch.div(rows[order.long()], G, rounding_mode="floor").to(torch.int64)
        recv_counts = torch.empty_like(counts)
        dist.all_to_all_single(recv_counts, counts, group=self.group)            # bucket sizes
        send_splits, recv_splits = counts.tolist(), recv_counts.tolist()         # (one host sync per step)
        recv_ids = self._a2a(send_ids, send_splits, recv_splits)
        pooled_local, plan = be.owner_lookup(self.store.weight, recv_ids, want_grad)
        pooled = self._a2a(pooled_local, recv_splits, send_splits)               # [M, E] in bucket order
        inv = torch.empty(M, dtype=torch.int64, device=rows.device)
        inv[order.long()] = torch.arange(M, dtype=torch.int64, device=rows.device)
        be.place_rows(pooled, inv, sides, B)
        return {"order": order, "send": send_splits, "recv": recv_splits, "plan": plan} if want_grad else None

    def backward(self, state, srcs, B: int):
        be = self.backend
        d_bucket = be.collect_grads(srcs, state["order"], B, self.E)
        d_rows = self._a2a(d_bucket, state["send"], state["recv"])               # to the owners
        if state["plan"] is not None:
            be.owner_accumulate(self.store, state["plan"], d_rows)

    def all_reduce_dense(self, flat_grads: Sequence[torch.Tensor]):
        for g in flat_grads:                                                     # one call per tower
            self.comm.all_reduce_sum(g)


class ExchangeOverflowError(RuntimeError):
    """A batch needed more than the calibrated capacity of a (source, owner) bucket of the fixed-capacity exchange: the rows
    that did not fit were fed to the towers as zero rows and their gradients were not sent.  The steps since the last clean
    check are invalid: restore a checkpoint (or accept the few perturbed rows), call `reset_capacity()` -- the next forward
    re-calibrates -- and re-capture the graph if one is in use."""


class PaddedRowExchange(RowExchange):
    """The same routing with (1) the duplicate-row plan BEFORE the exchange -- a rank sends each distinct row once:
    65 k instead of 311 k entries at B = 8192 on the real schema, and a hot row no longer floods one owner -- and
    (2) fixed-capacity buckets: no bucket size ever reaches the host, every all-to-all has equal splits, and the whole
    training step -- collectives included -- can be captured into ONE graph and replayed.

    Capacity C (entries per (source, owner) pair, identical on all ranks) is calibrated on the first forward (one
    host sync): 1.25 x the largest bucket any rank needs, rounded up to 256.  A later batch that needs more sets a sticky
    device-side flag.  Its over-capacity rows are NOT exchanged: the towers see zero rows for them (never another row's
    embedding) and their gradients are dropped, so such a step must be rejected.  The product path does that itself:
    every eager forward and every `GraphedTrainStep.step()` calls `poll_overflow()`, which reads the flag through an
    asynchronous device-to-host copy (no stall; the answer is at most `poll_lag` steps old) and raises
    `ExchangeOverflowError`; `check_overflow()` is the synchronous form (epoch end, checkpoint, `GraphedTrainStep.close()`)."""

    def __init__(self, store: ShardedStore, group=None, backend=None, capacity: Optional[int] = None, slack: float = 1.25, comm=None):
        super().__init__(store, group, backend, comm)
        if store.grad_mode != "sparse":
            raise ValueError("the fixed-capacity exchange needs embedding_grad='sparse' (its bucket pads are skipped by the "
                             "row-sparse Adam; a dense gradient buffer has no row for them)")
        self.C, self.slack = capacity, slack
        self._overflow = None
        self._place_buf = None           # [G * C + 1, E] rows as received; the LAST row stays zero (target of rows that did not fit)
        self._send_buf = None            # [G * C + 1, E] f32 row gradients in bucket order: allocated ZEROED once, so an unused bucket entry
        #                                  never puts uninitialised memory on the wire: it carries zeros or an older step's sum
        self._flag_host, self._flag_event, self._flag_pending = None, None, False
        self.poll_lag = 2                # steps a raised overflow may lag behind the step that caused it
        self.wire_bf16 = True            # looked-up rows travel as bf16 (the towers round them to bf16 anyway: bit-identical)
        self.grad_wire_bf16 = False      # row gradients travel as f32

    def local_rows_of(self, g: int) -> int:
        R = self.store.global_rows
        return (R - g + self.world - 1) // self.world if R > g else 0

    def reset_capacity(self):
        self.C = None
        self._flag_pending = False
        if self._overflow is not None:
            self._overflow.zero_()

    def overflowed(self) -> bool:
        """synchronous read of the sticky flag"""
        return self._overflow is not None and bool(self._overflow.item())

    def check_overflow(self):
        if self.overflowed():
            raise ExchangeOverflowError(f"fixed-capacity exchange: a bucket needed more than C = {self.C} rows (rank {self.rank})")

    def poll_overflow(self):
        """Non-blocking check: looks at the answer of the PREVIOUS poll's device-to-host copy if it has landed, then
        queues the next copy behind the work issued so far.  Never called while a stream is capturing."""
        if self._overflow is None or self._overflow.device.type != "cuda":
            return self.check_overflow()
        if torch.cuda.is_current_stream_capturing():
            return
        if self._flag_host is None:
            self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._flag_event, self._flag_mark = torch.cuda.Event(), torch.cuda.Event()
            self._flag_stream = torch.cuda.Stream(device=self._overflow.device)
        if self._flag_pending and self._flag_event.query():
            self._flag_pending = False
            if int(self._flag_host[0]) != 0:
                raise ExchangeOverflowError(f"fixed-capacity exchange: a bucket needed more than C = {self.C} rows (rank {self.rank}) "
                                            f"within the last {self.poll_lag} steps")
        if not self._flag_pending:
            # the copy runs on its own stream behind a marker of the work issued so far: the next step's launches do not queue
            # behind a device-to-host transfer (5 us per step on the training stream before round 3)
            self._flag_mark.record()
            with torch.cuda.stream(self._flag_stream):
                self._flag_stream.wait_event(self._flag_mark)
                self._flag_host.copy_(self._overflow, non_blocking=True)
                self._flag_event.record()
            self._flag_pending = True

    def _a2a_equal(self, send: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            return self.comm.all_to_all_equal(send)
        if isinstance(self.comm, DistComm):
            return self.comm.all_to_all_equal(send, out)
        return out.copy_(self.comm.all_to_all_equal(send))      # test communicators hand back a fresh tensor

    def _rows_buffer(self, like: torch.Tensor) -> torch.Tensor:
        """[G * C + 1, E] in the wire dtype: the received rows land in the first G * C rows every step, row G * C is
        zeroed once and never written -- tt_route_bucket points every row that did not fit at it."""
        n = self.world * self.C + 1
        b = self._place_buf
        if b is None or b.shape[0] != n or b.dtype != like.dtype or b.device != like.device:
            if like.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("PaddedRowExchange: run one eager forward before capturing (the receive buffer is allocated there)")
            b = self._place_buf = torch.zeros((n, self.E), dtype=like.dtype, device=like.device)
        return b

    def _calibrate(self, plan, dev):
        G = self.world
        probe = max(256, -(-plan.M // 256) * 256)                                 # generous probe capacity: counts only
        _, _, _, counts = self.backend.route_bucket(plan, G, probe, [self.local_rows_of(g) for g in range(G)], -1,
                                                    self.backend.new_flag(dev))
        mx = self.comm.all_reduce_max(counts.max().to(torch.int64).reshape(1))
        need = int(mx.item())                                                     # the one host sync of the exchange
        self.C = max(256, -(-int(need * self.slack) // 256) * 256)

    def forward(self, sides: Sequence[ops.LookupSide], B: int, want_grad: bool):
        G, be, E = self.world, self.backend, self.E
        # a graph-replayed step hands the batch over with ops.batch_ingest, which leaves the GLOBAL fused rows of exactly these id
        # tensors in key-major order (the same id -> row rule): the plan sorts those and the rows-only lookup is not needed
        km = self.store.rows_km_for([s.ids for s in sides]) if (want_grad and 0 < B <= ops.KEYED_MAX_B and hasattr(be, "global_rows")
                                                               and isinstance(be, HipBackend)) else None
        rows = km if km is not None else be.global_rows(sides, B, E, self.store.global_rows)    # int32 [M], slot order (or key-major)
        if isinstance(be, HipBackend):
            plan = be.local_plan(rows, [s.K for s in sides], B, self.store.global_rows, key_major=km is not None, E=E if want_grad else 0)
        else:
            plan = be.local_plan(rows, [s.K for s in sides], B, self.store.global_rows)
        if self.C is None:
            self._calibrate(plan, rows.device)
        pads = [self.local_rows_of(g) for g in range(G)]
        if self._overflow is None:
            self._overflow = be.new_flag(rows.device)
        idx_slot = None
        if isinstance(be, HipBackend):       # routing and the slots' bucket positions from one launch
            send_ids, send_u, pos_u, _counts, idx_slot = be.route_bucket(plan, G, self.C, pads, -1, self._overflow, expand=True)
        else:
            send_ids, send_u, pos_u, _counts = be.route_bucket(plan, G, self.C, pads, -1, self._overflow)
        recv_ids = self._a2a_equal(send_ids)                                     # [G*C] local row ids, pad = my local_rows
        # bf16 tower inputs (mlp_dtype="bf16"): the rows are rounded where they are gathered instead of where they are
        # placed -- the same bits in x, half the bytes on the wire
        wire = torch.bfloat16 if all(s.out.dtype == torch.bfloat16 for s in sides) and E % 8 == 0 and self.wire_bf16 else torch.float32
        pooled_local = be.gather_rows(self.store.weight, recv_ids, wire)         # pads clamp to the last row (unused)
        buf = self._rows_buffer(pooled_local)
        self._a2a_equal(pooled_local, buf[:G * self.C])                          # [G*C, E] in my send order (+ the zero row)
        be.place_rows(buf, idx_slot if idx_slot is not None else be.route_expand(plan, pos_u), sides, B)
        if not want_grad:
            return None
        return {"plan": plan, "send_u": send_u, "pos_u": pos_u, "owner_plan": be.owner_plan(recv_ids, self.store.local_rows, G), "padded": True}

    def backward(self, state, srcs, B: int):
        be = self.backend
        # opt-in (grad_wire_bf16): the per-rank row sums travel as bf16 and are added in f32 by their owner -- half the bytes of
        # the step's largest message, at 2^-9 relative rounding per contribution
        wire = torch.bfloat16 if (self.grad_wire_bf16 and self.E % 8 == 0) else torch.float32
        if isinstance(be, HipBackend) and wire == torch.float32 and state.get("pos_u") is not None:
            # the local reduction writes every distinct row's sum straight at its bucket position (the reduction's dense-set mode
            # with the bucket positions as its row indices): no [M, E] intermediate, no gather launch.  Unused bucket entries keep
            # whatever the buffer held: their ids are pads, which the owner's plan leaves out of its row count -- never read.
            # Rows that did not fit all land on the spare row behind the buckets (the step is rejected anyway).
            n = self.world * self.C + 1
            sb = self._send_buf
            if sb is None or sb.shape[0] != n or sb.device != state["pos_u"].device:
                if state["pos_u"].device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                    sb = None                            # (a capture without an eager step before it: a fresh buffer, as before)
                else:
                    sb = self._send_buf = torch.zeros((n, self.E), dtype=torch.float32, device=state["pos_u"].device)
            send = be.reduce_into_buckets(state["plan"], state["pos_u"], srcs, B, self.E, n, self.store.grad_counters(), out=sb) \
                if isinstance(be, HipBackend) else be.reduce_into_buckets(state["plan"], state["pos_u"], srcs, B, self.E, n, self.store.grad_counters())
            d_rows = self._a2a_equal(send[:self.world * self.C])
        else:
            grad_u = be.reduce_local(state["plan"], srcs, B, self.E, self.store.grad_counters())  # one row per distinct row (+ a zero row)
            d_rows = self._a2a_equal(be.gather_rows(grad_u, state["send_u"], wire))               # to the owners, pads carry zeros
        be.owner_accumulate(self.store, state["owner_plan"], d_rows, self.world)


class _GlobalScoreCEFn(torch.autograd.Function):
    """In-batch negatives over the GLOBAL batch (two_tower_train_task.py:99-134 at B_global = world * B): every rank scores
    its B notice rows against all world*B company rows (and its company rows against all notices).  Collectives: all-gather
    of both towers' outputs (forward) and of the per-row softmax sums (2 floats per pair); NO gradient collective -- the
    weight E * (1/sum_a + 1/sum_b) that the backward kernel applies already holds the terms in which a local row is some
    other rank's negative, because sum_b is the gathered sum of the row's owner.  bf16-operand MFMA path only."""

    @staticmethod
    def forward(ctx, n, c, inv_t, comm):
        n, c = n.contiguous().float(), c.contiguous().float()
        B, D = n.shape
        G, off = comm.world, comm.rank * B
        shift = abs(inv_t)
        n_all, c_all = comm.all_gather(n), comm.all_gather(c)
        Np, Cp = ops.score_pack2_bf16(n, c)
        NpA, CpA = ops.score_pack2_bf16(n_all, c_all)
        rs1, rs2, diag, rk1, rk2, ss = ops.score_fwd_bf16_rect(Np, CpA, Cp, NpA, B, G * B, off, D, inv_t, shift, True)
        out8, loss = ops.score_loss_finish(B, shift, rs1, rs2, diag, rk1, rk2, ss)
        out8 = out8.clone()
        neg = (out8[6] - out8[2] * B) / float(B * G * B - B)      # mean over the off-diagonal of the B x (G*B) block
        out8[3], out8[4] = neg, out8[2] - neg
        ctx.packed = (Np, Cp, NpA, CpA)
        ctx.save_for_backward(rs1, rs2, comm.all_gather(rs1), comm.all_gather(rs2))
        ctx.inv_t, ctx.shift, ctx.dims = inv_t, shift, (B, G * B, off, D)
        ctx.mark_non_differentiable(out8, rk1)
        ctx.set_materialize_grads(False)
        return loss, out8, rk1

    @staticmethod
    def backward(ctx, d_loss, _d8, _dr):
        if d_loss is None:
            return None, None, None, None
        rs1, rs2, rs1_all, rs2_all = ctx.saved_tensors
        B, Rb, off, D = ctx.dims
        Np, Cp, NpA, CpA = ctx.packed
        if d_loss.dtype != torch.float32 or not d_loss.is_contiguous():
            d_loss = d_loss.contiguous().float()
        dN, dC = ops.score_bwd_bf16_rect(Np, CpA, Cp, NpA, B, Rb, off, D, ctx.inv_t, ctx.shift, rs1, rs2_all, rs2, rs1_all, d_loss,
                                         ctx.inv_t / (2.0 * B))
        return dN, dC, None, None


class DistributedTwoTowerTrainTask(TwoTowerTrainTask):
    """TwoTowerTrainTask whose tables are row-wise sharded over the process group."""

    def __init__(self, two_tower_model: TwoTowerModel, store: ShardedStore, group=None, backend=None, exchange: Optional[str] = None,
                 negatives: str = "local", sync_bn: bool = False, comm=None, **kw):
        super().__init__(two_tower_model, **kw)
        if exchange is None:             # the fixed-capacity exchange wherever it applies (row-sparse table gradients)
            exchange = "padded" if store.grad_mode == "sparse" else "exact"
        if negatives not in ("local", "global"):
            raise ValueError(f"negatives must be 'local' or 'global', got {negatives!r}")
        self.negatives = negatives
        self.sync_bn = bool(sync_bn)
        self.sharded_store = store
        self.embedding_shard = store.shard_param                       # registered => in .parameters()
        if exchange not in ("exact", "padded"):
            raise ValueError(f"exchange must be 'exact' or 'padded', got {exchange!r}")
        self.exchange = (PaddedRowExchange if exchange == "padded" else RowExchange)(store, group, backend, comm=comm)
        two_tower_model.notice_tower.exchange = self.exchange
        two_tower_model.company_tower.exchange = self.exchange
        if self.sync_bn and self.exchange.world > 1:
            # BatchNorm statistics over the rows of ALL ranks (with negatives="global": the single-process reference at the
            # global batch); dropout masks are drawn per global row, so give every rank the same torch seed for that
            for tw in (two_tower_model.notice_tower, two_tower_model.company_tower):
                if len(tw.tower_hidden_dims) != 2:
                    raise NotImplementedError("sync_bn needs towers with exactly one hidden block (tower_hidden_dims of length 2, as the "
                                              "reference's [512, 256] and [128, 64]): the pass is cut at that block's BatchNorm")
                tw.sync_comm = self.exchange.comm
        for p in self._dense_parameters():                             # replicas start identical
            dist.broadcast(p.data, src=0, group=group)
        for b in self.buffers():
            if b.is_floating_point():
                dist.broadcast(b, src=0, group=group)

    @_no_dynamo
    def forward(self, batch, return_metrics: bool = False):
        if hasattr(self.exchange, "poll_overflow"):
            self.exchange.poll_overflow()          # rejects a step whose rows did not fit the buckets (non-blocking, <= poll_lag steps late)
        return super().forward(batch, return_metrics)

    def _score_ce(self, n, c, inv_t, first_call):
        if self.negatives == "global" and self.exchange.world > 1:
            if self._dense_loss:
                raise NotImplementedError("global in-batch negatives support the default cross-entropy loss only")
            if self.score_dtype != "bf16":
                raise NotImplementedError("global in-batch negatives run on the bf16 score kernels (score_dtype='bf16')")
            return _GlobalScoreCEFn.apply(n, c, inv_t, self.exchange.comm)
        return super()._score_ce(n, c, inv_t, first_call)

    def _dense_parameters(self):
        return [p for n, p in self.named_parameters() if n != "embedding_shard"]

    def embedding_stores(self):
        return [self.sharded_store]

    def _apply(self, fn, recurse=True):
        super()._apply(fn, recurse)                 # moves embedding_shard like any parameter ...
        st = self.sharded_store                     # ... and the store keeps aliasing it
        st.weight = self.embedding_shard.data
        st.device = st.weight.device
        st.grad, st.sparse_grad = None, None
        st.version += 1
        return self

    def key_directory(self):
        out, base = [], 0
        for side, tower in (("notice", self.two_tower_model.notice_tower), ("company", self.two_tower_model.company_tower)):
            emb = tower.categorical_embedder
            for k in emb.keys:
                out.append((f"two_tower_model.{side}_tower.categorical_embedder.embeddings.{k}.weight", base, emb.vocab_sizes[k]))
                base += emb.vocab_sizes[k]
        return out

    def full_state_dict(self):
        """State dict in the REFERENCE layout (per-key [V_k, E] tables gathered from all ranks)."""
        sd = {k: v for k, v in self.state_dict().items() if k != "embedding_shard"}
        comm = self.exchange.comm
        full = self.sharded_store.gather_global(self.exchange.group, None if isinstance(comm, DistComm) else comm)
        for name, base, v in self.key_directory():
            sd[name] = full[base:base + v].clone()
        return sd

    def load_full_state_dict(self, sd):
        own = {k: v for k, v in sd.items() if "categorical_embedder.embeddings." not in k}
        missing = self.load_state_dict(own, strict=False)
        glob = torch.cat([torch.as_tensor(sd[name]) for name, _, _ in self.key_directory()])
        self.sharded_store.load_global(glob)
        return missing


def create_distributed_train_task(notice_categorical_keys, company_categorical_keys, metadata_path: str = "meta/metadata.csv",
                                  categorical_embedding_dim: int = 64, notice_dense_input_dim: int = 256,
                                  company_dense_input_dim: int = 128, tower_hidden_dims=None, final_embedding_dim: int = 128,
                                  dropout_rate: float = 0.2, temperature: float = 1.0, loss_type: str = "cross_entropy",
                                  device="cuda:0", embedding_grad: Optional[str] = "sparse", score_dtype=None, mlp_dtype=None,
                                  group=None, backend=None, seed: int = 0, exchange: Optional[str] = None,
                                  negatives: str = "local", sync_bn: bool = False, comm=None) -> DistributedTwoTowerTrainTask:
    """Same arguments as create_two_tower_train_task; requires an initialised process group.
    exchange: None = "padded" (dedup-first fixed-capacity all-to-alls, every step of the routing inside the .so, capturable)
    with row-sparse table gradients, "exact" with dense ones.  "exact" (RowExchange: exact bucket sizes, one host sync
    and a few ATen index ops per step, eager only) is the simple implementation the padded one is tested against."""
    if not dist.is_initialized():
        raise RuntimeError("create_distributed_train_task needs torch.distributed.init_process_group first")
    if tower_hidden_dims is None:
        tower_hidden_dims = [256, 128]
    common = dict(metadata_path=metadata_path, categorical_embedding_dim=categorical_embedding_dim,
                  tower_hidden_dims=tower_hidden_dims, final_embedding_dim=final_embedding_dim, dropout_rate=dropout_rate,
                  device=device, embedding_grad=embedding_grad, materialize_tables=False, mlp_dtype=mlp_dtype)
    model = TwoTowerModel(
        notice_tower_config=dict(categorical_keys=notice_categorical_keys, dense_input_dim=notice_dense_input_dim, **common),
        company_tower_config=dict(categorical_keys=company_categorical_keys, dense_input_dim=company_dense_input_dim, **common),
        final_embedding_dim=final_embedding_dim, device=device)
    ne, ce = model.notice_tower.categorical_embedder, model.company_tower.categorical_embedder
    ne.set_row_base(0)
    ce.set_row_base(ne.total_rows)
    store = ShardedStore(categorical_embedding_dim, ne.total_rows + ce.total_rows, dist.get_rank(group),
                         dist.get_world_size(group), torch.device(device), embedding_grad or "sparse", seed)
    return DistributedTwoTowerTrainTask(model, store, group=group, backend=backend, exchange=exchange, negatives=negatives, sync_bn=sync_bn, comm=comm,
                                        temperature=temperature,
                                        loss_type=loss_type, score_dtype=score_dtype)
