"""Thin Python wrappers over the C ABI (one per entry point of include/twotower.h).  They only
translate torch tensors to pointers/sizes and allocate outputs; no arithmetic happens here."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib as L
from .config import settings
from ._lib import TT_BF16, TT_F32, TT_GRAD_DENSE_ACC, TT_GRAD_DENSE_SET, TT_GRAD_SPARSE  # noqa: F401


class KernelTimer:
    """HIP-event timing of C-ABI calls on the stream they are enqueued on (bench.py: roofline legs)."""

    def __init__(self, names=None):
        self.names = set(names) if names else None
        self.records = {}

    def wants(self, name):
        return self.names is None or name in self.names

    def summary(self):
        """name -> (launches, mean ms); synchronises."""
        torch.cuda.synchronize()
        return {n: (len(ev), sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)) for n, ev in self.records.items()}


_TIMER: Optional[KernelTimer] = None
_SYNC_DEBUG = settings.sync_debug


def set_timer(t: Optional[KernelTimer]):
    global _TIMER
    _TIMER = t


class _timed:
    __slots__ = ("name", "a")

    def __init__(self, name):
        self.name = name
        self.a = None

    def __enter__(self):
        if _TIMER is not None and _TIMER.wants(self.name):
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if _SYNC_DEBUG and not torch.cuda.is_current_stream_capturing():     # TT_SYNC_DEBUG=1: name every C-ABI call and wait for it (fault hunting)
            import sys
            print(f"[tt] {self.name} ...", end="", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            print(" done", file=sys.stderr, flush=True)
        if self.a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _TIMER.records.setdefault(self.name, []).append((self.a, b))
        return False


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return TT_F32
    if t.dtype == torch.bfloat16:
        return TT_BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


# ---------------------------------------------------------------------------------------------- lookup
@dataclass
class LookupSide:
    ids: torch.Tensor             # int64 [B*K] sample-major
    key_row_offset: torch.Tensor  # int64 [K] (device)
    key_vocab: torch.Tensor       # int64 [K] (device)
    out: torch.Tensor             # 2-D view [B, K*E] inside the destination buffer (row stride = ld)
    K: int


def embed_lookup(table: Optional[torch.Tensor], sides: Sequence[LookupSide], B: int, want_rows: bool, E: int = 0,
                 table_rows: int = 0, tag: str = "") -> Optional[torch.Tensor]:
    """table None => rows-only mode (E and table_rows then describe the row space)."""
    dev = table.device if table is not None else sides[0].ids.device
    E = table.shape[1] if table is not None else E
    table_rows = table.shape[0] if table is not None else table_rows
    arr = (L.EmbedSide * len(sides))()
    M = 0
    for i, s in enumerate(sides):
        if s.ids.dtype != torch.int64 or not s.ids.is_contiguous() or s.ids.device != dev:
            raise ValueError("ids must be a contiguous int64 tensor on the table's device")
        if s.ids.numel() != B * s.K:
            raise ValueError(f"side {i}: {s.ids.numel()} ids for B={B}, K={s.K}")
        if s.out is not None:
            assert s.out.stride(1) == 1
            arr[i] = L.EmbedSide(L.ptr(s.ids), L.ptr(s.key_row_offset), L.ptr(s.key_vocab), L.ptr(s.out),
                                 s.out.stride(0), s.K, _dt(s.out))
        else:
            arr[i] = L.EmbedSide(L.ptr(s.ids), L.ptr(s.key_row_offset), L.ptr(s.key_vocab), None, s.K * E, s.K, TT_F32)
        M += B * s.K
    rows = torch.empty(M, dtype=torch.int32, device=dev) if want_rows else None
    with _timed("tt_embed_lookup_fwd" + tag):
        L.check(L.load().tt_embed_lookup_fwd(L.ctx(dev), L.ptr(table), table_rows, E, arr, len(sides), B,
                                             L.ptr(rows), L.stream(dev)), "tt_embed_lookup_fwd")
    return rows


def embed_lookup_rows(table: torch.Tensor, rows: torch.Tensor, sides: Sequence[LookupSide], B: int):
    """tt_embed_lookup_rows_fwd: the lookup from precomputed fused rows (int32, slot order: the hand-over launch's `rows_sm`)."""
    dev, E = table.device, table.shape[1]
    arr = (L.EmbedSide * len(sides))()
    M = 0
    for i, s in enumerate(sides):
        assert s.out is not None and s.out.stride(1) == 1
        arr[i] = L.EmbedSide(None, None, None, L.ptr(s.out), s.out.stride(0), s.K, _dt(s.out))
        M += B * s.K
    if rows.dtype != torch.int32 or rows.numel() != M or not rows.is_contiguous() or rows.device != dev:
        raise ValueError("embed_lookup_rows: rows must be a contiguous int32 tensor of sum(B*K) elements on the table's device")
    with _timed("tt_embed_lookup_fwd"):
        L.check(L.load().tt_embed_lookup_rows_fwd(L.ctx(dev), L.ptr(table), table.shape[0], E, arr, len(sides), B, L.ptr(rows), L.stream(dev)),
                "tt_embed_lookup_rows_fwd")


class LookupProfile:
    """Device-clock stamps of the lookup kernel, one block of per-workgroup pairs per launch (works inside captured graphs,
    needs no host synchronisation and no extra launch while measuring; reduced on the host afterwards)."""

    MAX_WG = 8192

    def __init__(self, device, n_slots: int = 256):
        self.device, self.n = torch.device(device), n_slots
        self.ring = torch.zeros(self.MAX_WG + n_slots * self.MAX_WG * 2, dtype=torch.int64, device=self.device)
        L.check(L.load().tt_embed_lookup_set_profile(L.ctx(self.device), L.ptr(self.ring), n_slots), "tt_embed_lookup_set_profile")

    def reset(self):
        self.ring.zero_()

    def durations_us(self, first_wg: int = 0, last_wg: Optional[int] = None):
        """Kernel durations (us) of the (last n_slots) launches since the last reset(), oldest first; synchronises.
        first_wg / last_wg: only the workgroups [first_wg, last_wg) (the fused hand-over + lookup launch: its tiles -- the gather
        phase -- are the first workgroups, the copy roles follow)."""
        torch.cuda.synchronize(self.device)
        r = self.ring.cpu()
        launches = int(r[0])
        pairs = r[self.MAX_WG:].view(self.n, self.MAX_WG, 2)
        same = r[:self.MAX_WG] == launches          # workgroups that took part in every launch (grids of one size)
        if first_wg or last_wg is not None:
            sel = torch.zeros_like(same)
            sel[first_wg:last_wg] = True
            same = same & sel
        out = []
        for n in range(max(0, launches - self.n), launches):
            blk = pairs[n % self.n]
            live = (blk[:, 1] > 0) & same
            if bool(live.any()):
                out.append(float(blk[live, 1].max() - blk[live, 0].min()) * 0.01)   # 100 MHz clock
        return out

    def close(self):
        L.check(L.load().tt_embed_lookup_set_profile(L.ctx(self.device), None, 0), "tt_embed_lookup_set_profile")

    def reopen(self):
        L.check(L.load().tt_embed_lookup_set_profile(L.ctx(self.device), L.ptr(self.ring), self.n), "tt_embed_lookup_set_profile")


@dataclass
class DedupPlan:
    sorted_src: torch.Tensor
    unique_rows: torch.Tensor
    seg_offsets: torch.Tensor
    n_unique: torch.Tensor        # int32 [1] on device
    M: int
    keep: object = None           # tensors that must outlive the plan's kernels
    staging: object = None        # the keyed sort's staging arrays when the compaction is deferred (TT_OPT_DEFER_RIDERS)
    grad_ws: object = None        # (uint8 workspace, E): the gradient reduction's workspace with the long-row list already in it
    finish_deferred: object = None  # grad_rows whose long rows embed_grad left unfinished (adam_fused / embed_grad_finish complete them)


def dedup_plan(rows: torch.Tensor, table_rows: int) -> DedupPlan:
    dev, M = rows.device, rows.numel()
    buf = torch.empty(3 * M + 2, dtype=torch.int32, device=dev)
    plan = DedupPlan(buf[:M], buf[M:2 * M], buf[2 * M:3 * M + 1], buf[3 * M + 1:], M)
    lib = L.load()
    nb = lib.tt_dedup_workspace_bytes(M)
    ws = L.workspace(dev, nb)
    with _timed("tt_dedup_plan"):
        L.check(lib.tt_dedup_plan(L.ctx(dev), L.ptr(rows), M, table_rows, L.ptr(plan.sorted_src), L.ptr(plan.unique_rows),
                                  L.ptr(plan.seg_offsets), L.ptr(plan.n_unique), L.ptr(ws), ws.numel(), L.stream(dev)),
                "tt_dedup_plan")
    return plan


def dedup_plan_runs(rows: torch.Tensor, G: int, C: int, row_limit: int = 0) -> DedupPlan:
    """Plan of G ascending runs of C ids (stable merge; same outputs as dedup_plan on the concatenation).  row_limit > 0:
    ids >= row_limit are pads, grouped last and left out of n_unique."""
    dev, M = rows.device, G * C
    assert rows.numel() == M and rows.dtype == torch.int32
    buf = torch.empty(3 * M + 2, dtype=torch.int32, device=dev)
    plan = DedupPlan(buf[:M], buf[M:2 * M], buf[2 * M:3 * M + 1], buf[3 * M + 1:], M)
    lib = L.load()
    ws = L.workspace(dev, lib.tt_dedup_workspace_bytes(M))
    with _timed("tt_dedup_plan_runs"):
        L.check(lib.tt_dedup_plan_runs(L.ctx(dev), L.ptr(rows), G, C, row_limit, L.ptr(plan.sorted_src), L.ptr(plan.unique_rows),
                                       L.ptr(plan.seg_offsets), L.ptr(plan.n_unique), L.ptr(ws), ws.numel(), L.stream(dev)),
                "tt_dedup_plan_runs")
    return plan


KEYED_MAX_B = 8192


def dedup_plan_keyed(rows: torch.Tensor, side_K: Sequence[int], B: int, key_major: bool = False, E: int = 0) -> DedupPlan:
    """Plan for slot rows straight from embed_lookup (slot = side_base + b*K + k): per-key LDS sorts, 2 launches.
    key_major: `rows` is batch_ingest's [key][sample] array instead (rows[side_base + k*B + b]); same plan.
    E > 0: the plan also prepares embed_grad's long-row list for rows of E floats (tt_dedup_plan_keyed_long): its own gradient
    workspace travels with the plan, and embed_grad then runs its row and chunk passes as one launch."""
    dev, M = rows.device, rows.numel()
    assert M == B * sum(side_K)
    buf = torch.empty(3 * M + 2, dtype=torch.int32, device=dev)
    plan = DedupPlan(buf[:M], buf[M:2 * M], buf[2 * M:3 * M + 1], buf[3 * M + 1:], M)
    lib = L.load()
    nk = sum(side_K)
    if L.riders_deferred(dev, 1):    # the compaction runs later, inside the towers' launch: its staging arrays must outlive whatever
        ws = torch.empty(lib.tt_dedup_keyed_workspace_bytes(M, nk), dtype=torch.uint8, device=dev)      # takes the shared scratch meanwhile
    else:
        ws = L.workspace(dev, lib.tt_dedup_keyed_workspace_bytes(M, nk))
    plan.staging = ws
    ks = (L.i32 * len(side_K))(*side_K)
    if E > 0 and settings.grad_planned:
        gws = torch.empty(lib.tt_embed_grad_workspace_bytes(M, E), dtype=torch.uint8, device=dev)
        with _timed("tt_dedup_plan_keyed"):
            L.check(lib.tt_dedup_plan_keyed_long(L.ctx(dev), L.ptr(rows), int(key_major), ks, len(side_K), B, E, L.ptr(plan.sorted_src),
                                                 L.ptr(plan.unique_rows), L.ptr(plan.seg_offsets), L.ptr(plan.n_unique), L.ptr(gws),
                                                 gws.numel(), L.ptr(ws), ws.numel(), L.stream(dev)), "tt_dedup_plan_keyed_long")
        plan.grad_ws = (gws, E)
        return plan
    fn = lib.tt_dedup_plan_keyed_km if key_major else lib.tt_dedup_plan_keyed
    with _timed("tt_dedup_plan_keyed"):
        L.check(fn(L.ctx(dev), L.ptr(rows), ks, len(side_K), B, L.ptr(plan.sorted_src), L.ptr(plan.unique_rows),
                   L.ptr(plan.seg_offsets), L.ptr(plan.n_unique), L.ptr(ws), ws.numel(), L.stream(dev)),
                "tt_dedup_plan_keyed")
    return plan


def embed_grad(plan: DedupPlan, srcs: Sequence[tuple], B: int, E: int, mode: int, out: torch.Tensor, short_segments: bool = False,
               counters: Optional[torch.Tensor] = None, defer_finish: bool = False):
    """srcs: [(d_out 2-D view [B, K*E], K)].  short_segments: the caller knows no row has many contributions (skips the
    chunk passes; results do not depend on it).  counters: >= 3 int32 words the caller keeps ZERO between calls (allocated
    once, outside any graph capture): the reduction re-zeroes them itself and needs no zeroing launch in front.
    defer_finish (sparse mode, planned workspace; ignored otherwise): the long rows of `out` are left for adam_fused /
    embed_grad_finish -- plan.finish_deferred then holds `out` until one of them has run."""
    if short_segments:
        mode |= L.TT_GRAD_SHORT_SEGMENTS
    dev = out.device
    arr = (L.GradSrc * len(srcs))()
    for i, (d, K) in enumerate(srcs):
        assert d.stride(1) == 1
        arr[i] = L.GradSrc(L.ptr(d), d.stride(0), K, _dt(d))
    lib = L.load()
    nb = lib.tt_embed_grad_workspace_bytes(plan.M, E)
    if plan.grad_ws is not None and plan.grad_ws[1] == E and not short_segments:
        ws = plan.grad_ws[0]                                # the plan left the long-row list in its own workspace
        deferred = defer_finish and mode == L.TT_GRAD_SPARSE and plan.M >= 1
        mode |= L.TT_GRAD_PLANNED | (L.TT_GRAD_DEFER_FINISH if deferred else 0)
    else:
        ws = L.workspace(dev, nb)
        deferred = False
    with _timed("tt_embed_grad_bwd"):
        L.check(lib.tt_embed_grad_bwd(L.ctx(dev), arr, len(srcs), B, E, L.ptr(plan.sorted_src), L.ptr(plan.seg_offsets),
                                      L.ptr(plan.unique_rows), L.ptr(plan.n_unique), plan.M, mode, L.ptr(out),
                                      L.ptr(counters), L.ptr(ws), ws.numel(), L.stream(dev)), "tt_embed_grad_bwd")
    plan.finish_deferred = out if deferred else None


def embed_grad_finish(plan: DedupPlan):
    """Complete a gradient whose long-row finish was deferred (no-op otherwise)."""
    out = plan.finish_deferred
    if out is None:
        return
    ws, E = plan.grad_ws
    dev = out.device
    with _timed("tt_embed_grad_finish"):
        L.check(L.load().tt_embed_grad_finish(L.ctx(dev), E, L.ptr(plan.seg_offsets), plan.M, L.ptr(out), L.ptr(ws), ws.numel(),
                                              L.stream(dev)), "tt_embed_grad_finish")
    plan.finish_deferred = None


# ---------------------------------------------------------------------------------------------- Adam
def adam_dense(p, g, m, v, step, lr, b1, b2, eps, wd, hp_dev=None):
    dev = p.device
    with _timed("tt_adam_dense_step"):
        L.check(L.load().tt_adam_dense_step(L.ctx(dev), L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), step, lr, b1, b2,
                                            eps, wd, L.ptr(hp_dev), L.stream(dev)), "tt_adam_dense_step")


def adam_multi(items, step, lr, b1, b2, eps, wd, hp_dev=None):
    """items: [(p, g, m, v)] float32 contiguous tensors on one device."""
    if not items:
        return
    dev = items[0][0].device
    arr = (L.AdamTensor * len(items))()
    for i, (p, g, m, v) in enumerate(items):
        arr[i] = L.AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel())
    with _timed("tt_adam_multi_step"):
        L.check(L.load().tt_adam_multi_step(L.ctx(dev), arr, len(items), step, lr, b1, b2, eps, wd, L.ptr(hp_dev), L.stream(dev)),
                "tt_adam_multi_step")


def adam_sparse(table, m, v, plan: DedupPlan, grad_rows, step, lr, b1, b2, eps, wd, hp_dev=None):
    dev = table.device
    embed_grad_finish(plan)
    with _timed("tt_sparse_adam_step"):
        L.check(L.load().tt_sparse_adam_step(L.ctx(dev), L.ptr(table), L.ptr(m), L.ptr(v), table.shape[0], table.shape[1],
                                             L.ptr(plan.unique_rows), L.ptr(grad_rows), L.ptr(plan.n_unique), plan.M, step,
                                             lr, b1, b2, eps, wd, L.ptr(hp_dev), L.stream(dev)), "tt_sparse_adam_step")


def adam_fused(items, table, m, v, plan: DedupPlan, grad_rows, step, lr, b1, b2, eps, wd, hp_dev=None):
    """adam_multi(items) + adam_sparse(table rows) with one set of hyper-parameters, one launch."""
    dev = table.device
    arr = (L.AdamTensor * len(items))()
    for i, (p, g, mm, vv) in enumerate(items):
        arr[i] = L.AdamTensor(p.data_ptr(), g.data_ptr(), mm.data_ptr(), vv.data_ptr(), p.numel())
    if plan.finish_deferred is not None:
        if plan.finish_deferred.data_ptr() != grad_rows.data_ptr():
            raise RuntimeError("adam_fused: the plan's deferred gradient is not the one handed to the optimiser")
        ws, E = plan.grad_ws
        with _timed("tt_adam_fused_step"):
            L.check(L.load().tt_adam_fused_step_finish(L.ctx(dev), arr, len(items), L.ptr(table), L.ptr(m), L.ptr(v), table.shape[0],
                                                       table.shape[1], L.ptr(plan.unique_rows), L.ptr(grad_rows), L.ptr(plan.n_unique),
                                                       plan.M, L.ptr(plan.seg_offsets), L.ptr(ws), ws.numel(), step, lr, b1, b2, eps, wd,
                                                       L.ptr(hp_dev), L.stream(dev)), "tt_adam_fused_step_finish")
        plan.finish_deferred = None
        return
    with _timed("tt_adam_fused_step"):
        L.check(L.load().tt_adam_fused_step(L.ctx(dev), arr, len(items), L.ptr(table), L.ptr(m), L.ptr(v), table.shape[0], table.shape[1],
                                            L.ptr(plan.unique_rows), L.ptr(grad_rows), L.ptr(plan.n_unique), plan.M, step,
                                            lr, b1, b2, eps, wd, L.ptr(hp_dev), L.stream(dev)), "tt_adam_fused_step")


# ---------------------------------------------------------------------------------------------- tower MLP
def _fill(arr, tensors):
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else 0


def tower_params_struct(din, h0, kcat_e, hidden, d_out, w_proj, b_proj, ws, bs, bn_w, bn_b, bn_rm, bn_rv, w_out, b_out,
                        bn_nbt=(), compute_dtype=TT_F32, x_dtype=TT_F32, dx_dtype=TT_F32, flags=0, w_proj_bf16=None, w_bf16=()):
    if len(hidden) > L.TT_MAX_HIDDEN:
        raise ValueError(f"at most {L.TT_MAX_HIDDEN} hidden blocks per tower are supported")
    p = L.TowerParams()
    p.din, p.h0, p.kcat_e, p.n_hidden, p.d_out = din, h0, kcat_e, len(hidden), d_out
    for i, h in enumerate(hidden):
        p.hidden[i] = h
    p.w_proj, p.b_proj, p.w_out, p.b_out = w_proj.data_ptr(), b_proj.data_ptr(), w_out.data_ptr(), b_out.data_ptr()
    _fill(p.w, ws); _fill(p.b, bs); _fill(p.bn_w, bn_w); _fill(p.bn_b, bn_b); _fill(p.bn_rm, bn_rm); _fill(p.bn_rv, bn_rv)
    _fill(p.bn_nbt, bn_nbt)
    p.compute_dtype, p.x_dtype, p.dx_dtype, p.flags = compute_dtype, x_dtype, dx_dtype, flags
    p.w_proj_bf16 = w_proj_bf16.data_ptr() if w_proj_bf16 is not None else 0      # bf16 shadows (tt_tower_params.w_proj_bf16 / w_bf16)
    _fill(p.w_bf16, w_bf16)
    return p


def adam_hparams(step, lr, b1, b2, eps, wd):
    out = (L.f32 * 6)()
    L.load().tt_adam_hparams(step, lr, b1, b2, eps, wd, C.byref(out))
    return list(out)


def tower_workspace(params, B, dev):
    nb = L.load().tt_tower_workspace_bytes(C.byref(params), B)
    return L.workspace(dev, nb)


def tower_fwd(params, acts, B, train, p_drop, seed, dev, seed_dev=None):
    ws = tower_workspace(params, B, dev)
    with _timed("tt_tower_mlp_fwd"):
        L.check(L.load().tt_tower_mlp_fwd(L.ctx(dev), C.byref(params), C.byref(acts), B, int(train), p_drop, seed, L.ptr(seed_dev), L.ptr(ws),
                                          ws.numel(), L.stream(dev)), "tt_tower_mlp_fwd")


def tower_bwd(params, acts, d_emb, grads, B, train, p_drop, seed, dev, seed_dev=None):
    ws = tower_workspace(params, B, dev)
    with _timed("tt_tower_mlp_bwd"):
        L.check(L.load().tt_tower_mlp_bwd(L.ctx(dev), C.byref(params), C.byref(acts), L.ptr(d_emb), C.byref(grads), B,
                                          int(train), p_drop, seed, L.ptr(seed_dev), L.ptr(ws), ws.numel(), L.stream(dev)), "tt_tower_mlp_bwd")


def _tower_workspaces(params_list, B, dev):
    """One growable buffer per (device, stream), carved into per-tower regions."""
    lib = L.load()
    sizes = [(lib.tt_tower_workspace_bytes(C.byref(p), B) + 255) // 256 * 256 for p in params_list]
    buf = L.workspace(dev, sum(sizes))
    ptrs, off = (L.vp * len(sizes))(), 0
    for i, z in enumerate(sizes):
        ptrs[i] = buf.data_ptr() + off
        off += z
    return buf, ptrs, (L.sz * len(sizes))(*sizes)


def towers_fwd(params_list, acts_list, B, train, p_drop, seed, dev, seed_dev=None):
    """All towers in one launch per layer step (tt_towers_mlp_fwd)."""
    n = len(params_list)
    _, wptrs, wsizes = _tower_workspaces(params_list, B, dev)
    P = (C.POINTER(L.TowerParams) * n)(*[C.pointer(p) for p in params_list])
    A = (C.POINTER(L.TowerActs) * n)(*[C.pointer(a) for a in acts_list])
    with _timed("tt_towers_mlp_fwd"):
        L.check(L.load().tt_towers_mlp_fwd(L.ctx(dev), n, P, A, B, int(train), p_drop, seed, L.ptr(seed_dev), wptrs, wsizes,
                                           L.stream(dev)), "tt_towers_mlp_fwd")


def towers_bwd(params_list, acts_list, d_embs, grads_list, B, train, p_drop, seed, dev, seed_dev=None):
    n = len(params_list)
    _, wptrs, wsizes = _tower_workspaces(params_list, B, dev)
    P = (C.POINTER(L.TowerParams) * n)(*[C.pointer(p) for p in params_list])
    A = (C.POINTER(L.TowerActs) * n)(*[C.pointer(a) for a in acts_list])
    G = (C.POINTER(L.TowerGrads) * n)(*[C.pointer(g) for g in grads_list])
    D = (L.vp * n)(*[d.data_ptr() for d in d_embs])
    with _timed("tt_towers_mlp_bwd"):
        L.check(L.load().tt_towers_mlp_bwd(L.ctx(dev), n, P, A, D, G, B, int(train), p_drop, seed, L.ptr(seed_dev), wptrs, wsizes,
                                           L.stream(dev)), "tt_towers_mlp_bwd")


# ---------------------------------------------------------------------------------------------- score / loss
def score_dir_fwd(A, Bm, inv_t, shift, diag_offset=0, want_sumscore=True):
    dev, Ra, Rb, D = A.device, A.shape[0], Bm.shape[0], A.shape[1]
    f = torch.empty((3, Ra), dtype=torch.float32, device=dev)       # sumexp, diag, sumscore
    rank = torch.empty(Ra, dtype=torch.int32, device=dev)
    with _timed("tt_score_dir_fwd"):
        L.check(L.load().tt_score_dir_fwd(L.ctx(dev), L.ptr(A), L.ptr(Bm), Ra, Rb, D, inv_t, shift, diag_offset, L.ptr(f[0]),
                                          L.ptr(f[1]), L.ptr(rank), L.ptr(f[2]) if want_sumscore else None, L.stream(dev)),
                "tt_score_dir_fwd")
    return f[0], f[1], rank, f[2]


def score_loss_finish(B, shift, rowsum, colsum, diag, row_rank, col_rank, sumscore):
    """Returns (out8, loss): the loss is its own 0-dim tensor so that autograd sees a plain output."""
    dev = rowsum.device
    out = torch.empty(8, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    with _timed("tt_score_loss_finish"):
        L.check(L.load().tt_score_loss_finish(L.ctx(dev), B, shift, L.ptr(rowsum), L.ptr(colsum), L.ptr(diag), L.ptr(row_rank),
                                              L.ptr(col_rank), L.ptr(sumscore), L.ptr(out), L.ptr(loss), L.stream(dev)),
                "tt_score_loss_finish")
    return out, loss


def score_dir_bwd(A, Bm, inv_t, shift, diag_offset, sumexp_a, sumexp_b, d_loss, scale):
    dev, Ra, Rb, D = A.device, A.shape[0], Bm.shape[0], A.shape[1]
    dA = torch.empty_like(A)
    with _timed("tt_score_dir_bwd"):
        L.check(L.load().tt_score_dir_bwd(L.ctx(dev), L.ptr(A), L.ptr(Bm), Ra, Rb, D, inv_t, shift, diag_offset, L.ptr(sumexp_a),
                                          L.ptr(sumexp_b), L.ptr(d_loss), scale, L.ptr(dA), L.stream(dev)), "tt_score_dir_bwd")
    return dA


def score_pack_bf16(X, scale: float = 1.0):
    """f32 [R, D] -> packed bf16 operand images of scale * X (uint8 buffer)."""
    dev, R, D = X.device, X.shape[0], X.shape[1]
    lib = L.load()
    buf = torch.empty(lib.tt_score_pack_bytes(R, D), dtype=torch.uint8, device=dev)
    with _timed("tt_score_pack_bf16"):
        L.check(lib.tt_score_pack_bf16(L.ctx(dev), L.ptr(X), R, D, scale, L.ptr(buf), L.stream(dev)), "tt_score_pack_bf16")
    return buf


def score_unit_scale(inv_t: float) -> float:
    """inv_t * log2(e) exactly as the library computes it: packing ONE tower's operand images with it lets the score
    kernels drop the exponent's multiply-add (tt_score_fwd_dir.ab_scale)."""
    return float(L.load().tt_score_unit_scale(inv_t))


def score_pack2_bf16(X0, X1, scale0: float = 1.0, scale1: float = 1.0):
    """Both operands of a step in one launch (images of scale * X)."""
    dev, D = X0.device, X0.shape[1]
    lib = L.load()
    b0 = torch.empty(lib.tt_score_pack_bytes(X0.shape[0], D), dtype=torch.uint8, device=dev)
    b1 = torch.empty(lib.tt_score_pack_bytes(X1.shape[0], D), dtype=torch.uint8, device=dev)
    with _timed("tt_score_pack_bf16"):
        L.check(lib.tt_score_pack2_bf16(L.ctx(dev), L.ptr(X0), X0.shape[0], L.ptr(b0), L.ptr(X1), X1.shape[0], L.ptr(b1), D,
                                        scale0, scale1, L.stream(dev)), "tt_score_pack2_bf16")
    return b0, b1


def score_fwd_bf16(Np, Cp, B, D, inv_t, shift, want_col_rank=True, full_rank=True, scale_n: float = 1.0, with_inv: bool = False):
    """Both softmax directions of the square in-batch problem in one launch.  scale_n: the scale Np was packed with.
    with_inv: also return (inv_row, inv_col), the reciprocals score_bwd_bf16 can take instead of computing its own."""
    dev = Np.device
    Bp = (B + 63) // 64 * 64                                           # tt_score_bwd_bf16 reads its per-row arrays a 32-row tile at a time
    f = (torch.empty if B == Bp else torch.ones)((6, Bp), dtype=torch.float32, device=dev)    # rowsum, colsum, diag, sumscore, 1/rowsum', 1/colsum'
    ranks = torch.empty((2, B), dtype=torch.int32, device=dev)
    arr = (L.ScoreFwdDir * 2)()
    rm = 2 if full_rank else 1
    arr[0] = L.ScoreFwdDir(L.ptr(Np), L.ptr(Cp), B, B, 0, L.ptr(f[0]), L.ptr(f[2]), L.ptr(ranks[0]), L.ptr(f[3]), rm, scale_n,
                           L.ptr(f[4]) if with_inv else None)
    arr[1] = L.ScoreFwdDir(L.ptr(Cp), L.ptr(Np), B, B, 0, L.ptr(f[1]), None, L.ptr(ranks[1]) if want_col_rank else None, None, rm, scale_n,
                           L.ptr(f[5]) if with_inv else None)
    with _timed("tt_score_fwd_bf16"):
        L.check(L.load().tt_score_fwd_bf16(L.ctx(dev), arr, 2, D, inv_t, shift, L.stream(dev)), "tt_score_fwd_bf16")
    out = (f[0][:B], f[1][:B], f[2][:B], ranks[0], ranks[1], f[3][:B])
    return out + ((f[4][:B], f[5][:B]),) if with_inv else out


def score_pack2_fp8(X0, X1, scale0: float = 1.0, scale1: float = 1.0):
    """Both operands of a step as [fp8 rows image | bf16 fragment image] (tt_score_pack2_fp8)."""
    dev, D = X0.device, X0.shape[1]
    lib = L.load()
    b0 = torch.empty(lib.tt_score_pack_fp8_bytes(X0.shape[0], D), dtype=torch.uint8, device=dev)
    b1 = torch.empty(lib.tt_score_pack_fp8_bytes(X1.shape[0], D), dtype=torch.uint8, device=dev)
    with _timed("tt_score_pack2_fp8"):
        L.check(lib.tt_score_pack2_fp8(L.ctx(dev), L.ptr(X0), X0.shape[0], L.ptr(b0), L.ptr(X1), X1.shape[0], L.ptr(b1), D,
                                       scale0, scale1, L.stream(dev)), "tt_score_pack2_fp8")
    return b0, b1


def score_fwd_sym(Np, Cp, B, D, inv_t, shift, scale_n: float = 1.0, want_rank: bool = True, fp8: bool = False):
    """Single-pass symmetric forward of the square problem (tt_score_fwd_sym_bf16 / _fp8): returns (rowsum, colsum, diag, row_rank,
    (inv_row, inv_col), out8, loss) -- loss its own 0-dim tensor so that autograd sees a plain output."""
    dev = Np.device
    Bp = (B + 63) // 64 * 64                                           # the kernel fills the entries past B (read by tt_score_bwd_bf16)
    f = torch.empty((5, Bp), dtype=torch.float32, device=dev)         # rowsum, colsum, diag, 1/rowsum', 1/colsum'
    rank = torch.empty(B, dtype=torch.int32, device=dev)
    out8 = torch.empty(8, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    lib = L.load()
    if L.riders_deferred(dev, 2):    # the last reduction runs later (inside the towers' backward launch): its partial records must
        ws = torch.empty(lib.tt_score_fwd_sym_workspace_bytes(B, D), dtype=torch.uint8, device=dev)     # outlive the shared scratch's next user
        out8._tt_keep = ws
    else:
        ws = L.workspace(dev, lib.tt_score_fwd_sym_workspace_bytes(B, D))
    fn, name = (lib.tt_score_fwd_sym_fp8, "tt_score_fwd_sym_fp8") if fp8 else (lib.tt_score_fwd_sym_bf16, "tt_score_fwd_sym_bf16")
    with _timed(name):
        L.check(fn(L.ctx(dev), L.ptr(Np), L.ptr(Cp), B, D, inv_t, shift, scale_n, int(want_rank), L.ptr(f[0]), L.ptr(f[1]),
                                          L.ptr(f[3]), L.ptr(f[4]), L.ptr(f[2]), L.ptr(rank), L.ptr(out8), L.ptr(loss), L.ptr(ws), ws.numel(),
                                          L.stream(dev)), name)
    return f[0][:B], f[1][:B], f[2][:B], rank, (f[3][:B], f[4][:B]), out8, loss


def score_bwd_bf16(Np, Cp, B, D, inv_t, shift, rowsum, colsum, d_loss, scale, scale_n: float = 1.0, inv=None, fp8: bool = False):
    """inv: (inv_row, inv_col) from score_fwd_bf16(..., with_inv=True) with the same scale_n (optional).
    fp8: the operands are tt_score_pack2_fp8 buffers (tt_score_bwd_fp8)."""
    dev = Np.device
    dN = torch.empty((B, D), dtype=torch.float32, device=dev)
    dC = torch.empty((B, D), dtype=torch.float32, device=dev)
    arr = (L.ScoreBwdDir * 2)()
    ir, ic = (L.ptr(inv[0]), L.ptr(inv[1])) if inv is not None else (None, None)
    arr[0] = L.ScoreBwdDir(L.ptr(Np), L.ptr(Cp), B, B, 0, L.ptr(rowsum), L.ptr(colsum), L.ptr(dN), scale_n, 1.0, ir, ic)   # B = company: unscaled
    arr[1] = L.ScoreBwdDir(L.ptr(Cp), L.ptr(Np), B, B, 0, L.ptr(colsum), L.ptr(rowsum), L.ptr(dC), scale_n, scale_n, ic, ir)  # B = notice image
    fn, name = (L.load().tt_score_bwd_fp8, "tt_score_bwd_fp8") if fp8 else (L.load().tt_score_bwd_bf16, "tt_score_bwd_bf16")
    with _timed(name):
        L.check(fn(L.ctx(dev), arr, 2, D, inv_t, shift, L.ptr(d_loss), scale, L.stream(dev)), name)
    return dN, dC


def score_fwd_bf16_rect(Ap0, Bp0, Ap1, Bp1, Ra, Rb, off, D, inv_t, shift, full_rank=True):
    """Rectangular form (global in-batch negatives): direction 0 = rows of A0 [Ra] against all rows of B0 [Rb], direction 1 =
    rows of A1 [Ra] against B1 [Rb]; the positive of local row a sits at column a + off.  Returns (sumexp0, sumexp1, diag,
    rank0, rank1, sumscore0)."""
    dev = Ap0.device
    Rp = (Ra + 3) // 4 * 4
    f = torch.empty((4, Rp), dtype=torch.float32, device=dev)
    ranks = torch.empty((2, Ra), dtype=torch.int32, device=dev)
    arr = (L.ScoreFwdDir * 2)()
    rm = 2 if full_rank else 1
    arr[0] = L.ScoreFwdDir(L.ptr(Ap0), L.ptr(Bp0), Ra, Rb, off, L.ptr(f[0]), L.ptr(f[2]), L.ptr(ranks[0]), L.ptr(f[3]), rm)
    arr[1] = L.ScoreFwdDir(L.ptr(Ap1), L.ptr(Bp1), Ra, Rb, off, L.ptr(f[1]), None, L.ptr(ranks[1]), None, rm)
    with _timed("tt_score_fwd_bf16"):
        L.check(L.load().tt_score_fwd_bf16(L.ctx(dev), arr, 2, D, inv_t, shift, L.stream(dev)), "tt_score_fwd_bf16")
    return f[0][:Ra], f[1][:Ra], f[2][:Ra], ranks[0], ranks[1], f[3][:Ra]


def _tile_padded(t: torch.Tensor) -> torch.Tensor:
    """per-row float array readable a whole 32-row tile at a time (tt_score_bwd_dir): ragged sizes get a padded copy"""
    n = t.numel()
    if n % 32 == 0 and t.is_contiguous():
        return t
    out = torch.ones((n + 31) // 32 * 32, dtype=t.dtype, device=t.device)
    out[:n] = t
    return out


def score_bwd_bf16_rect(Ap0, Bp0, Ap1, Bp1, Ra, Rb, off, D, inv_t, shift, sa0, sb0, sa1, sb1, d_loss, scale):
    """dA0 [Ra, D], dA1 [Ra, D]: sa* = sum-exp of the A rows (this direction), sb* = sum-exp of the B rows in the OTHER
    direction (all Rb of them: gathered from their owners)."""
    dev = Ap0.device
    sb0, sb1 = _tile_padded(sb0), _tile_padded(sb1)
    d0 = torch.empty((Ra, D), dtype=torch.float32, device=dev)
    d1 = torch.empty((Ra, D), dtype=torch.float32, device=dev)
    arr = (L.ScoreBwdDir * 2)()
    arr[0] = L.ScoreBwdDir(L.ptr(Ap0), L.ptr(Bp0), Ra, Rb, off, L.ptr(sa0), L.ptr(sb0), L.ptr(d0))
    arr[1] = L.ScoreBwdDir(L.ptr(Ap1), L.ptr(Bp1), Ra, Rb, off, L.ptr(sa1), L.ptr(sb1), L.ptr(d1))
    with _timed("tt_score_bwd_bf16"):
        L.check(L.load().tt_score_bwd_bf16(L.ctx(dev), arr, 2, D, inv_t, shift, L.ptr(d_loss), scale, L.stream(dev)),
                "tt_score_bwd_bf16")
    return d0, d1


def score_matrix(A, Bm, inv_t):
    dev, Ra, Rb, D = A.device, A.shape[0], Bm.shape[0], A.shape[1]
    S = torch.empty((Ra, Rb), dtype=torch.float32, device=dev)
    with _timed("tt_score_matrix"):
        L.check(L.load().tt_score_matrix(L.ctx(dev), L.ptr(A), L.ptr(Bm), Ra, Rb, D, inv_t, L.ptr(S), Rb, L.stream(dev)),
                "tt_score_matrix")
    return S


def score_dense_fwd(n, c, inv_t: float, loss_type: int, label_smoothing: float):
    """Dense loss path (tt_score_dense_fwd): returns (S [B, B], stats, out8, loss[1]) -- S and stats feed score_dense_bwd."""
    dev, B, D = n.device, n.shape[0], n.shape[1]
    S = torch.empty((B, B), dtype=torch.float32, device=dev)
    stats = torch.empty(6 * B, dtype=torch.float32, device=dev)
    hit = torch.empty(2 * B, dtype=torch.int32, device=dev)
    out8 = torch.empty(8, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    with _timed("tt_score_dense_fwd"):
        L.check(L.load().tt_score_dense_fwd(L.ctx(dev), L.ptr(n), L.ptr(c), B, D, inv_t, loss_type, label_smoothing, L.ptr(S), L.ptr(stats),
                                            L.ptr(hit), L.ptr(out8), L.ptr(loss), L.stream(dev)), "tt_score_dense_fwd")
    return S, stats, out8, loss


def score_dense_bwd(n, c, inv_t: float, loss_type: int, label_smoothing: float, S, stats, d_loss):
    """(dN, dC) of the dense loss path; S is overwritten by d loss / d (N C^T)."""
    dev, B, D = n.device, n.shape[0], n.shape[1]
    dN, dC = torch.empty_like(n), torch.empty_like(c)
    ws = torch.empty(L.load().tt_score_dense_workspace_bytes(B, D), dtype=torch.uint8, device=dev)
    with _timed("tt_score_dense_bwd"):
        L.check(L.load().tt_score_dense_bwd(L.ctx(dev), L.ptr(n), L.ptr(c), B, D, inv_t, loss_type, label_smoothing, L.ptr(S), L.ptr(stats),
                                            L.ptr(d_loss), L.ptr(dN), L.ptr(dC), L.ptr(ws), ws.numel(), L.stream(dev)), "tt_score_dense_bwd")
    return dN, dC


def diag_rank_rows(S, diag_offset=0):
    dev, R, Cc = S.device, S.shape[0], S.shape[1]
    assert S.stride(1) == 1 and S.dtype == torch.float32
    rank = torch.empty(R, dtype=torch.int32, device=dev)
    L.check(L.load().tt_diag_rank_rows(L.ctx(dev), L.ptr(S), R, Cc, S.stride(0), diag_offset, L.ptr(rank), L.stream(dev)),
            "tt_diag_rank_rows")
    return rank


def topk_rows(S, k):
    dev, R, Cc = S.device, S.shape[0], S.shape[1]
    assert S.stride(1) == 1
    vals = torch.empty((R, k), dtype=torch.float32, device=dev)
    idx = torch.empty((R, k), dtype=torch.int64, device=dev)
    with _timed("tt_topk_rows"):
        L.check(L.load().tt_topk_rows(L.ctx(dev), L.ptr(S), R, Cc, S.stride(0), k, L.ptr(vals), L.ptr(idx), L.stream(dev)),
                "tt_topk_rows")
    return vals, idx


# ---------------------------------------------------------------------------------------------- misc
def linear_fwd(X, W, bias, relu=False):
    dev, M, K, N = X.device, X.shape[0], X.shape[1], W.shape[0]
    assert X.stride(1) == 1 and W.is_contiguous()
    Y = torch.empty((M, N), dtype=torch.float32, device=dev)
    L.check(L.load().tt_linear_fwd(L.ctx(dev), L.ptr(X), X.stride(0), L.ptr(W), L.ptr(bias), L.ptr(Y), N, M, N, K, int(relu),
                                   L.stream(dev)), "tt_linear_fwd")
    return Y


def copy_multi(pairs):
    """pairs: [(dst, src)] contiguous same-sized tensors on one device -> one launch."""
    if not pairs:
        return
    dev, n = pairs[0][0].device, len(pairs)
    dst = (L.vp * n)(*[d.data_ptr() for d, _ in pairs])
    src = (L.vp * n)(*[s.data_ptr() for _, s in pairs])
    nb = (L.i64 * n)(*[d.numel() * d.element_size() for d, _ in pairs])
    for d, s_ in pairs:
        if d.numel() * d.element_size() != s_.numel() * s_.element_size() or not d.is_contiguous() or not s_.is_contiguous():
            raise ValueError("copy_multi: segments must be contiguous and equally sized")
    with _timed("tt_copy_multi"):
        L.check(L.load().tt_copy_multi(L.ctx(dev), n, dst, src, nb, L.stream(dev)), "tt_copy_multi")


def ingest_lookup_supported(table: Optional[torch.Tensor], sides: Sequence[LookupSide]) -> bool:
    """Shapes the fused hand-over + lookup launch takes (tt_batch_ingest_lookup): E in {8, 16, 32, 64}, 8-element aligned outputs."""
    if table is None or table.dtype != torch.float32 or table.dim() != 2 or table.shape[1] not in (8, 16, 32, 64) or table.data_ptr() % 16:
        return False
    for s_ in sides:
        o = s_.out
        if o is None or o.dtype not in (torch.float32, torch.bfloat16) or o.stride(1) != 1 or o.stride(0) % 8 or \
                o.data_ptr() % (8 * o.element_size()) or not (1 <= s_.K <= 64):
            return False
    return True


def ingest_lookup_tiles(B: int, side_K: Sequence[int]) -> int:
    """Tile workgroups of the fused hand-over + lookup launch (csrc/tt_embed.hip fill_lookup_part: TS = 64 samples, halved until
    TS * K <= 512 slots, not below 8)."""
    n = 0
    for K in side_K:
        sh = 6
        while sh > 3 and (K << sh) > 512:
            sh -= 1
        n += (B + (1 << sh) - 1) >> sh
    return n


def _cvt_list(cvt):
    """[(dst bf16, src f32)] -> ctypes tt_cvt_list pointer (or None): conversions that ride in the hand-over launch."""
    if not cvt:
        return None
    if len(cvt) > L.TT_MAX_CVT:
        raise ValueError(f"at most {L.TT_MAX_CVT} conversions per hand-over launch")
    c = L.CvtList()
    c.n = len(cvt)
    for i, (d, s_) in enumerate(cvt):
        if d.dtype != torch.bfloat16 or s_.dtype != torch.float32 or d.numel() != s_.numel() or not d.is_contiguous() or not s_.is_contiguous() \
                or d.device != s_.device:
            raise ValueError("cvt: (bf16 destination, f32 source) pairs of contiguous tensors of one size on one device")
        c.src[i], c.dst[i], c.count[i] = s_.data_ptr(), d.data_ptr(), s_.numel()
    return C.byref(c)


def _lookup_part(table: torch.Tensor):
    return L.IngestLookup(L.ptr(table), table.shape[0], table.shape[1], 0)


def _embed_sides(sides, B, dev, who, with_ids: bool, with_out: bool):
    arr = (L.EmbedSide * len(sides))()
    M = 0
    for i, s in enumerate(sides):
        if with_ids and (s.ids.dtype != torch.int64 or not s.ids.is_contiguous() or s.ids.device != dev or s.ids.numel() != B * s.K):
            raise ValueError(f"{who}: side {i} needs {B}*{s.K} contiguous int64 ids on {dev}")
        if with_out:
            if s.out.shape[0] != B or s.out.shape[1] < s.K or s.out.device != dev:
                raise ValueError(f"{who}: side {i}: output view must be [B, K*E] on {dev}")
            arr[i] = L.EmbedSide(L.ptr(s.ids) if with_ids else None, L.ptr(s.key_row_offset), L.ptr(s.key_vocab), L.ptr(s.out), s.out.stride(0),
                                 s.K, _dt(s.out))
        else:
            arr[i] = L.EmbedSide(L.ptr(s.ids) if with_ids else None, L.ptr(s.key_row_offset), L.ptr(s.key_vocab), None, 0, s.K, TT_F32)
        M += B * s.K
    return arr, M


def batch_ingest(pairs, sides: Sequence[LookupSide], B: int, rows_km: Optional[torch.Tensor], table: Optional[torch.Tensor] = None,
                 rows_sm: Optional[torch.Tensor] = None, cvt=None, table_rows: int = 0):
    """copy_multi's segments plus, per side, the fused rows of the side's ids in key-major order (tt_batch_ingest): the batch
    hand-over of a graph-replayed step in one launch.  sides[i].ids is the id source (the incoming batch or the static buffer).
    table given (and sides[i].out set): the same launch also looks the rows up and writes them into sides[i].out -- the towers'
    input (tt_batch_ingest_lookup); rows_km may then be None.  rows_sm (without `table`): the same fused rows also in slot order,
    the input of embed_lookup_rows.  table_rows (without `table`): the size of the row space the fused rows index -- a row outside
    it is stored as the last row and raises the device error word (_lib.check_device_errors), so neither embed_lookup_rows nor the
    plan's consumers touch memory outside the table; 0 = unchecked."""
    if table is not None:
        dev, n = table.device, len(pairs)
        dst = (L.vp * max(n, 1))(*[d.data_ptr() for d, _ in pairs])
        src = (L.vp * max(n, 1))(*[s.data_ptr() for _, s in pairs])
        nb = (L.i64 * max(n, 1))(*[d.numel() * d.element_size() for d, _ in pairs])
        for d, s_ in pairs:
            if d.numel() * d.element_size() != s_.numel() * s_.element_size() or not d.is_contiguous() or not s_.is_contiguous():
                raise ValueError("batch_ingest: segments must be contiguous and equally sized")
        arr, M = _embed_sides(sides, B, dev, "batch_ingest", True, True)
        if rows_km is not None and (rows_km.dtype != torch.int32 or rows_km.numel() != M or not rows_km.is_contiguous()):
            raise ValueError("batch_ingest: rows_km must be a contiguous int32 tensor of sum(B*K) elements")
        lk = _lookup_part(table)
        with _timed("tt_batch_ingest_lookup"):
            L.check(L.load().tt_batch_ingest_lookup(L.ctx(dev), n, dst, src, nb, arr, len(sides), B, L.ptr(rows_km), C.byref(lk), _cvt_list(cvt),
                                                    L.stream(dev)), "tt_batch_ingest_lookup")
        return
    dev, n = rows_km.device, len(pairs)
    dst = (L.vp * max(n, 1))(*[d.data_ptr() for d, _ in pairs])
    src = (L.vp * max(n, 1))(*[s.data_ptr() for _, s in pairs])
    nb = (L.i64 * max(n, 1))(*[d.numel() * d.element_size() for d, _ in pairs])
    for d, s_ in pairs:
        if d.numel() * d.element_size() != s_.numel() * s_.element_size() or not d.is_contiguous() or not s_.is_contiguous():
            raise ValueError("batch_ingest: segments must be contiguous and equally sized")
    arr = (L.EmbedSide * len(sides))()
    M = 0
    for i, s in enumerate(sides):
        if s.ids.dtype != torch.int64 or not s.ids.is_contiguous() or s.ids.device != dev or s.ids.numel() != B * s.K:
            raise ValueError(f"batch_ingest: side {i} needs {B}*{s.K} contiguous int64 ids on {dev}")
        arr[i] = L.EmbedSide(L.ptr(s.ids), L.ptr(s.key_row_offset), L.ptr(s.key_vocab), None, 0, s.K, TT_F32)
        M += B * s.K
    if rows_km.dtype != torch.int32 or rows_km.numel() != M or not rows_km.is_contiguous():
        raise ValueError("batch_ingest: rows_km must be a contiguous int32 tensor of sum(B*K) elements")
    if rows_sm is not None and (rows_sm.dtype != torch.int32 or rows_sm.numel() != M or not rows_sm.is_contiguous()):
        raise ValueError("batch_ingest: rows_sm must be a contiguous int32 tensor of sum(B*K) elements")
    with _timed("tt_batch_ingest"):
        L.check(L.load().tt_batch_ingest(L.ctx(dev), n, dst, src, nb, arr, len(sides), B, L.ptr(rows_km), L.ptr(rows_sm), int(table_rows),
                                         _cvt_list(cvt), L.stream(dev)), "tt_batch_ingest")


@dataclass
class StoreSide:
    """One tower's device-resident feature store as the source of a batch (tt_store_side)."""
    entity: torch.Tensor          # int64: entity index per pair, element [o * entity_stride] (a column of the [P, 2] pair list)
    entity_stride: int
    dense_store: torch.Tensor     # f32 [N, dense_dim]
    cat_store: torch.Tensor       # int64 [N, K]
    dense_out: torch.Tensor       # f32 [B, dense_dim]  (the step's static dense buffer)
    ids_out: torch.Tensor         # int64 [B * K]       (the step's static id buffer)


def batch_ingest_store(pairs, sides: Sequence[LookupSide], stores: Sequence[StoreSide], B: int, order: Optional[torch.Tensor],
                       rows_km: Optional[torch.Tensor], order_offset: int = 0, table: Optional[torch.Tensor] = None,
                       rows_sm: Optional[torch.Tensor] = None, cvt=None, table_rows: int = 0):
    """tt_batch_ingest_store: the batch `order[order_offset : order_offset + B]` of the pair list gathered out of the device
    stores straight into the step's static buffers (+ key-major fused rows, + the copy segments `pairs`), one launch.
    table given (and sides[i].out set): the launch also looks the batch's rows up into sides[i].out (tt_batch_ingest_store_lookup).
    Without `order` the entity view must hold the batch (B entries at its stride); a pair list or offset that does not is a
    ValueError here, not an out-of-bounds read on the device (the reference raises IndexError / KeyError:
    unified_bid_data_loader.py:495-498); entity indices are clamped into the store by the kernel (tt_store_side.n_rows)."""
    dev, n = stores[0].dense_out.device, len(pairs)
    for i, t in enumerate(stores):
        if t.entity.dtype != torch.int64 or t.entity.dim() != 1 or t.entity_stride < 1 or t.entity.device != dev:
            raise ValueError(f"batch_ingest_store: store {i}: entity must be a 1-D int64 view on {dev}")
        need = (B - 1) * t.entity_stride + 1 if order is None else 1
        if t.entity.numel() < need:
            raise ValueError(f"batch_ingest_store: store {i}: the batch runs past the pair list ({t.entity.numel()} entries, {need} needed)")
    dst = (L.vp * max(n, 1))(*[d.data_ptr() for d, _ in pairs])
    src = (L.vp * max(n, 1))(*[s.data_ptr() for _, s in pairs])
    nb = (L.i64 * max(n, 1))(*[d.numel() * d.element_size() for d, _ in pairs])
    arr = (L.EmbedSide * len(sides))()
    st = (L.StoreSide * len(sides))()
    M = 0
    for i, (s, t) in enumerate(zip(sides, stores)):
        dd = t.dense_store.shape[1]
        if t.cat_store.dtype != torch.int64 or t.cat_store.shape[1] != s.K or not t.cat_store.is_contiguous() or \
                t.dense_store.dtype != torch.float32 or not t.dense_store.is_contiguous() or t.entity.dtype != torch.int64:
            raise ValueError(f"batch_ingest_store: store {i}: need contiguous f32 [N, D] features and int64 [N, {s.K}] ids")
        if t.dense_out.shape != (B, dd) or not t.dense_out.is_contiguous() or t.dense_out.dtype != torch.float32 or \
                t.ids_out.numel() != B * s.K or t.ids_out.dtype != torch.int64 or not t.ids_out.is_contiguous():
            raise ValueError(f"batch_ingest_store: side {i}: static buffers must be contiguous f32 [{B}, {dd}] and int64 [{B * s.K}]")
        if table is not None:
            if s.out.shape[0] != B or s.out.device != dev:
                raise ValueError(f"batch_ingest_store: side {i}: output view must be [B, K*E] on {dev}")
            arr[i] = L.EmbedSide(None, L.ptr(s.key_row_offset), L.ptr(s.key_vocab), L.ptr(s.out), s.out.stride(0), s.K, _dt(s.out))
        else:
            arr[i] = L.EmbedSide(None, L.ptr(s.key_row_offset), L.ptr(s.key_vocab), None, 0, s.K, TT_F32)
        st[i] = L.StoreSide(L.ptr(t.entity), t.entity_stride, L.ptr(t.dense_store), L.ptr(t.cat_store), L.ptr(t.dense_out), L.ptr(t.ids_out), dd,
                            min(int(t.dense_store.shape[0]), 2 ** 31 - 1))
        M += B * s.K
    if rows_km is not None and (rows_km.dtype != torch.int32 or rows_km.numel() != M or not rows_km.is_contiguous()):
        raise ValueError("batch_ingest_store: rows_km must be a contiguous int32 tensor of sum(B*K) elements")
    if order is not None:
        if order.dtype != torch.int64 or not order.is_contiguous() or order_offset < 0 or order_offset + B > order.numel():
            raise ValueError("batch_ingest_store: order must be a contiguous int64 tensor holding the batch's B entries")
        order_ptr = L.vp(order.data_ptr() + 8 * order_offset)
    else:
        order_ptr = L.vp(0)
    if table is not None:
        lk = _lookup_part(table)
        with _timed("tt_batch_ingest_lookup"):
            L.check(L.load().tt_batch_ingest_store_lookup(L.ctx(dev), n, dst, src, nb, arr, st, len(sides), B, order_ptr, L.ptr(rows_km),
                                                          C.byref(lk), _cvt_list(cvt), L.stream(dev)), "tt_batch_ingest_store_lookup")
        return
    if rows_sm is not None and (rows_sm.dtype != torch.int32 or rows_sm.numel() != M or not rows_sm.is_contiguous()):
        raise ValueError("batch_ingest_store: rows_sm must be a contiguous int32 tensor of sum(B*K) elements")
    with _timed("tt_batch_ingest_store"):
        L.check(L.load().tt_batch_ingest_store(L.ctx(dev), n, dst, src, nb, arr, st, len(sides), B, order_ptr, L.ptr(rows_km), L.ptr(rows_sm),
                                               int(table_rows), _cvt_list(cvt), L.stream(dev)), "tt_batch_ingest_store")


# ---------------------------------------------------------------------------------------------- multi-GPU routing
def route_bucket(plan: DedupPlan, G: int, C: int, pad_id: Sequence[int], pad_u: int, overflow: torch.Tensor, expand: bool = False):
    """Distinct rows of `plan` -> fixed-capacity owner buckets.  Returns (send_ids [G*C] i32, send_u [G*C] i32,
    pos_u [M] i32, counts [G] i32) -- all on the device, no host sync; `overflow` (int32 [1], caller-owned, sticky)
    is set to 1 when a bucket needed more than C entries.  expand: also idx_slot [M] int64 = pos_u[u(slot)] as a fifth
    result (tt_route_bucket_expand: route_expand's output from the same launch)."""
    dev, M = plan.unique_rows.device, plan.M
    buf = torch.empty(2 * G * C + M + G, dtype=torch.int32, device=dev)
    send_ids, send_u = buf[:G * C], buf[G * C:2 * G * C]
    pos_u, counts = buf[2 * G * C:2 * G * C + M], buf[2 * G * C + M:]
    lib = L.load()
    ws = L.workspace(dev, lib.tt_route_workspace_bytes(M, G))
    pads = (L.i32 * G)(*[int(p) for p in pad_id])
    if expand:
        idx = torch.empty(M, dtype=torch.int64, device=dev)
        with _timed("tt_route_bucket_expand"):
            L.check(lib.tt_route_bucket_expand(L.ctx(dev), L.ptr(plan.unique_rows), L.ptr(plan.n_unique), M, G, C, pads, pad_u, L.ptr(send_ids),
                                               L.ptr(send_u), L.ptr(pos_u), L.ptr(counts), L.ptr(overflow), L.ptr(ws), ws.numel(),
                                               L.ptr(plan.sorted_src), L.ptr(plan.seg_offsets), L.ptr(idx), L.stream(dev)),
                    "tt_route_bucket_expand")
        return send_ids, send_u, pos_u, counts, idx
    with _timed("tt_route_bucket"):
        L.check(lib.tt_route_bucket(L.ctx(dev), L.ptr(plan.unique_rows), L.ptr(plan.n_unique), M, G, C, pads, pad_u, L.ptr(send_ids),
                                    L.ptr(send_u), L.ptr(pos_u), L.ptr(counts), L.ptr(overflow), L.ptr(ws), ws.numel(), L.stream(dev)),
                "tt_route_bucket")
    return send_ids, send_u, pos_u, counts


def gather_rows(table: torch.Tensor, rows: torch.Tensor, out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """out[i] = 0 if rows[i] < 0 else table[min(rows[i], R - 1)]  (rows int32; out_dtype float32 or bfloat16)."""
    dev, n, E = table.device, rows.numel(), table.shape[1]
    out = torch.empty((n, E), dtype=out_dtype, device=dev)
    with _timed("tt_gather_rows"):
        L.check(L.load().tt_gather_rows(L.ctx(dev), L.ptr(table), table.shape[0], E, L.ptr(rows), n, L.ptr(out), _dt(out),
                                        L.stream(dev)), "tt_gather_rows")
    return out


def route_expand(plan: DedupPlan, pos_u: torch.Tensor) -> torch.Tensor:
    """idx_slot [M] int64 = pos_u[u(slot)]."""
    dev = pos_u.device
    idx = torch.empty(plan.M, dtype=torch.int64, device=dev)
    with _timed("tt_route_expand"):
        L.check(L.load().tt_route_expand(L.ctx(dev), L.ptr(plan.sorted_src), L.ptr(plan.seg_offsets), L.ptr(plan.n_unique), L.ptr(pos_u),
                                         plan.M, L.ptr(idx), L.stream(dev)), "tt_route_expand")
    return idx


def batch_gather(entity, dense_store, cat_store):
    dev, B = entity.device, entity.numel()
    dd = dense_store.shape[1] if dense_store is not None else 0
    K = cat_store.shape[1] if cat_store is not None else 0
    dense = torch.empty((B, dd), dtype=torch.float32, device=dev)
    ids = torch.empty(B * K, dtype=torch.int64, device=dev)
    L.check(L.load().tt_batch_gather(L.ctx(dev), L.ptr(entity), B, L.ptr(dense_store), dd, L.ptr(cat_store), K, L.ptr(dense),
                                     L.ptr(ids), L.stream(dev)), "tt_batch_gather")
    return dense, ids
