"""ctypes binding of libtwotower_hip.so (include/twotower.h).  PyTorch is plumbing here: it owns
device memory and streams; every kernel on the hot path is reached through this C ABI.

There is NO CPU or eager-PyTorch fallback: if the HIP library is missing, or there is no gfx950
device, the first call raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

import torch

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libtwotower_hip.so"

TT_F32, TT_BF16 = 0, 1
TT_MAX_SIDES, TT_MAX_HIDDEN = 4, 8
TT_GRAD_SPARSE, TT_GRAD_DENSE_SET, TT_GRAD_DENSE_ACC = 0, 1, 2
TT_TOWER_UNFUSED_TAIL = 1
TT_TOWER_UNFUSED_FRONT = 2
TT_TOWER_UNFUSED_BACK = 4
TT_GRAD_SHORT_SEGMENTS = 0x100
TT_GRAD_PLANNED = 0x200
TT_GRAD_DEFER_FINISH = 0x400

vp = C.c_void_p
i32, i64, f32, u64, sz = C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_size_t


class EmbedSide(C.Structure):
    _fields_ = [("ids", vp), ("key_row_offset", vp), ("key_vocab", vp), ("out", vp), ("ld_out", i64),
                ("K", i32), ("out_dtype", i32)]


class StoreSide(C.Structure):
    _fields_ = [("entity", vp), ("entity_stride", i64), ("dense_store", vp), ("cat_store", vp), ("dense_out", vp), ("ids_out", vp),
                ("dense_dim", i32), ("n_rows", i32)]


TT_MAX_CVT = 8


class CvtList(C.Structure):
    _fields_ = [("n", i32), ("reserved", i32), ("src", vp * TT_MAX_CVT), ("dst", vp * TT_MAX_CVT), ("count", i64 * TT_MAX_CVT)]


class IngestLookup(C.Structure):
    _fields_ = [("table", vp), ("table_rows", i64), ("E", i32), ("reserved", i32)]


class GradSrc(C.Structure):
    _fields_ = [("d_out", vp), ("ld", i64), ("K", i32), ("dtype", i32)]


class AdamTensor(C.Structure):
    _fields_ = [("p", vp), ("g", vp), ("m", vp), ("v", vp), ("n", i64)]


class ScoreFwdDir(C.Structure):
    _fields_ = [("A_packed", vp), ("B_packed", vp), ("Ra", i64), ("Rb", i64), ("diag_offset", i64),
                ("sumexp", vp), ("diag", vp), ("rank", vp), ("sumscore", vp), ("rank_mode", i32), ("ab_scale", f32), ("inv_sumexp", vp)]


class ScoreBwdDir(C.Structure):
    _fields_ = [("A_packed", vp), ("B_packed", vp), ("Ra", i64), ("Rb", i64), ("diag_offset", i64),
                ("sumexp_a", vp), ("sumexp_b", vp), ("dA", vp), ("ab_scale", f32), ("b_scale", f32), ("inv_a", vp), ("inv_b", vp)]


_H = vp * TT_MAX_HIDDEN


class TowerParams(C.Structure):
    _fields_ = [("din", i32), ("h0", i32), ("kcat_e", i32), ("n_hidden", i32), ("d_out", i32),
                ("hidden", i32 * TT_MAX_HIDDEN),
                ("w_proj", vp), ("b_proj", vp), ("w", _H), ("b", _H), ("bn_w", _H), ("bn_b", _H),
                ("bn_rm", _H), ("bn_rv", _H), ("bn_nbt", _H), ("w_out", vp), ("b_out", vp), ("compute_dtype", i32),
                ("x_dtype", i32), ("dx_dtype", i32), ("flags", i32),
                ("sync_phase", i32), ("sync_ranks", i32), ("rng_row_offset", i64),
                ("w_proj_bf16", vp), ("w_bf16", _H)]


class TowerActs(C.Structure):
    _fields_ = [("dense", vp), ("x", vp), ("pre", _H), ("act", _H), ("mean", _H), ("rstd", _H), ("y", vp), ("emb", vp), ("emb_packed", vp), ("emb_pack_scale", f32),
                ("bn_sync_local", vp), ("bn_sync_all", vp), ("bn_sync_stride", i64)]


class TowerGrads(C.Structure):
    _fields_ = [("w_proj", vp), ("b_proj", vp), ("w", _H), ("b", _H), ("bn_w", _H), ("bn_b", _H),
                ("w_out", vp), ("b_out", vp), ("d_x", vp), ("scratch", _H), ("d_y", vp), ("s_sync_local", vp), ("s_sync_all", vp), ("s_sync_stride", i64)]


# name -> (restype, argtypes); every symbol include/twotower.h declares
SIGNATURES = {
    "tt_abi_version": (C.c_int, []),
    "tt_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "tt_ctx_destroy": (C.c_int, [vp]),
    "tt_last_error_string": (C.c_char_p, []),
    "tt_ctx_num_cus": (C.c_int, [vp]),
    "tt_embed_lookup_set_profile": (C.c_int, [vp, vp, i32]),
    "tt_embed_lookup_fwd": (C.c_int, [vp, vp, i64, i32, C.POINTER(EmbedSide), i32, i64, vp, vp]),
    "tt_embed_lookup_rows_fwd": (C.c_int, [vp, vp, i64, i32, C.POINTER(EmbedSide), i32, i64, vp, vp]),
    "tt_dedup_workspace_bytes": (sz, [i64]),
    "tt_dedup_plan": (C.c_int, [vp, vp, i64, i64, vp, vp, vp, vp, vp, sz, vp]),
    "tt_dedup_keyed_workspace_bytes": (sz, [i64, i32]),
    "tt_dedup_plan_keyed": (C.c_int, [vp, vp, C.POINTER(i32), i32, i64, vp, vp, vp, vp, vp, sz, vp]),
    "tt_dedup_plan_keyed_km": (C.c_int, [vp, vp, C.POINTER(i32), i32, i64, vp, vp, vp, vp, vp, sz, vp]),
    "tt_dedup_plan_keyed_long": (C.c_int, [vp, vp, i32, C.POINTER(i32), i32, i64, i32, vp, vp, vp, vp, vp, sz, vp, sz, vp]),
    "tt_embed_grad_workspace_bytes": (sz, [i64, i32]),
    "tt_embed_grad_bwd": (C.c_int, [vp, C.POINTER(GradSrc), i32, i64, i32, vp, vp, vp, vp, i64, i32, vp, vp, vp, sz, vp]),
    "tt_adam_hparams": (None, [i64, f32, f32, f32, f32, f32, C.POINTER(f32 * 6)]),
    "tt_adam_dense_step": (C.c_int, [vp, vp, vp, vp, vp, i64, i64, f32, f32, f32, f32, f32, vp, vp]),
    "tt_adam_multi_step": (C.c_int, [vp, C.POINTER(AdamTensor), i32, i64, f32, f32, f32, f32, f32, vp, vp]),
    "tt_sparse_adam_step": (C.c_int, [vp, vp, vp, vp, i64, i32, vp, vp, vp, i64, i64, f32, f32, f32, f32, f32, vp, vp]),
    "tt_adam_fused_step": (C.c_int, [vp, C.POINTER(AdamTensor), i32, vp, vp, vp, i64, i32, vp, vp, vp, i64, i64, f32, f32, f32, f32,
                                     f32, vp, vp]),
    "tt_adam_fused_step_finish": (C.c_int, [vp, C.POINTER(AdamTensor), i32, vp, vp, vp, i64, i32, vp, vp, vp, i64, vp, vp, sz, i64, f32,
                                            f32, f32, f32, f32, vp, vp]),
    "tt_embed_grad_finish": (C.c_int, [vp, i32, vp, i64, vp, vp, sz, vp]),
    "tt_ctx_set_option": (C.c_int, [vp, i32, i32]),
    "tt_flush_deferred": (C.c_int, [vp, vp]),
    "tt_ctx_check_device_errors": (C.c_int, [vp, vp]),
    "tt_handover_retarget": (C.c_int, [vp, vp, vp]),
    "tt_handover_captured_node": (C.c_int, [vp, C.POINTER(vp)]),
    "tt_deferred_pending": (C.c_int, [vp]),
    "tt_launch_count": (C.c_uint64, []),
    "tt_flush_deferred_slabs": (C.c_int, [vp, vp]),
    "tt_tower_workspace_bytes": (sz, [C.POINTER(TowerParams), i64]),
    "tt_tower_mlp_fwd": (C.c_int, [vp, C.POINTER(TowerParams), C.POINTER(TowerActs), i64, i32, f32, u64, vp, vp, sz, vp]),
    "tt_tower_mlp_bwd": (C.c_int, [vp, C.POINTER(TowerParams), C.POINTER(TowerActs), vp, C.POINTER(TowerGrads), i64, i32,
                                   f32, u64, vp, vp, sz, vp]),
    "tt_towers_mlp_fwd": (C.c_int, [vp, i32, C.POINTER(C.POINTER(TowerParams)), C.POINTER(C.POINTER(TowerActs)), i64, i32, f32, u64,
                                    vp, C.POINTER(vp), C.POINTER(sz), vp]),
    "tt_towers_mlp_bwd": (C.c_int, [vp, i32, C.POINTER(C.POINTER(TowerParams)), C.POINTER(C.POINTER(TowerActs)), C.POINTER(vp),
                                    C.POINTER(C.POINTER(TowerGrads)), i64, i32, f32, u64, vp, C.POINTER(vp), C.POINTER(sz), vp]),
    "tt_score_dir_fwd": (C.c_int, [vp, vp, vp, i64, i64, i32, f32, f32, i64, vp, vp, vp, vp, vp]),
    "tt_score_loss_finish": (C.c_int, [vp, i64, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "tt_score_dir_bwd": (C.c_int, [vp, vp, vp, i64, i64, i32, f32, f32, i64, vp, vp, vp, f32, vp, vp]),
    "tt_score_pack_bytes": (sz, [i64, i32]),
    "tt_score_unit_scale": (f32, [f32]),
    "tt_score_pack_bf16": (C.c_int, [vp, vp, i64, i32, f32, vp, vp]),
    "tt_score_pack2_bf16": (C.c_int, [vp, vp, i64, vp, vp, i64, vp, i32, f32, f32, vp]),
    "tt_score_fwd_bf16": (C.c_int, [vp, C.POINTER(ScoreFwdDir), i32, i32, f32, f32, vp]),
    "tt_score_bwd_bf16": (C.c_int, [vp, C.POINTER(ScoreBwdDir), i32, i32, f32, f32, vp, f32, vp]),
    "tt_score_pack_fp8_bytes": (sz, [i64, i32]),
    "tt_score_pack2_fp8": (C.c_int, [vp, vp, i64, vp, vp, i64, vp, i32, f32, f32, vp]),
    "tt_score_fwd_sym_fp8": (C.c_int, [vp, vp, vp, i64, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "tt_score_bwd_fp8": (C.c_int, [vp, C.POINTER(ScoreBwdDir), i32, i32, f32, f32, vp, f32, vp]),
    "tt_score_fwd_sym_workspace_bytes": (sz, [i64, i32]),
    "tt_score_fwd_sym_bf16": (C.c_int, [vp, vp, vp, i64, i32, f32, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "tt_score_matrix": (C.c_int, [vp, vp, vp, i64, i64, i32, f32, vp, i64, vp]),
    "tt_score_dense_workspace_bytes": (sz, [i64, i32]),
    "tt_score_dense_fwd": (C.c_int, [vp, vp, vp, i64, i32, f32, i32, f32, vp, vp, vp, vp, vp, vp]),
    "tt_score_dense_bwd": (C.c_int, [vp, vp, vp, i64, i32, f32, i32, f32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "tt_diag_rank_rows": (C.c_int, [vp, vp, i64, i64, i64, i64, vp, vp]),
    "tt_topk_rows": (C.c_int, [vp, vp, i64, i64, i64, i32, vp, vp, vp]),
    "tt_linear_fwd": (C.c_int, [vp, vp, i64, vp, vp, vp, i64, i64, i32, i32, i32, vp]),
    "tt_route_workspace_bytes": (sz, [i64, i32]),
    "tt_route_bucket": (C.c_int, [vp, vp, vp, i64, i32, i32, C.POINTER(i32), i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "tt_route_expand": (C.c_int, [vp, vp, vp, vp, vp, i64, vp, vp]),
    "tt_route_bucket_expand": (C.c_int, [vp, vp, vp, i64, i32, i32, C.POINTER(i32), i32, vp, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]),
    "tt_dedup_plan_runs": (C.c_int, [vp, vp, i32, i64, i64, vp, vp, vp, vp, vp, sz, vp]),
    "tt_gather_rows": (C.c_int, [vp, vp, i64, i32, vp, i64, vp, i32, vp]),
    "tt_copy_multi": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), vp]),
    "tt_batch_ingest": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), C.POINTER(EmbedSide), i32, i64, vp, vp, i64, C.POINTER(CvtList), vp]),
    "tt_batch_gather": (C.c_int, [vp, vp, i64, vp, i32, vp, i32, vp, vp, vp]),
    "tt_batch_ingest_store": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), C.POINTER(EmbedSide), C.POINTER(StoreSide), i32,
                                        i64, vp, vp, vp, i64, C.POINTER(CvtList), vp]),
    "tt_batch_ingest_lookup": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), C.POINTER(EmbedSide), i32, i64, vp,
                                         C.POINTER(IngestLookup), C.POINTER(CvtList), vp]),
    "tt_batch_ingest_store_lookup": (C.c_int, [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), C.POINTER(EmbedSide), C.POINTER(StoreSide),
                                               i32, i64, vp, vp, C.POINTER(IngestLookup), C.POINTER(CvtList), vp]),
}

_lib: Optional[C.CDLL] = None
_ctxs: dict = {}
_defer_on: set = set()       # devices whose context queues the towers' slab reduction (set_defer_slab_reduce)
_riders_on: dict = {}        # device -> mask of what its context queues: 1 plan compaction, 2 loss reduction (set_defer_riders)
_workspaces: dict = {}


class TwoTowerHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the HIP library and declare every entry point; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise TwoTowerHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the two-tower hot path.")
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().tt_last_error_string().decode("utf-8", "replace")
        raise TwoTowerHipError(f"{what or 'twotower call'} failed (status {rc}): {msg}")


def ctx(device: torch.device) -> vp:
    device = torch.device(device)
    if device.type != "cuda":
        raise TwoTowerHipError(f"the two-tower hot path runs on MI355X only; got tensors on '{device}' "
                               "(no CPU fallback exists)")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ctxs:
        h = vp()
        check(load().tt_ctx_create(idx, C.byref(h)), "tt_ctx_create")
        _ctxs[idx] = h
    return _ctxs[idx]


def stream(device: torch.device) -> vp:
    return vp(torch.cuda.current_stream(device).cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> vp:
    return vp(0 if t is None else t.data_ptr())


TT_OPT_DEFER_SLAB_REDUCE = 1
TT_OPT_KEYED_PARTS = 2
TT_OPT_SCORE_BWD_ROWS_MIN = 3
TT_OPT_DEFER_RIDERS = 4
TT_OPT_FP8_GRAD = 5
TT_OPT_CHAINED = 6
TT_OPT_LOOKUP_NT = 7
TT_OPT_CHAIN_SPIN = 8


def set_option(device: torch.device, option: int, value: int):
    """tt_ctx_set_option on the device's context (include/twotower.h: TT_OPT_*)."""
    check(load().tt_ctx_set_option(ctx(device), int(option), int(value)), "tt_ctx_set_option")


def set_defer_slab_reduce(device: torch.device, on: bool):
    """tt_towers_mlp_bwd leaves its slab reduction queued in the context (include/twotower.h: TT_OPT_DEFER_SLAB_REDUCE)."""
    check(load().tt_ctx_set_option(ctx(device), TT_OPT_DEFER_SLAB_REDUCE, 1 if on else 0), "tt_ctx_set_option")
    idx = torch.device(device).index
    (_defer_on.add if on else _defer_on.discard)(idx if idx is not None else torch.cuda.current_device())
    if not on:
        flush_deferred(device)


def set_defer_riders(device: torch.device, on, loss_only: bool = False):
    """tt_dedup_plan_keyed* / tt_score_fwd_sym_* leave the plan's compaction and the loss reduction queued in the context; the towers'
    fused tail launches run them (include/twotower.h: TT_OPT_DEFER_RIDERS).  loss_only: the plan is compacted at once (somebody
    reads it before the towers run).  Switching it off launches what is still queued."""
    mask = (2 if loss_only else 3) if on else 0
    check(load().tt_ctx_set_option(ctx(device), TT_OPT_DEFER_RIDERS, mask), "tt_ctx_set_option")
    idx = torch.device(device).index
    idx = idx if idx is not None else torch.cuda.current_device()
    if mask:
        _riders_on[idx] = mask
    else:
        _riders_on.pop(idx, None)
        flush_deferred(device)


def riders_deferred(device: torch.device, which: int = 3) -> bool:
    """which: 1 the plan compaction, 2 the loss reduction"""
    idx = torch.device(device).index
    return bool(_riders_on.get(idx if idx is not None else torch.cuda.current_device(), 0) & which)


def check_device_errors(device: torch.device):
    """tt_ctx_check_device_errors: raises TwoTowerHipError if a kernel has raised the context's sticky device error word since the
    last check (synchronises the current stream; the word and the chain buffers are reset).  No-op on a device without a context."""
    device = torch.device(device)
    idx = device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else None)
    if device.type != "cuda" or idx not in _ctxs:
        return
    check(load().tt_ctx_check_device_errors(_ctxs[idx], stream(device)), "tt_ctx_check_device_errors")


def handover_captured_node(device: torch.device) -> Optional[int]:
    """tt_handover_captured_node: the graph node the last tt_batch_ingest* call became (its stream was being captured), or None."""
    h = vp()
    check(load().tt_handover_captured_node(ctx(device), C.byref(h)), "tt_handover_captured_node")
    return h.value


class handover_retarget:
    """with handover_retarget(device, graph_exec, node): every tt_batch_ingest* call inside re-points that node of the executable
    graph at its own kernel and arguments instead of launching (tt_handover_retarget)."""

    def __init__(self, device: torch.device, graph_exec: int, node: int):
        self.device, self.exec, self.node = device, graph_exec, node

    def __enter__(self):
        check(load().tt_handover_retarget(ctx(self.device), vp(self.exec), vp(self.node)), "tt_handover_retarget")
        return self

    def __exit__(self, *exc):
        check(load().tt_handover_retarget(ctx(self.device), vp(0), vp(0)), "tt_handover_retarget")
        return False


def flush_deferred(device: torch.device):
    """Launch a queued slab reduction now (no-op when nothing is queued)."""
    lib = load()
    c = ctx(device)
    if lib.tt_deferred_pending(c):
        check(lib.tt_flush_deferred(c, stream(device)), "tt_flush_deferred")


def workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Stream-ordered scratch: one growable buffer per (device, stream).  The slabs of a queued (deferred) reduction live in
    it: they are reduced before the buffer is handed to anybody else."""
    if _defer_on:                     # (only the slabs live in this buffer: queued riders keep their own)
        lib, c = load(), ctx(device)
        if lib.tt_deferred_pending(c) & 1:
            check(lib.tt_flush_deferred_slabs(c, stream(device)), "tt_flush_deferred_slabs")
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def num_cus(device: torch.device) -> int:
    return int(load().tt_ctx_num_cus(ctx(device)))


def no_dynamo(fn):
    """Keep torch.compile's tracer out of `fn`: the arithmetic behind it is ctypes calls into the HIP library, which the
    tracer cannot see through (it hands our stream / pointer plumbing proxy objects).  torch.compile(task) -- the reference
    driver's optional mode, scripts/train.py:223-225 -- then runs the step as it is; GraphedTrainStep is the counterpart."""
    try:
        import torch._dynamo as _d
        return _d.disable(fn)
    except Exception:                                      # pragma: no cover - torch builds without dynamo
        return fn
