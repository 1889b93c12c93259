"""GraphedTrainStep: the whole training step (forward + backward + FusedAdam) captured ONCE into a HIP
graph and replayed per batch -- the MI355X counterpart of the reference's optional
torch.compile(train_task, mode="reduce-overhead") (scripts/train.py:223-225, off by default there).

Why: at batch 8192 the step is ~75 short kernels (0.9 ms of GPU time); launched one by one from Python the
host needs ~1.5 ms, so the GPU idles 40 % of the time.  A replay costs one launch.

What makes the capture replayable with NEW data every step:
  * the batch is copied into static input buffers before the replay -- one launch (`ops.batch_ingest`) that also leaves the
    fused table rows of the batch's ids in key-major order for the duplicate-row plan (the plan's own strided load of one
    key's rows is 6.5 of a sort workgroup's 18 us);
  * Adam's per-step scalars (lr from the scheduler, the bias corrections) and the dropout seed live in
    device memory (`hparams_dev` / `seed_dev` arguments of the C ABI) and are refreshed by one small
    pinned-memory copy before each replay;
  * every buffer the step allocates comes from the graph's private pool, so addresses are stable.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib as L
from . import ops
from .config import settings
from .kjt import KeyedJaggedTensor
from .optim import FusedAdam


class GraphedTrainStep:
    _steps_per_replay = 1          # (unrolled.UnrolledTrainStep: several steps, hand-overs included, per graph launch)

    def __init__(self, task, optimizer: FusedAdam, example_batch: Dict, return_metrics: bool = True, warmup: int = 3,
                 defer_long: bool = True, defer_slabs: bool = True, defer_riders: bool = True, preserve_state: bool = True,
                 accumulate_metrics: bool = False, metric_sums: Optional[torch.Tensor] = None):
        """defer_long / defer_slabs: the long rows' finish rides in the optimiser's launch and the towers' slab reduction in the
        embedding gradient's (both bit-identical to the separate launches; the arguments exist for that comparison).
        preserve_state: the eager warm-up steps (allocator + first-call paths) are REAL steps on the example batch; with this flag
        everything they change -- weights, Adam moments and step counts, BatchNorm running statistics -- is put back before the
        capture, so a captured loop starts from the state an eager loop starts from (row-sparse tables: only the example batch's
        rows move and only those are saved; dense-gradient tables are saved whole).
        accumulate_metrics: `metric_sums` (device, 8 floats: loss, accuracy, positive mean, negative mean, gap, ...: twotower.h
        tt_score_loss_finish) += the step's figures inside the replay -- an epoch's mean loss / accuracy (the reference driver's
        avg_train_loss: scripts/train.py:357-358) without a host sync per step.  metric_sums: add into THIS tensor (another captured
        step's sums: the two then share one epoch total; what it holds survives this object's warm-up)."""
        if not isinstance(optimizer, FusedAdam):
            raise TypeError("GraphedTrainStep needs jodalrob_twotower_amd.optim.FusedAdam (device-side hyper-parameters)")
        self.task, self.opt, self.return_metrics = task, optimizer, return_metrics
        self._defer_long, self._defer_slabs, self._defer_riders = bool(defer_long), bool(defer_slabs), bool(defer_riders)
        dev = example_batch["notice"]["dense"].device
        self.static = {side: {"dense": example_batch[side]["dense"].clone(),
                              "kjt": KeyedJaggedTensor(example_batch[side]["kjt"].keys(), example_batch[side]["kjt"].values().clone())}
                       for side in ("notice", "company")}
        ng = len(optimizer.param_groups)
        # per-step scalars (Adam step sizes, dropout seed) travel through a RING of pinned host slots: the copy
        # kernel reads the slot when it executes, and the host may be many steps ahead of the GPU by then
        self._n_scalar = ng * 8 + 2
        self._slot_len = (self._n_scalar + 3) // 4 * 4                 # 16-byte slots
        self._ring = 64
        self._host_ring = torch.zeros(self._ring, self._slot_len, dtype=torch.float32).pin_memory()
        self._slot_events = [None] * self._ring
        self._slot = 0
        self._unmarked = []                                            # ring slots filled since the last _mark_slot
        self._host = self._host_ring[0, :self._n_scalar]
        self._dev = torch.zeros(ng * 8 + 2, dtype=torch.float32, device=dev)
        self._hp_dev = self._dev[:ng * 8].view(ng, 8)
        self._seed_dev = self._dev[ng * 8:].view(torch.int64)          # 2 floats = one 64-bit word
        self._ones = torch.ones((), dtype=torch.float32, device=dev)
        self._towers = [m for m in task.modules() if hasattr(m, "_seed_dev") and hasattr(m, "dense_parameters")]
        self.metric_sums = torch.zeros(8, dtype=torch.float32, device=dev) if (accumulate_metrics and return_metrics) else None
        sums_before = None
        if metric_sums is not None and self.metric_sums is not None:
            self.metric_sums, sums_before = metric_sums, metric_sums.clone()
        self._ingest = self._setup_ingest()
        self._shadows = self._setup_shadows()
        if self._ingest is not None:
            self._run_ingest([], None)                           # rows_km of the example batch: warm-up and capture see it
        # eager warm-up on a side stream (allocator + first-call paths), then capture
        snap = self._snapshot_state() if (preserve_state and warmup > 0) else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_once()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if snap is not None:
            self._restore_state(snap)
        if self.metric_sums is not None:
            self.metric_sums.zero_() if sums_before is None else self.metric_sums.copy_(sums_before)
        if self._ingest is not None:
            self._run_ingest([], None)
        self._push_scalars()
        for t in self._towers:
            t._seed_dev = self._seed_dev
        optimizer._hp_dev = self._hp_dev
        self.graph = None
        # With a process group alive, its watchdog thread polls events while we capture; under the default "global" capture
        # mode that poll is an error ("operation not permitted when stream is capturing") that aborts the process --
        # now and then, depending on timing.  "thread_local" restricts the check to the capturing thread.
        import torch.distributed as _dist
        mode = "thread_local" if (_dist.is_available() and _dist.is_initialized()) else "global"
        n0 = L.load().tt_launch_count()
        try:
            self._capture(mode)
        finally:
            self.library_launches = int(L.load().tt_launch_count() - n0) + 1       # (+ the hand-over launch in front of every replay)
            optimizer._hp_dev = None
            for t in self._towers:
                t._seed_dev = None
        # the capture itself ran the host-side bookkeeping of one optimiser step (per captured body) without executing it
        optimizer.advance_steps(-self._steps_per_replay)

    def _capture(self, mode: str):
        """Captures self._body() into self.graph (ONE graph; segmented.SegmentedTrainStep cuts it at the collectives instead)."""
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.result = self._body()

    def _replay_graph(self):
        self.graph.replay()

    def _setup_ingest(self):
        """Key-major row hand-over (ops.batch_ingest) when the step looks the static ids up in ONE local fused table with the
        per-key plan: returns (store, embedders, rows_km, static id tensors, B) or None (then the batch is handed over by plain
        copies and the plan gathers its rows out of the lookup's slot-major array)."""
        self._x_static, self._rows_sm = None, None
        if not settings.graph_ingest:
            return None
        model = getattr(self.task, "two_tower_model", None)
        towers = [getattr(model, n, None) for n in ("notice_tower", "company_tower")]
        if model is None or any(t is None for t in towers):
            return None
        embs = [t.categorical_embedder for t in towers]
        ex = getattr(self.task, "exchange", None)
        if ex is not None:
            # row-wise sharded tables: the key-major rows are GLOBAL fused rows (the embedders' offsets span the global row space);
            # the fixed-capacity exchange sorts them in place of its rows-only lookup
            store = getattr(ex, "store", None)
            ids = [self.static[side]["kjt"].values() for side in ("notice", "company")]
            B = self.static["notice"]["dense"].shape[0]
            if store is None or not hasattr(ex, "poll_overflow") or not (0 < B <= ops.KEYED_MAX_B) or \
                    any(len(e.keys) == 0 or len(e.keys) > 64 or v.dtype != torch.int64 or not v.is_contiguous() for e, v in zip(embs, ids)):
                return None
            rows_km = torch.empty(sum(v.numel() for v in ids), dtype=torch.int32, device=ids[0].device)
            return store, embs, rows_km, ids, B
        store = embs[0].store
        ids = [self.static[side]["kjt"].values() for side in ("notice", "company")]
        B = self.static["notice"]["dense"].shape[0]
        if any(e.store is not store for e in embs) or store.weight is None or not (0 < B <= ops.KEYED_MAX_B):
            return None
        if any(len(e.keys) == 0 or len(e.keys) > 64 or v.dtype != torch.int64 or not v.is_contiguous() or v.numel() != B * len(e.keys)
               for e, v in zip(embs, ids)):
            return None
        rows_km = torch.empty(sum(v.numel() for v in ids), dtype=torch.int32, device=ids[0].device)
        # persistent tower inputs x = [projection | embedding rows]: the hand-over launch looks the batch's rows up straight into
        # their embedding columns (tt_batch_ingest_lookup), the captured forward finds them filled and has no lookup launch
        if settings.graph_ingest_lookup:
            xs = [torch.zeros((B, t.x_width), dtype=t.x_dtype, device=ids[0].device) for t in towers]
            outs = [ops.LookupSide(v, e._key_row_offset, e._key_vocab, x[:, t.tower_hidden_dims[0]:], len(e.keys))
                    for e, v, x, t in zip(embs, ids, xs, towers)]
            if ops.ingest_lookup_supported(store.weight, outs):
                self._x_static = xs
        # the same rows in slot order: the captured lookup reads them instead of decoding the ids again (tt_embed_lookup_rows_fwd)
        if self._x_static is None and settings.graph_ingest_rows and store.weight.shape[1] % 4 == 0:
            self._rows_sm = torch.empty_like(rows_km)
        return store, embs, rows_km, ids, B

    def _setup_shadows(self):
        """bf16 shadows of the towers' projection and block weights (bf16 towers): [(tower, w_proj16, [w16 per block])].  The hand-over
        launch converts the f32 weights into them at EVERY step (ops.batch_ingest(cvt=...)), so whoever updated the weights since
        the last hand-over -- the replayed Adam, an eager step, load_state_dict -- the replay reads current values; the towers see the
        shadows only while this object's step runs (_body)."""
        if not settings.graph_weight_shadows or self._ingest is None:
            return []
        out = []
        for t in self._towers:
            if getattr(t, "mlp_dtype", None) != "bf16" or t.n_hidden < 1:
                continue
            lins = [t.mlp[4 * i] for i in range(t.n_hidden)]
            ws = [t.dense_projection.weight] + [l.weight for l in lins]
            if any(w.dtype != torch.float32 or not w.is_contiguous() or w.numel() % 8 for w in ws):
                continue
            sh = [torch.empty(w.shape, dtype=torch.bfloat16, device=w.device) for w in ws]
            out.append((t, ws, sh))
        n = sum(len(ws) for _, ws, _ in out)
        return out if 0 < n <= L.TT_MAX_CVT else []

    def _cvt(self):
        return [(s_, w) for _, ws, sh in self._shadows for w, s_ in zip(ws, sh)]

    def _lookup_outs(self):
        towers = [self.task.two_tower_model.notice_tower, self.task.two_tower_model.company_tower]
        return [x[:, t.tower_hidden_dims[0]:] for x, t in zip(self._x_static, towers)]

    def _register(self, store, ids, rows_km):
        xs = getattr(self, "_x_static", None)
        store.ingest = (ids, [v._version for v in ids], rows_km, xs, getattr(self, "_rows_sm", None))
        store.ingest_x_fresh = xs is not None

    @staticmethod
    def _table_rows(store) -> int:
        """Size of the row space the hand-over's fused rows index: the fused table, or the GLOBAL row space of row-wise sharded tables."""
        g = getattr(store, "global_rows", None)
        return int(g) if g is not None else int(store.weight.shape[0])

    def _run_ingest(self, pairs, src_ids):
        """One launch: the copy segments + the key-major rows of `src_ids` (default: the static id buffers themselves) (+ the
        lookup of those rows into the persistent tower inputs)."""
        store, embs, rows_km, ids, B = self._ingest
        src = ids if src_ids is None else src_ids
        xs = getattr(self, "_x_static", None)
        if xs is not None:
            sides = [ops.LookupSide(v, e._key_row_offset, e._key_vocab, o, len(e.keys)) for e, v, o in zip(embs, src, self._lookup_outs())]
            ops.batch_ingest(pairs, sides, B, rows_km, table=store.weight, cvt=self._cvt())
        else:
            sides = [ops.LookupSide(v, e._key_row_offset, e._key_vocab, None, len(e.keys)) for e, v in zip(embs, src)]
            ops.batch_ingest(pairs, sides, B, rows_km, rows_sm=getattr(self, "_rows_sm", None), cvt=self._cvt(), table_rows=self._table_rows(store))
        self._register(store, ids, rows_km)

    def _body(self):
        self.opt.zero_grad(set_to_none=True)
        # backward and optimiser step are one unit here: the gradient reduction leaves its long rows to the optimiser's launch
        # (FusedAdam finishes them whichever of its paths it takes) -- one launch fewer in the dependent chain
        defer = self._defer_long
        stores = list(getattr(self.opt, "_stores", ()))
        for st in stores:
            st.defer_long_finish = defer
        # likewise the slab reduction of the towers' weight gradients rides in the embedding gradient's launch (the two do not
        # depend on each other); whatever is still queued after the backward is launched on its own
        dev = self._dev.device
        # (a sharded task's dense-gradient all-reduce comes BEHIND the exchange's backward, whose first launch hosts the queue, and
        # flushes whatever is still queued before it reads the gradients: towers.py)
        ex = getattr(self.task, "exchange", None)
        slabs = self._defer_slabs
        # the keyed plan's compaction and the score forward's loss reduction ride in the towers' tail launches (two launches fewer
        # in the chain).  A sharded task's exchange reads the plan at once: only the loss reduction rides there (and not under SyncBN,
        # whose tail kernels run in two phases with a collective between them)
        riders = self._defer_riders and self._ingest is not None and (ex is None or not getattr(self.task, "sync_bn", False))
        for t, _, sh in self._shadows:                 # the hand-over launch in front of this step has refreshed them
            t._w16 = (sh[0], sh[1:])
        try:
            if riders:
                L.set_defer_riders(dev, True, loss_only=ex is not None)
            res = self.task(self.static, return_metrics=self.return_metrics)
            loss = res["loss"] if isinstance(res, dict) else res
            if slabs:
                L.set_defer_slab_reduce(dev, True)
            try:
                loss.backward(self._ones)              # preallocated seed gradient: no fill kernel per step
            finally:
                if slabs:
                    L.set_defer_slab_reduce(dev, False)      # (flushes)
            self.opt.step()
            for st in stores:                          # a store this optimiser did not step: nobody else will finish its gradient
                if st.sparse_grad is not None:
                    ops.embed_grad_finish(st.sparse_grad[0])
            if self.metric_sums is not None:           # (behind the backward: a riding loss reduction has written out8 by now)
                self.metric_sums.add_(res.out8)
        finally:
            for t, _, _ in self._shadows:
                t._w16 = None
            if riders:
                L.set_defer_riders(dev, False)         # (launches what nobody hosted: e.g. the loss reduction of a forward-only pass)
            for st in stores:
                st.defer_long_finish = False
        return res

    def _eager_once(self):
        if self._ingest is not None:
            self._run_ingest([], None)          # (an optimiser step has changed the rows since the last hand-over filled x)
        self._body()

    # ---- warm-up without side effects ---------------------------------------------------------------------------------
    def _touched_rows(self, store):
        """Fused rows (sorted, distinct) the example batch looks up in `store`, or None when they cannot be told from here
        (row-wise sharded tables: the rows this rank updates come from every rank's batch)."""
        model = getattr(self.task, "two_tower_model", None)
        if model is None or getattr(self.task, "exchange", None) is not None:
            return None
        rows = []
        for name, side in (("notice_tower", "notice"), ("company_tower", "company")):
            tw = getattr(model, name, None)
            if tw is None:
                return None
            e = tw.categorical_embedder
            if e.store is not store or not e.keys:
                continue
            ids = self.static[side]["kjt"].values().view(-1, len(e.keys))
            rows.append((ids.clamp(min=0).minimum(e._key_vocab - 1) + e._key_row_offset).reshape(-1))      # cat_embed.py:114-117
        return torch.unique(torch.cat(rows)) if rows else None

    def _snapshot_state(self):
        opt = self.opt
        opt._flush_pending()
        table_ids = opt._table_param_ids()
        snap = {"buffers": [(b, b.detach().clone()) for b in self.task.buffers()], "dense": [], "stores": [], "steps": []}
        for group in opt.param_groups:
            for p in group["params"]:
                if id(p) in table_ids:
                    continue
                st = opt.state.get(p) or {}
                snap["dense"].append((p, p.detach().clone(), None if "exp_avg" not in st else
                                      (st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"].clone())))
        for store in getattr(opt, "_stores", ()):
            if store.weight is None or not store.optim_parameters():
                continue
            st = opt._state_of(store)
            rows = self._touched_rows(store) if store.grad_mode == "sparse" else None
            if rows is None and 3 * store.weight.numel() * 4 > (8 << 30):
                rows = torch.empty(0, dtype=torch.int64, device=store.weight.device)     # too large to copy whole: left as the warm-up leaves it
            pick = (lambda t: t[rows].clone()) if rows is not None else (lambda t: t.clone())
            snap["stores"].append((store, rows, pick(store.weight), pick(st["m"]), pick(st["v"]), st["step"]))
        snap["steps"] = [(st, st["step"].clone()) for st in opt.state.values() if "step" in st]
        return snap

    @torch.no_grad()
    def _restore_state(self, snap):
        opt = self.opt
        opt._flush_pending()
        for b, saved in snap["buffers"]:
            b.copy_(saved)
        for p, w, st0 in snap["dense"]:
            p.copy_(w)
            st = opt.state.get(p)
            if st and "exp_avg" in st:
                if st0 is None:                        # the state came into being during the warm-up: back to Adam's initial state
                    st["exp_avg"].zero_(); st["exp_avg_sq"].zero_(); st["step"] = torch.tensor(0.0)
                else:
                    st["exp_avg"].copy_(st0[0]); st["exp_avg_sq"].copy_(st0[1]); st["step"] = st0[2].clone()
        for store, rows, w, m, v, step in snap["stores"]:
            st = opt._state_of(store)
            if rows is None:
                store.weight.copy_(w); st["m"].copy_(m); st["v"].copy_(v)
            elif rows.numel():
                store.weight[rows] = w; st["m"][rows] = m; st["v"][rows] = v
            st["step"] = step
        known = {id(st): val for st, val in snap["steps"]}
        for st in opt.state.values():
            if "step" in st:
                st["step"] = known[id(st)].clone() if id(st) in known else torch.tensor(0.0)
        opt._step_cache = None
        opt.zero_grad(set_to_none=True)

    def _fill_slot(self, ahead: int = 0):
        """Writes the scalars of the step `ahead` steps after the next one into the next ring slot; returns the (dst, src) copy pair."""
        self._slot = (self._slot + 1) % self._ring
        self._unmarked.append(self._slot)
        ev = self._slot_events[self._slot]
        if ev is not None:
            ev.synchronize()                                        # the copy that last read this slot has run
        host = self._host_ring[self._slot, :self._n_scalar]
        ng = len(self.opt.param_groups)
        # the optimiser's own count: eager steps taken between replays (a ragged last batch, another captured step sharing
        # the optimiser) advance the bias corrections exactly as they do in an all-eager loop
        step = self.opt.peek_step() + 1 + ahead
        for gi, g in enumerate(self.opt.param_groups):
            hp = ops.adam_hparams(step, float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"])
            host[gi * 8: gi * 8 + 6] = torch.tensor(hp)
        host[ng * 8:].view(torch.int64).random_()
        return (self._dev, host)

    def _mark_slot(self):
        """One event behind the launch(es) that read the slots filled since the last mark: a slot is refilled only after it."""
        slots, self._unmarked = self._unmarked, []
        if not slots:
            return
        ev = self._slot_events[slots[0]]
        if ev is None or len(slots) > 1:
            ev = torch.cuda.Event()
        ev.record()
        for sl in slots:
            self._slot_events[sl] = ev

    def _push_scalars(self):
        ops.copy_multi([self._fill_slot()])
        self._mark_slot()

    def _handover_pairs(self, batch: Optional[Dict]):
        """(copy segments, per-side id sources) that hand `batch` over into the static buffers (None: nothing to copy)."""
        pairs, src_ids = [], []                                 # src_ids per side: where this step's ids are read from
        if batch is None:
            return pairs, None
        for side in ("notice", "company"):
            d, v = batch[side]["dense"], batch[side]["kjt"].values()
            sd, sv = self.static[side]["dense"], self.static[side]["kjt"].values()
            if d.device == sd.device and d.dtype == sd.dtype and v.dtype == sv.dtype and d.is_contiguous() and v.is_contiguous():
                pairs += [(sd, d), (sv, v)]
                src_ids.append(v)
            else:                                               # host batch / other dtype: ordinary copies
                sd.copy_(d, non_blocking=True)
                sv.copy_(v, non_blocking=True)
                src_ids.append(sv)
        return pairs, src_ids

    def step(self, batch: Optional[Dict] = None):
        """Train on `batch` (or on whatever the static buffers hold); returns the static result."""
        pairs, src_ids = self._handover_pairs(batch)
        if self._ingest is not None:
            # ONE launch: the four batch buffers + the scalars + the key-major rows of this step's ids -- also when nobody handed a
            # batch over: the static id buffers may have been written to by means no version counter sees (`.data`), and the
            # replayed plan sorts whatever rows_km holds
            self._run_ingest(pairs + [self._fill_slot()], src_ids)
            self._mark_slot()
        else:
            ops.copy_multi(pairs + [self._fill_slot()])         # ONE launch: the four batch buffers + the scalars
            self._mark_slot()
        return self._replay()

    def _replay(self):
        self._replay_graph()
        self.opt.advance_steps(self._steps_per_replay)
        ex = getattr(self.task, "exchange", None)
        if ex is not None and hasattr(ex, "poll_overflow"):
            ex.poll_overflow()                                  # sharded tables: a bucket overflow rejects the step (non-blocking)
        return self.result

    def step_from_store(self, notice_store, company_store, pairs: torch.Tensor, order: Optional[torch.Tensor] = None, offset: int = 0):
        """Train on the B pairs `pairs[order[offset : offset + B]]` (order None: `pairs[offset : offset + B]`) gathered straight
        out of the device-resident feature stores into the static buffers -- ONE launch (tt_batch_ingest_store: dense rows, ids,
        key-major fused rows, the step scalars) and the replay; nothing per step crosses PCIe and no batch tensors are built
        (the reference assembles the batch on the host and copies it over: unified_bid_data_loader.py:461-504, :630-684;
        scripts/train.py:261-273).  `pairs`: int64 [P, 2] on the device, (notice row, company row) per pair; the stores are
        data_loader.DeviceFeatureStore objects (`.dense` f32 [N, D], `.categorical` int64 [N, K])."""
        self._ingest_from_store(notice_store, company_store, pairs, order, offset)
        self._mark_slot()
        return self._replay()

    def _ingest_from_store(self, notice_store, company_store, pairs: torch.Tensor, order: Optional[torch.Tensor], offset: int, ahead: int = 0):
        """The hand-over launch of step_from_store (tt_batch_ingest_store) into the static buffers; `ahead`: _fill_slot."""
        B = self.static["notice"]["dense"].shape[0]
        if pairs.dtype != torch.int64 or pairs.dim() != 2 or pairs.shape[1] != 2 or not pairs.is_contiguous():
            raise ValueError("step_from_store: pairs must be a contiguous int64 [P, 2] tensor")
        if order is None and (offset < 0 or offset + B > pairs.shape[0]):
            raise ValueError("step_from_store: the batch runs past the pair list")
        flat = pairs.view(-1)
        base = 0 if order is not None else 2 * offset
        model = self.task.two_tower_model
        embs = [model.notice_tower.categorical_embedder, model.company_tower.categorical_embedder]
        sides, stores = [], []
        xs = getattr(self, "_x_static", None) if self._ingest is not None else None
        outs = self._lookup_outs() if xs is not None else [None, None]
        for i, (side, fs, e) in enumerate(zip(("notice", "company"), (notice_store, company_store), embs)):
            sd, sv = self.static[side]["dense"], self.static[side]["kjt"].values()
            sides.append(ops.LookupSide(None, e._key_row_offset, e._key_vocab, outs[i], len(e.keys)))
            stores.append(ops.StoreSide(flat[base + i:], 2, fs.dense, fs.categorical, sd, sv))
        rows_km = self._ingest[2] if self._ingest is not None else None
        ops.batch_ingest_store([self._fill_slot(ahead)], sides, stores, B, order, rows_km, offset if order is not None else 0,
                               table=self._ingest[0].weight if xs is not None else None,
                               rows_sm=getattr(self, "_rows_sm", None) if (self._ingest is not None and xs is None) else None,
                               cvt=self._cvt(), table_rows=self._table_rows(self._ingest[0]) if self._ingest is not None else 0)
        if self._ingest is not None:
            store, _, _, ids, _ = self._ingest
            self._register(store, ids, rows_km)

    def close(self):
        """Final (synchronous) overflow check, then drop the captured graph and its private pool.  With a process group
        alive this must happen BEFORE `destroy_process_group()`: the graph's nodes reference the communicator's buffers
        and streams, and tearing the communicator down first aborts inside RCCL."""
        dev = self._dev.device
        torch.cuda.synchronize(dev)
        ex = getattr(self.task, "exchange", None)
        try:
            L.check_device_errors(dev)                          # (a chained plan launch that gave up: the steps since the last check are invalid)
            if ex is not None and hasattr(ex, "check_overflow"):
                ex.check_overflow()
        finally:
            self.result = None
            if self._ingest is not None:
                self._ingest[0].ingest = None                   # the key-major rows belonged to this object's static buffers
                self._ingest = None
            if self.graph is not None:
                self.graph.reset()
                self.graph = None
            import gc
            gc.collect()
            torch.cuda.synchronize(dev)


class GraphedEvalStep:
    """The evaluation pass of one batch -- towers in eval mode, loss and metrics, the rank of every positive in its row, and from
    the ranks Recall@5 / Recall@10 / MRR (src/evaluation/evaluator.py:20-71, :123-155) -- captured once and replayed per batch,
    fed like GraphedTrainStep.step_from_store from the device-resident feature stores.  Per batch the eager evaluator costs ~30
    launches from Python and a dozen `.item()` round trips; here it is one hand-over launch, one replay and NO host sync: the
    eight per-batch figures are added into a device accumulator, read once at the end (`means()`).
    metrics order: loss, accuracy, similarity_gap, positive_similarity_mean, negative_similarity_mean, recall@5, recall@10, mrr."""

    KEYS = ("loss", "accuracy", "similarity_gap", "positive_similarity_mean", "negative_similarity_mean", "recall@5", "recall@10", "mrr")

    def __init__(self, task, example_batch: Dict, warmup: int = 2):
        self.task = task
        dev = example_batch["notice"]["dense"].device
        self.static = {side: {"dense": example_batch[side]["dense"].clone(),
                              "kjt": KeyedJaggedTensor(example_batch[side]["kjt"].keys(), example_batch[side]["kjt"].values().clone())}
                       for side in ("notice", "company")}
        self.B = self.static["notice"]["dense"].shape[0]
        self.acc = torch.zeros(8, dtype=torch.float32, device=dev)
        self.batches = 0
        was_training = task.training
        task.eval()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.acc.zero_()
        self.graph = torch.cuda.CUDAGraph()
        import torch.distributed as _dist
        mode = "thread_local" if (_dist.is_available() and _dist.is_initialized()) else "global"
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode=mode):
            self._body()
        task.train(was_training)

    def _body(self):
        res, ranks = self.task.forward_with_ranks(self.static)
        k5, k10 = min(5, self.B), min(10, self.B)
        r = ranks.float()
        per = torch.stack([res["loss"].reshape(()), res["accuracy"].reshape(()), res["similarity_gap"].reshape(()),
                           res["positive_similarity_mean"].reshape(()), res["negative_similarity_mean"].reshape(()),
                           (ranks < k5).float().mean(), (ranks < k10).float().mean(), (1.0 / (r + 1.0)).mean()])
        self.acc.add_(per)

    def reset(self):
        self.acc.zero_()
        self.batches = 0

    def step_from_store(self, notice_store, company_store, pairs: torch.Tensor, order: Optional[torch.Tensor] = None, offset: int = 0):
        flat = pairs.view(-1)
        base = 0 if order is not None else 2 * offset
        model = self.task.two_tower_model
        embs = [model.notice_tower.categorical_embedder, model.company_tower.categorical_embedder]
        sides, stores = [], []
        for i, (side, fs, e) in enumerate(zip(("notice", "company"), (notice_store, company_store), embs)):
            sides.append(ops.LookupSide(None, e._key_row_offset, e._key_vocab, None, len(e.keys)))
            stores.append(ops.StoreSide(flat[base + i:], 2, fs.dense, fs.categorical, self.static[side]["dense"], self.static[side]["kjt"].values()))
        ops.batch_ingest_store([], sides, stores, self.B, order, None, offset if order is not None else 0)
        self.graph.replay()
        self.batches += 1

    def add_eager(self, metrics: Dict):
        """A batch evaluated outside the graph (the ragged last one) joins the sums."""
        self.acc.add_(torch.tensor([float(metrics[k]) for k in self.KEYS], dtype=torch.float32, device=self.acc.device))
        self.batches += 1

    def means(self) -> Dict[str, float]:
        vals = (self.acc / max(self.batches, 1)).cpu().tolist()          # the one host sync of the evaluation
        return dict(zip(self.KEYS, vals))

    def close(self):
        torch.cuda.synchronize(self.acc.device)
        if self.graph is not None:
            self.graph.reset()
            self.graph = None
