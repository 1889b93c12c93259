"""UnrolledTrainStep: U whole training steps -- hand-overs included -- captured into ONE HIP graph and replayed with one launch.

Why: on this runtime one graph launch costs ~8 us on top of its nodes (K-node graphs back to back: 1.79 us per node + 8.0 us per
launch, whatever runtime switch is set: tools/probe/graph_setparams.hip, profiles/NOTES.md round 4), i.e. 3.4 % of a 0.233-ms step.
U steps per launch pay it once.

What makes it possible: the hand-over launch (tt_batch_ingest*: batch buffers, step scalars, fused rows, bf16 weight shadows) has to
run BETWEEN the steps of the graph -- step j + 1 reads the weights step j has written -- so it is captured as a kernel node of the
graph, once per step, each writing its own set of static buffers ("lane"); before every replay the host re-points those U nodes at
the incoming batches and the next U ring slots (tt_handover_retarget: hipGraphExecKernelNodeSetParams, 1.4 us per node; replays
already enqueued keep the arguments they were launched with).  Everything else is GraphedTrainStep's step, U times in a row.

Results: step j of a replay == step j of U single-step replays, bit for bit (test_unrolled_step_equals_single_steps).

Reference counterpart: none (torch.compile(mode="reduce-overhead"), scripts/train.py:223-225, replays one step per launch).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import _lib as L
from .graph import GraphedTrainStep
from .kjt import KeyedJaggedTensor


class UnrolledTrainStep(GraphedTrainStep):
    """GraphedTrainStep with `unroll` steps per graph launch: `step_many(batches)` / `steps_from_store(..., offsets)` run `unroll`
    steps; `step` / `step_from_store` (one step: an epoch's remainder) go through a single-step sibling captured on first use."""

    def __init__(self, task, optimizer, example_batch: Dict, unroll: int = 2, **kw):
        if int(unroll) < 2:
            raise ValueError("UnrolledTrainStep: unroll must be >= 2 (one step per launch is GraphedTrainStep)")
        self.unroll = self._steps_per_replay = int(unroll)
        self._example, self._kw = example_batch, dict(kw)
        self._lanes, self._nodes, self._exec, self.single = [], [], None, None
        super().__init__(task, optimizer, example_batch, **kw)
        # (the base counted the U captured hand-overs + U bodies, + 1 for a hand-over in front that this form does not have)
        self.library_launches = (self.library_launches - 1) // self.unroll

    # ---- capture ------------------------------------------------------------------------------------------------------
    def _capture(self, mode: str):
        if self._ingest is None or self._x_static is not None or getattr(self.task, "exchange", None) is not None:
            raise ValueError("UnrolledTrainStep needs the key-major hand-over of an unsharded step (GraphedTrainStep._setup_ingest) "
                             "without the fused hand-over + lookup option")
        store, embs, rows_km, ids, B = self._ingest
        self._lanes = [dict(static=self.static, ingest=self._ingest, rows_sm=self._rows_sm)]
        for _ in range(1, self.unroll):                          # lanes 1 .. U-1: their own static buffers and row arrays
            st = {side: {"dense": self.static[side]["dense"].clone(),
                         "kjt": KeyedJaggedTensor(self.static[side]["kjt"].keys(), self.static[side]["kjt"].values().clone())}
                  for side in ("notice", "company")}
            lane_ids = [st[side]["kjt"].values() for side in ("notice", "company")]
            self._lanes.append(dict(static=st, ingest=(store, embs, torch.empty_like(rows_km), lane_ids, B),
                                    rows_sm=None if self._rows_sm is None else torch.empty_like(self._rows_sm)))
        dev = self._dev.device
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)       # (the node handles below belong to the graph: it must stay alive)
        self.results, self._nodes = [], []
        try:
            with torch.cuda.graph(self.graph, capture_error_mode=mode):
                for j in range(self.unroll):
                    self._select(j)
                    self._run_ingest([(self._dev, self._host)], None)      # captured: node j (its arguments are replaced before every replay)
                    node = L.handover_captured_node(dev)
                    if not node:
                        raise RuntimeError("UnrolledTrainStep: the captured hand-over launch left no graph node")
                    self._nodes.append(node)
                    self.results.append(self._body())
        finally:
            self._select(0)
        self.graph.instantiate()
        self._exec = self.graph.raw_cuda_graph_exec()
        self.result = self.results[-1]

    def _select(self, j: int):
        lane = self._lanes[j]
        self.static, self._ingest, self._rows_sm = lane["static"], lane["ingest"], lane["rows_sm"]

    # ---- U steps per launch -------------------------------------------------------------------------------------------
    def step_many(self, batches: Sequence[Dict], after_each: Optional[Callable[[], None]] = None) -> List:
        """Train on the `unroll` batches in order; returns their results (static: overwritten by the next replay).  after_each():
        called after each step's scalars have been read off the optimiser (the LR scheduler's step() belongs there)."""
        if len(batches) != self.unroll:
            raise ValueError(f"step_many takes {self.unroll} batches, got {len(batches)}")
        dev = self._dev.device
        try:
            for j, batch in enumerate(batches):
                self._select(j)
                pairs, src_ids = self._handover_pairs(batch)
                with L.handover_retarget(dev, self._exec, self._nodes[j]):
                    self._run_ingest(pairs + [self._fill_slot(j)], src_ids)
                if after_each is not None:
                    after_each()
        finally:
            self._select(0)
        self._replay()
        self._mark_slot()
        return self.results

    def steps_from_store(self, notice_store, company_store, pairs: torch.Tensor, order: Optional[torch.Tensor], offsets: Sequence[int],
                         after_each: Optional[Callable[[], None]] = None) -> List:
        """step_from_store for `unroll` batches (their offsets into `order`, or into `pairs` when order is None), one launch."""
        if len(offsets) != self.unroll:
            raise ValueError(f"steps_from_store takes {self.unroll} offsets, got {len(offsets)}")
        dev = self._dev.device
        try:
            for j, off in enumerate(offsets):
                self._select(j)
                with L.handover_retarget(dev, self._exec, self._nodes[j]):
                    self._ingest_from_store(notice_store, company_store, pairs, order, int(off), ahead=j)
                if after_each is not None:
                    after_each()
        finally:
            self._select(0)
        self._replay()
        self._mark_slot()
        return self.results

    # ---- one step (an epoch's remainder): a single-step sibling ------------------------------------------------------------
    def _single(self) -> GraphedTrainStep:
        if self.single is None:
            kw = dict(self._kw)
            kw["preserve_state"] = True                          # its warm-up steps must leave no trace: training is under way
            kw["metric_sums"] = self.metric_sums                 # one epoch total for both (None: no accumulation)
            self.single = GraphedTrainStep(self.task, self.opt, self._example, **kw)
        return self.single

    def step(self, batch: Optional[Dict] = None):
        return self._single().step(batch)

    def step_from_store(self, notice_store, company_store, pairs: torch.Tensor, order: Optional[torch.Tensor] = None, offset: int = 0):
        return self._single().step_from_store(notice_store, company_store, pairs, order, offset)

    def close(self):
        if self.single is not None:
            self.single.close()
            self.single = None
        lanes, self._lanes = self._lanes, []
        if lanes:
            self._select_from(lanes[0])
        super().close()
        self.results, self._nodes, self._exec = [], [], None

    def _select_from(self, lane):
        self.static, self._ingest, self._rows_sm = lane["static"], lane["ingest"], lane["rows_sm"]
