"""SegmentedTrainStep: the fallback of GraphedTrainStep for a row-wise sharded step whose collectives cannot be captured.

GraphedTrainStep captures the whole sharded step -- RCCL collectives included -- into ONE HIP graph.  Whether a given RCCL /
driver / world size accepts collective kernels inside a stream capture is only known when it is tried on that machine; if it
refuses, replaying NOTHING would leave ~75 launches per step to Python (1.5 ms of host time for a 0.3 ms step).  This class keeps
the launches out of Python anyway: the compute BETWEEN two collectives is captured into a graph of its own (all segments share one
private memory pool, so every tensor keeps its address), and only the 3-5 collectives of a step are issued eagerly between the
replays:

    replay(segment 0)  all-to-all(ids)  replay(1)  all-to-all(rows)  replay(2)  all-to-all(row gradients)  replay(3)  all-reduce(dense)  replay(4)

How the cut is made: while the step's body runs under capture, the exchange's communicator is replaced by a wrapper whose every
collective (1) ends the capture of the current segment, (2) issues the collective eagerly and remembers the call -- same tensors,
so a replay re-issues it on the data the previous segment's replay has just produced -- and (3) begins the capture of the next
segment.  Autograd's worker threads are switched off for the capture (the backward then runs on the capturing thread: a capture
must end on the thread that began it).

Reference counterpart: none (the reference is single-process; its optional torch.compile(mode="reduce-overhead"),
scripts/train.py:223-225, is what GraphedTrainStep mirrors).
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .config import settings
from .distributed import DistComm
from .graph import GraphedTrainStep


class _SegmentingComm(DistComm):
    """DistComm's interface; every collective closes the segment being captured, runs eagerly, and opens the next segment."""

    def __init__(self, real, owner: "SegmentedTrainStep"):      # (no super().__init__: it would ask torch.distributed for a group)
        self._real, self._owner = real, owner
        self.group = getattr(real, "group", None)
        self.world, self.rank = real.world, real.rank

    def _cut(self, make_issue):
        """make_issue() runs between two captures (result buffers allocated there live as long as the step) and returns
        issue() -> result tensor, which is called once now and once per replay."""
        o = self._owner
        o._end_segment()
        issue = make_issue()
        out = issue()
        o._between.append(issue)
        o._begin_segment()
        return out

    @staticmethod
    def _need_contiguous(t: torch.Tensor, who: str):
        if not t.is_contiguous():                                # a .contiguous() here would be a launch outside every segment
            raise ValueError(f"SegmentedTrainStep: {who} needs a contiguous tensor")

    def all_to_all_equal(self, send: torch.Tensor, out=None) -> torch.Tensor:
        self._need_contiguous(send, "all_to_all_equal")

        def make():
            res = torch.empty_like(send) if out is None else out
            if isinstance(self._real, DistComm):
                return lambda: self._real.all_to_all_equal(send, res)
            return lambda: res.copy_(self._real.all_to_all_equal(send))
        return self._cut(make)

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist
        self._need_contiguous(t, "all_gather")

        def make():
            res = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)

            def issue():
                if isinstance(self._real, DistComm):
                    dist.all_gather_into_tensor(res, t, group=self._real.group)
                else:
                    res.copy_(self._real.all_gather(t))
                return res
            return issue
        return self._cut(make)

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        return self._cut(lambda: (lambda: self._real.all_reduce_sum(t)))

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        return self._cut(lambda: (lambda: self._real.all_reduce_max(t)))


class SegmentedTrainStep(GraphedTrainStep):
    """GraphedTrainStep whose capture is cut at every collective (see the module docstring).  Same interface and results."""

    segmented = True

    def _capture(self, mode: str):
        ex = getattr(self.task, "exchange", None)
        if ex is None:
            raise ValueError("SegmentedTrainStep is for the row-wise sharded task (a step without collectives is ONE graph: GraphedTrainStep)")
        self._segments: List[torch.cuda.CUDAGraph] = []
        self._all_graphs: List[torch.cuda.CUDAGraph] = []
        self._between = []
        self._pool = torch.cuda.graph_pool_handle()
        self._cap_stream = torch.cuda.Stream(device=self._dev.device)
        self._cap_mode, self._cap_ctx = mode, None
        real = ex.comm
        wrapper = _SegmentingComm(real, self)
        towers = [t for t in self._towers if getattr(t, "sync_comm", None) is real]
        ex.comm = wrapper
        for t in towers:
            t.sync_comm = wrapper
        try:
            with torch.autograd.set_multithreading_enabled(False):     # backward on THIS thread: it ends and begins captures
                self._begin_segment()
                self.result = self._body()
                self._end_segment()
        except BaseException:
            if self._cap_ctx is not None:                               # leave no capture open behind an error
                try:
                    self._cap_ctx.__exit__(None, None, None)
                except Exception:
                    pass
                self._cap_ctx = None
            raise
        finally:
            ex.comm = real
            for t in towers:
                t.sync_comm = real
        self.graph = next((g for g in self._segments if g is not None), None)

    def _begin_segment(self):
        g = torch.cuda.CUDAGraph()
        ctx = torch.cuda.graph(g, pool=self._pool, stream=self._cap_stream, capture_error_mode=self._cap_mode)
        ctx.__enter__()
        self._cap_ctx, self._cap_graph = ctx, g
        self._all_graphs.append(g)               # (every graph object lives until close(): destroying one -- an EMPTY segment's -- while a later
        #                                          segment is being captured is "operation not permitted when stream is capturing", from a destructor)

    def _end_segment(self):
        import warnings
        ctx = self._cap_ctx
        self._cap_ctx = None
        with warnings.catch_warnings(record=True) as seen:      # two collectives back to back leave an EMPTY segment between them:
            warnings.simplefilter("always")                     # torch warns and builds no executable graph -- nothing to replay
            ctx.__exit__(None, None, None)
        empty = any("empty" in str(w.message).lower() for w in seen)
        self._segments.append(None if empty else self._cap_graph)

    def _replay_graph(self):
        between = self._between
        if settings.sync_debug:                      # TT_SYNC_DEBUG=1: name every replay and every collective and wait for it (fault hunting)
            return self._replay_graph_traced()
        for i, g in enumerate(self._segments):
            if g is not None:
                g.replay()
            if i < len(between):
                between[i]()

    def _replay_graph_traced(self):
        import sys
        dev = self._dev.device
        for i, g in enumerate(self._segments):
            if g is not None:
                print(f"[tt] segment {i} replay ...", end="", file=sys.stderr, flush=True)
                g.replay()
                torch.cuda.synchronize(dev)
                print(" done", file=sys.stderr, flush=True)
            if i < len(self._between):
                print(f"[tt] collective {i} ...", end="", file=sys.stderr, flush=True)
                self._between[i]()
                torch.cuda.synchronize(dev)
                print(" done", file=sys.stderr, flush=True)

    def collectives_per_step(self) -> int:
        return len(self._between)

    def close(self):
        segs = getattr(self, "_all_graphs", [])
        super().close()
        for g in segs:
            try:
                g.reset()
            except Exception:
                pass
        self._segments, self._all_graphs, self._between = [], [], []
