#!/usr/bin/env python3
"""bench.py -- training pairs/sec of the two-tower step on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]

Self-contained for every N: with --gpus N > 1 and no WORLD_SIZE in the environment it starts its own N rank processes (one
per GPU, before anything touches a GPU) and rank 0 prints the JSON line; under `python -m torch.distributed.run` it uses the
ranks it is given.  When the box has fewer GPUs than ranks (a one-GPU rehearsal) the ranks share a device and their
collectives are staged through gloo (RCCL refuses two ranks on one device): every kernel is still the product path, the line
is labelled "rehearsal" and runs eagerly (with --dist-segmented: as replayed graph segments with the staged collectives between them).

Step = TwoTowerTrainTask forward (return_metrics=True) + loss.backward() + optimiser step + LR schedule on one synthetic
batch whose ids and dense features are already resident in HBM.

  N = 1 : BASELINE.json configs[1] -- real 32+6 key schema, per-key vocabularies scaled to 1 M rows per tower, E=32, towers
          [128,64], final 64, batch 8192, in-batch negatives.
  N > 1 : BASELINE.json configs[2] -- 100 M notice + 10 M company rows, sharded row-wise over the N GPUs (--zipf 1.2:
          configs[3]).  `value`: every GPU keeps the N = 1 job's batch (8192 pairs PER GPU, per-rank in-batch negatives and
          BN statistics: the data-parallel job whose global batch grows with N; per-GPU work fixed, "scaling": "weak" -- the
          leg in which north_star's 1 -> 8 scaling on 100 M-row tables is defined).  The line also carries `strong_scaling`
          (GLOBAL batch 8192 = 8192/N pairs per GPU, global in-batch negatives + SyncBN: the single-process job at batch 8192
          split over N GPUs; latency-bound) and `one_gpu_same_tables` (the unsharded single-GPU step on the SAME 100 M +
          10 M-row tables, measured by rank 0 in the same run: the like-for-like denominator of a speed-up).

Prints ONE JSON line (rank 0).  `roofline` is the embedding-lookup kernel (the kernel BASELINE.json's metric names);
`mfma` the score GEMMs; `value_with_h2d` the same step fed from pinned host batches through a staging stream;
`cpu_baseline` times oracle/oracle_torch.py (checker code, used here only as the reported baseline) on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense bf16 (MI355X_MICROARCH.md)
MFMA_FP8_PEAK_TFLOPS = 5000.0     # dense fp8 through the block-scaled f8f6f4 instruction


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 2000 timed steps behind 200 warm-up steps = half a second of GPU time.  Round 4 measured what a short region costs: 100 steps
    # behind 20 (23 ms) read 0.2308-0.2332 ms per step where 4000 behind 200 read 0.2263-0.2265 -- the first replays of a cold process are
    # slower and the region's fixed ends (first enqueue, final synchronise: ~0.15 ms) weigh 0.6 % at 100 steps (profiles/NOTES.md)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=8192, help="pairs per step: per GPU at N = 1 and in the weak leg, GLOBAL in the strong leg")
    ap.add_argument("--rows-notice", type=int, default=None, help="table rows, notice tower (default 1 M at N = 1; 100 M over all GPUs at N > 1)")
    ap.add_argument("--rows-company", type=int, default=None, help="table rows, company tower (default 1 M at N = 1; 10 M over all GPUs at N > 1)")
    ap.add_argument("--zipf", type=float, default=None, help="Zipf alpha for ids (default uniform; 1.2 = configs[3])")
    ap.add_argument("--final-dim", type=int, default=64, help="final_embedding_dim (BASELINE configs[4] uses 256 at --batch 65536)")
    ap.add_argument("--hidden", default="128,64", help="tower_hidden_dims (BASELINE configs: 128,64; the reference driver scripts/train.py:106-107 trains 512,256 with --final-dim 128)")
    ap.add_argument("--optimizer", choices=["fused_sparse", "fused_dense", "torch_adam"], default="fused_sparse")
    ap.add_argument("--score-dtype", choices=["bf16", "fp32", "fp8"], default="bf16")
    ap.add_argument("--mlp-dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--pool", type=int, default=8, help="distinct pre-generated batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work budget of the cpu_baseline sample")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="graph: whole step replayed from one captured HIP graph; eager: launched from Python")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded-table path even on one GPU")
    ap.add_argument("--legs", default=None, help="multi-GPU: comma list of strong,weak,one_gpu (default all three)")
    ap.add_argument("--negatives", choices=["local", "global"], default=None,
                    help="multi-GPU, overrides the leg's default: in-batch negatives of the rank's own batch or of the GLOBAL batch")
    ap.add_argument("--sync-bn", action="store_true", help="multi-GPU: BatchNorm statistics over all ranks' rows also in the weak leg")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-to-device-inclusive leg")
    ap.add_argument("--breakdown", action="store_true", help="extra instrumented pass: per-kernel HIP-event times")
    ap.add_argument("--fp8-grad", type=int, default=1, choices=[0, 1], help="TT_OPT_FP8_GRAD with --score-dtype fp8: 1 e4m3 gradient products "
                    "(default), 0 bf16 gradient products")
    ap.add_argument("--no-riders", action="store_true", help="plan compaction / loss reduction as launches of their own (A/B of TT_OPT_DEFER_RIDERS)")
    ap.add_argument("--dist-eager", action="store_true", help="sharded path launched from Python instead of replayed (analysis)")
    ap.add_argument("--dist-segmented", action="store_true", help="sharded path: the compute between two collectives as graphs of its own, the "
                    "collectives eager (the fallback when a capture with RCCL nodes fails; this flag forces it)")
    ap.add_argument("--no-lookup-profile", action="store_true", help="do not stamp the lookup launches (no `roofline` object then)")
    ap.add_argument("--lookup-wg-dump", default=None, help="write the lookup's per-workgroup stamps to this .npy (tools/lookup_wg.py)")
    ap.add_argument("--fused-handover", action="store_true", help="hand-over and lookup as ONE launch (tt_batch_ingest_lookup; measured neutral, off by default)")
    ap.add_argument("--lookup-from-ids", action="store_true", help="the captured lookup decodes the int64 ids itself (round 3's kernel input) instead of "
                    "reading the hand-over launch's precomputed rows")
    ap.add_argument("--lookup-nt", action="store_true", help="TT_OPT_LOOKUP_NT: non-temporal stores of the looked-up rows (A/B)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the HBM-resident lookup leg and the configs[4] leg of the default run")
    ap.add_argument("--unroll", type=int, default=1, help="training steps per graph launch (unrolled.UnrolledTrainStep: the hand-over launches become "
                                                          "graph nodes re-pointed per launch; one graph launch costs ~8 us on this runtime); unsharded step only")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch
def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n: int, argv) -> int:
    """Start n fresh rank processes of this script (never an exec from a process that has touched a GPU: nothing has
    here).  Rank 0 keeps our stdout (the JSON line); the other ranks' stdout goes to stderr."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                c = p.poll()
                if c is None:
                    continue
                pending.remove(p)
                if c != 0 and rc == 0:
                    rc = c
                    for q in pending:                    # one rank failed: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ------------------------------------------------------------------------------------------------ one timed leg
class Leg:
    """One workload: task + optimiser + batch pool (+ captured graph), timed over K steps."""

    def __init__(self, args, ctx, B, rows_n, rows_c, sharded, negatives="local", sync_bn=False, label=""):
        import torch
        import jodalrob_twotower_amd as tt
        from jodalrob_twotower_amd import ops, synthetic
        from jodalrob_twotower_amd.optim import FusedAdam
        self.args, self.ctx, self.B, self.sharded, self.label = args, ctx, B, sharded, label
        self.negatives, self.sync_bn = negatives, sync_bn
        dev, world, rank = ctx["dev"], ctx["world"], ctx["rank"]
        schema = synthetic.load_real_schema(ROOT / "jodalrob-twotower_amd" / "schema_real.json")
        self.keys_n, self.keys_c = schema["notice"]["categorical"], schema["company"]["categorical"]
        self.vocab_n = synthetic.scale_vocabs(schema["notice"]["vocab_sizes"], rows_n)
        self.vocab_c = synthetic.scale_vocabs(schema["company"]["vocab_sizes"], rows_c)
        self.E, self.hidden, self.D, self.din_n, self.din_c = 32, [int(h) for h in args.hidden.split(",")], args.final_dim, 256, 128
        tmp = tempfile.mkdtemp(prefix="tt_bench_")
        meta = synthetic.write_metadata(Path(tmp) / "metadata.csv", {"notice": dict(zip(self.keys_n, self.vocab_n)),
                                                                     "company": dict(zip(self.keys_c, self.vocab_c))})
        torch.manual_seed(1234)
        grad_mode = "sparse" if args.optimizer == "fused_sparse" else "dense"
        common = dict(metadata_path=str(meta), categorical_embedding_dim=self.E, notice_dense_input_dim=self.din_n,
                      company_dense_input_dim=self.din_c, tower_hidden_dims=self.hidden, final_embedding_dim=self.D, dropout_rate=0.1,
                      temperature=1.0, device=dev, embedding_grad=grad_mode, score_dtype=args.score_dtype, mlp_dtype=args.mlp_dtype)
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            if sharded:
                from jodalrob_twotower_amd.distributed import create_distributed_train_task
                # default exchange: dedup-first, fixed-capacity all-to-alls -- the whole step incl. RCCL is one graph replay
                self.task = create_distributed_train_task(self.keys_n, self.keys_c, negatives=negatives, sync_bn=sync_bn,
                                                          comm=ctx["comm"], **common)
            else:
                self.task = tt.create_two_tower_train_task(self.keys_n, self.keys_c, **common)
        task = self.task
        task.train()
        task._pair_check_done = True            # skip the first-call diagnostic printout (host sync)
        if args.fp8_grad != 1:
            from jodalrob_twotower_amd import _lib
            _lib.set_option(dev, _lib.TT_OPT_FP8_GRAD, args.fp8_grad)
        if args.optimizer == "torch_adam":
            self.opt = torch.optim.Adam(task.parameters(), lr=1e-3, weight_decay=1e-5)
        else:
            self.opt = FusedAdam.for_task(task, lr=1e-3, weight_decay=1e-5)
        total_steps = args.steps + args.warmup
        warm = max(1, int(total_steps * 0.05))
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda s: s / warm if s < warm else 1.0, last_epoch=-1)
        # global-batch legs: rank r holds rows [r*B_local, (r+1)*B_local) of the global batch -- its own seeded slice
        self.pool = [synthetic.make_batch(B, self.vocab_n, self.vocab_c, self.keys_n, self.keys_c, self.din_n, self.din_c, dev,
                                          seed=1234 + 7919 * (rank * args.pool + i), zipf_alpha=args.zipf) for i in range(args.pool)]
        # (a rehearsal's host-staged collectives cannot be captured; with --dist-segmented they run between the replayed segments)
        eager_only = (ctx["staged"] and not args.dist_segmented) or args.dist_eager
        self.use_graph = (args.mode == "graph" and args.optimizer == "fused_sparse" and not eager_only) if sharded else \
            (args.mode == "graph" and args.optimizer != "torch_adam")
        self.gstep, self.unroll = None, 1
        self.profile = ops.LookupProfile(dev) if (self.use_graph and not args.no_lookup_profile) else None
        ex = getattr(task, "exchange", None)
        if sharded and ex is not None and hasattr(ex, "reset_capacity"):
            # fixed-capacity exchange: size the buckets for the largest need over the whole batch pool (one forward per
            # batch, outside the timed region), not just for the batch the capture happens to see
            caps = []
            with torch.no_grad():
                for b in self.pool:
                    ex.reset_capacity()
                    task(b, return_metrics=False)
                    caps.append(ex.C)
            ex.C = max(caps)
            torch.cuda.synchronize()
        if self.use_graph:
            from jodalrob_twotower_amd.graph import GraphedTrainStep
            kw = dict(return_metrics=True, warmup=3, defer_riders=not args.no_riders, preserve_state=False)   # (warm-up steps: part of the bench's own)
            self.launch_form = "hip graph replay"
            try:
                if sharded and args.dist_segmented:
                    raise RuntimeError("--dist-segmented")
                if args.unroll > 1 and not sharded and not args.fused_handover:
                    from jodalrob_twotower_amd.unrolled import UnrolledTrainStep
                    self.gstep = UnrolledTrainStep(task, self.opt, self.pool[0], unroll=args.unroll, **kw)
                    self.unroll = args.unroll
                    self.launch_form = f"hip graph replay, {args.unroll} steps (hand-overs included) per graph launch"
                    if args.steps % args.unroll or args.warmup % args.unroll:
                        self.gstep._single()                # remainder steps: the single-step sibling, captured here and not in the timed region
                else:
                    self.gstep = GraphedTrainStep(task, self.opt, self.pool[0], **kw)
                task._bench_gstep = self.gstep
            except Exception as e:                      # a capture that fails on some RCCL / world size must not lose the run
                if not sharded:
                    raise
                # first fallback: the compute between two collectives as graphs of its own, the 3-5 collectives eager between the
                # replays (segmented.SegmentedTrainStep); last resort: ~75 launches per step from Python
                print(f"[bench] rank {rank}: one-graph capture of the sharded step not used ({type(e).__name__}: {e}); capturing it in segments",
                      file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                try:
                    from jodalrob_twotower_amd.segmented import SegmentedTrainStep
                    self.gstep = SegmentedTrainStep(task, self.opt, self.pool[0], **kw)
                    task._bench_gstep = self.gstep
                    self.launch_form = f"segmented graph replay ({self.gstep.collectives_per_step()} eager collectives between the segments)"
                except Exception as e2:
                    print(f"[bench] rank {rank}: segmented capture failed too ({type(e2).__name__}: {e2}); running eagerly", file=sys.stderr, flush=True)
                    if self.profile is not None:
                        self.profile.close()
                    self.gstep, self.profile = None, None
                    self.launch_form = "eager"
                    torch.cuda.synchronize()

    def step(self, i, batch=None, eager=False):
        b = batch if batch is not None else self.pool[i % len(self.pool)]
        if self.gstep is not None and not eager:
            res = self.gstep.step(b)
            self.sched.step()
            return res
        self.opt.zero_grad()
        res = self.task(b, return_metrics=True)
        res["loss"].backward()
        self.opt.step()
        self.sched.step()
        return res

    def steps(self, start: int, count: int):
        """issues steps start .. start + count - 1 (an unrolled captured step: `unroll` of them per graph launch, a remainder one by one)"""
        res, i, U = None, 0, self.unroll
        while i < count:
            if U > 1 and count - i >= U:
                res = self.gstep.step_many([self.pool[(start + i + j) % len(self.pool)] for j in range(U)], after_each=self.sched.step)[-1]
                i += U
            else:
                res = self.step(start + i)
                i += 1
        return res

    def run(self):
        """W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; MAX over ranks."""
        import torch
        from jodalrob_twotower_amd import ops
        args, ctx = self.args, self.ctx
        res = self.steps(0, args.warmup)
        ctx["fence"]()
        lookup_name = "tt_embed_lookup_fwd" if not self.sharded else "tt_embed_lookup_fwd[place]"   # sharded: the launch that fills the tower inputs
        timer = ops.KernelTimer(names=[lookup_name])
        if self.gstep is None:
            ops.set_timer(timer)
        elif self.profile is not None:
            self.profile.reset()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = self.steps(args.warmup, args.steps)
        t_enqueue = time.perf_counter() - t0                     # host time to issue the K steps (no sync inside)
        ctx["fence"]()
        dt = time.perf_counter() - t0
        ops.set_timer(None)
        # the captured step's hand-over launch also does the lookup (tt_batch_ingest_lookup)?  Then the stamps are that launch's:
        # its tile workgroups (the gather phase) come first, the copy roles behind them
        self.fused_handover = self.gstep is not None and getattr(self.gstep, "_x_static", None) is not None
        self.handover_body_us, self.handover_event_us = [], []
        if self.profile is not None and self.fused_handover:
            tiles = ops.ingest_lookup_tiles(self.B, [len(self.keys_n), len(self.keys_c)])
            self.lookup_us = self.profile.durations_us(0, tiles)
            self.handover_body_us = self.profile.durations_us()
        else:
            self.lookup_us = self.profile.durations_us() if self.profile is not None else []
        if self.profile is not None and self.args.lookup_wg_dump:     # per-workgroup stamps of the last launches (analysis)
            import numpy as _np
            _np.save(self.args.lookup_wg_dump, self.profile.ring.cpu().numpy())
        self.dispatch_us = lookup_dispatch_overhead_us(self.task, self.pool, ctx["dev"], self.profile) \
            if (self.profile is not None and not self.fused_handover) else None
        if self.profile is not None:
            self.profile.close()
        ex = getattr(self.task, "exchange", None)
        if self.sharded and ex is not None and hasattr(ex, "check_overflow"):
            ex.check_overflow()                                  # a bucket overflow in the timed region invalidates the result
        dt = ctx["max_over_ranks"](dt)
        self.timer_summary = timer.summary()
        self.lookup_name = lookup_name
        self.dt, self.t_enqueue, self.loss = dt, t_enqueue, float(res["loss"].detach())
        # median of per-step DEVICE times (HIP events at the step boundaries) in a second pass: a short timed region is
        # fragile evidence on its own, and events between replays would perturb the region above
        U = self.unroll                                                 # (an unrolled step: events between graph launches, time / U)
        n2 = max(min(args.steps, 50) // U, 1)
        t2 = ops.KernelTimer(names=["tt_batch_ingest_lookup"])          # the hand-over launch is eager: HIP events on its stream bracket it
        if self.fused_handover:
            ops.set_timer(t2)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n2 + 1)]
        evs[0].record()
        for i in range(n2):
            self.steps(args.warmup + args.steps + i * U, U)
            evs[i + 1].record()
        torch.cuda.synchronize()
        ops.set_timer(None)
        per = sorted(evs[i].elapsed_time(evs[i + 1]) / U for i in range(n2))
        self.device_ms_median = per[len(per) // 2] if per else None
        self.handover_event_us = sorted(a.elapsed_time(b) * 1e3 for a, b in t2.records.get("tt_batch_ingest_lookup", []))
        return dt

    def close(self):
        import gc
        import torch
        if self.task is not None and hasattr(self.task, "_bench_gstep"):
            self.task._bench_gstep = None
        if self.gstep is not None:
            self.gstep.close()                   # graph + pool first: its nodes reference the communicator
        self.gstep = self.task = self.opt = self.sched = self.pool = None
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def run(args):
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and args.gpus > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()                      # (counting devices does not initialise the GPU on this image)
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); the product path has no CPU fallback")
    staged = world > 1 and world > ndev                   # rehearsal: several ranks on one device
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    from jodalrob_twotower_amd import ops
    dist, comm, saved_stdout = None, None, None
    sharded = world > 1 or args.force_dist
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        # RCCL and gloo print banners to stdout when their communicators come up: park fd 1 on stderr until the JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        cpu_group = None
        if staged:
            from jodalrob_twotower_amd.distributed import HostStagedComm
            dist.init_process_group("gloo", rank=rank, world_size=world)
            comm = HostStagedComm()
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            # the bench's own barriers and the max-over-ranks of the timing go through a host-side gloo group: they are not
            # part of the step, and the step's communicator then carries nothing but the captured collectives
            cpu_group = dist.new_group(backend="gloo")

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(group=cpu_group)
        torch.cuda.synchronize()

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=cpu_group)
        return float(t.item())

    from jodalrob_twotower_amd.config import settings as _settings
    _settings.graph_ingest_lookup = bool(args.fused_handover)
    _settings.graph_ingest_rows = not args.lookup_from_ids
    if args.lookup_nt:
        from jodalrob_twotower_amd import _lib as _l
        _l.set_option(dev, _l.TT_OPT_LOOKUP_NT, 1)
    ctx = dict(dev=dev, world=world, rank=rank, staged=staged, comm=comm, fence=fence, max_over_ranks=max_over_ranks)
    out = None
    try:
        if world == 1:
            out = bench_single(args, ctx, sharded)
        else:
            out = bench_multi(args, ctx)
    finally:
        if dist is not None:
            import gc
            gc.collect()
            torch.cuda.synchronize()
            if rank == 0 and out is not None:
                sys.stdout.flush()
                os.dup2(saved_stdout, 1)
                print(json.dumps(out), flush=True)
                os.dup2(2, 1)
            dist.destroy_process_group()
        elif out is not None:
            print(json.dumps(out), flush=True)


def _tower_io_dtype() -> str:
    from jodalrob_twotower_amd.config import settings
    return settings.tower_io_dtype


def base_line(args, leg, world, scaling, B_global):
    dt = leg.dt
    bf = args.score_dtype == "bf16" and args.mlp_dtype == "bf16"
    x_bf16 = args.mlp_dtype == "bf16" and _tower_io_dtype() in ("x", "both")
    return {
        "metric": "training pairs/sec at batch 8192 (embedding-lookup HBM GB/s in roofline)",
        "value": B_global * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "device_ms_per_step_median": leg.device_ms_median,
        "host_enqueue_ms_per_step": leg.t_enqueue / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": ("bf16" if bf else f"score {args.score_dtype} / mlp {args.mlp_dtype}") +
                 " MFMA operands, f32 accumulate; f32 tables, master weights and activations" +
                 (" (the tower input x, which the GEMMs round to bf16 anyway, is stored bf16)" if x_bf16 else ""),
        "data": "synthetic",
    }


def roofline_fused(args, leg):
    """The hand-over + lookup launch (ingest_lookup_kernel, tt_batch_ingest_lookup): the kernel BASELINE.json's metric names, since
    round 4 one launch with the batch hand-over.  Algorithmic bytes per pair = the lookup's (table row + i64 id + output
    row per key) + the hand-over's (dense features read + written, ids read + written into the static buffers, key-major fused
    rows written).  mean_launch_us: HIP events around the eager launch on its stream, every step of bench.py's second pass;
    mean_body_us: min start .. max end of ALL its workgroups' device-clock stamps in the timed region; lookup_phase: the same over
    the tile workgroups only (ids -> rows -> gathers -> stores: the embedding lookup inside the launch)."""
    x_bf16 = args.mlp_dtype == "bf16" and _tower_io_dtype() in ("x", "both")
    K_tot, E, B = len(leg.keys_n) + len(leg.keys_c), leg.E, leg.B
    s_out = 2 if x_bf16 else 4
    lookup_pp = K_tot * (E * 4 + 8 + E * s_out)
    handover_pp = 2 * 4 * (leg.din_n + leg.din_c) + K_tot * (8 + 8 + 4)
    algo = B * (lookup_pp + handover_pp)
    ev = leg.handover_event_us
    launch_us = (sum(ev) / len(ev)) if ev else None
    body = leg.handover_body_us
    body_us = (sum(body) / len(body)) if body else None
    ph = leg.lookup_us
    ph_us = (sum(ph) / len(ph)) if ph else None
    use = launch_us or body_us
    achieved = algo / (use * 1e-6) / 1e9 if use else None
    traffic, src = None, None
    pmc = ROOT / "profiles" / "r04_ingest_lookup_pmc.json"
    if pmc.exists() and B == 8192 and sum(leg.vocab_n) == 1_000_000 and sum(leg.vocab_c) == 1_000_000 and args.zipf is None and x_bf16:
        try:
            traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
            src = f"profiles/{pmc.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same launch on the same workload, round 4; not this run)"
        except Exception:
            traffic = None
    return {"kernel": "ingest_lookup_kernel (tt_batch_ingest_lookup: batch hand-over + embedding lookup in one launch)", "bound": "hbm",
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
            "traffic": traffic, "traffic_source": src,
            "hbm_frac_physical": (traffic / (use * 1e-6) / 1e9 / HBM_PEAK_GBPS) if (traffic and use) else None,
            "algorithmic_bytes_per_launch": algo,
            "algorithmic_bytes_per_pair": {"lookup (table row + i64 id + output row per key)": lookup_pp,
                                           "hand-over (dense features in + out, ids in + out, key-major rows out)": handover_pp},
            "launches_timed": len(ev), "mean_launch_us": launch_us, "mean_body_us": body_us,
            "lookup_phase": {"what": "the launch's tile workgroups: ids -> clamped rows -> row gathers -> stores into the tower inputs",
                             "mean_us": ph_us, "algorithmic_bytes": B * lookup_pp,
                             "GBps": (B * lookup_pp / (ph_us * 1e-6) / 1e9) if ph_us else None,
                             "frac": (B * lookup_pp / (ph_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if ph_us else None, "launches_timed": len(ph)},
            "timed_in": "mean_launch_us: HIP events around the eager hand-over launch on its stream (second pass, every step); mean_body_us / "
                        "lookup_phase: device-clock (s_memrealtime, 100 MHz) stamps of the launch's workgroups in the timed region"}


def roofline_of(args, leg, world):
    """the lookup kernel: algorithmic bytes per launch / mean launch duration, measured live in the timed region"""
    if getattr(leg, "fused_handover", False) and not leg.sharded:
        return roofline_fused(args, leg)
    x_bf16 = args.mlp_dtype == "bf16" and _tower_io_dtype() in ("x", "both")
    K_tot, E, B = len(leg.keys_n) + len(leg.keys_c), leg.E, leg.B
    s_out = 2 if x_bf16 else 4                                # the lookup writes straight into the tower input x
    # table row + index + output row per key (10,032 / 7,600 B with the i64 id).  A captured step's lookup reads the
    # 4-byte fused row the hand-over launch left for it (tt_embed_lookup_rows_fwd) instead of the 8-byte id: 7,448 B per pair
    rows_mode = getattr(leg.gstep, "_rows_sm", None) is not None
    s_idx = 4 if rows_mode else 8
    bytes_per_pair = K_tot * (E * 4 + s_idx + E * s_out)
    ex = getattr(leg.task, "exchange", None)
    if leg.sharded and x_bf16 and getattr(ex, "wire_bf16", False):
        # sharded step: the stamped launch PLACES the exchanged rows, which arrive as bf16 (the f32 table rows are read by
        # tt_gather_rows on their owners): bf16 row in + i64 index + bf16 row out
        bytes_per_pair = K_tot * (E * 2 + 8 + E * 2)
    algo_bytes = B * bytes_per_pair                          # one launch = one batch on this GPU
    n_launch, lookup_ms = leg.timer_summary.get(leg.lookup_name, (0, float("nan")))
    body_us = None
    if leg.lookup_us:
        body_us = sum(leg.lookup_us) / len(leg.lookup_us)
        n_launch, lookup_ms = len(leg.lookup_us), (body_us + (leg.dispatch_us or 0.0)) * 1e-3
    achieved = algo_bytes / (lookup_ms * 1e-3) / 1e9 if lookup_ms == lookup_ms and lookup_ms > 0 else None
    traffic, src = None, None
    pmc = ROOT / "profiles" / (("r04_lookup_pmc.json" if rows_mode else "lookup_pmc_bf16out.json") if x_bf16 else "lookup_pmc.json")
    if pmc.exists() and not leg.sharded and B == 8192 and sum(leg.vocab_n) == 1_000_000 and sum(leg.vocab_c) == 1_000_000 and args.zipf is None:
        try:                                                 # the PMC passes were taken on this exact launch (configs[1]), not in this run
            traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
            src = (f"profiles/{pmc.name} (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same launch inside the replayed step, "
                   "collected in round 4 with tools/r04_profiles.sh; not this run)" if rows_mode else
                   f"profiles/{pmc.name} (static: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same launch, not this run)")
        except Exception:
            traffic = None
    return {"kernel": "lookup_wave_kernel (tt_embed_lookup_rows_fwd: precomputed fused rows)" if rows_mode else "lookup_wave_kernel (tt_embed_lookup_fwd)",
            "index_bytes_per_key": s_idx, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic, "traffic_source": src,
            # physical HBM traffic / time / peak: the hot binary-key rows are L2 hits, so fewer bytes than the algorithmic count move
            "hbm_frac_physical": (traffic / (lookup_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (traffic and achieved) else None,
            "algorithmic_bytes_per_launch": algo_bytes, "launches_timed": n_launch, "mean_launch_us": lookup_ms * 1e3,
            "mean_body_us": body_us, "dispatch_overhead_us": leg.dispatch_us,
            "timed_in": "timed region, HIP events on the launch stream" if leg.gstep is None else
                        "timed region (graph replay), every launch: mean_launch_us = mean_body_us + dispatch_overhead_us. mean_body_us: device-clock (s_memrealtime, 100 MHz) stamps, min start .. "
                        "max end over the kernel's workgroups, ring of per-launch slots read after the region "
                        "(HIP events cannot bracket one kernel inside a replayed graph)"}


def config_of(args, leg, world, ctx, B_global, workload):
    ex = getattr(leg.task, "exchange", None)
    x_bf16 = args.mlp_dtype == "bf16" and _tower_io_dtype() in ("x", "both")
    cfg = {"workload": workload, "batch_per_gpu": leg.B, "global_batch": B_global, "rows_notice": sum(leg.vocab_n),
           "rows_company": sum(leg.vocab_c), "ids": "uniform" if args.zipf is None else f"zipf({args.zipf})",
           "optimizer": args.optimizer, "score_dtype": args.score_dtype, **({"fp8_grad": args.fp8_grad} if args.score_dtype == "fp8" else {}), "mlp_dtype": args.mlp_dtype,
           "launch": getattr(leg, "launch_form", "hip graph replay") if leg.gstep is not None else "eager",
           "batch_handover": (("one launch: copies + key-major rows for the dedup plan + the embedding lookup into the towers' inputs (tt_batch_ingest_lookup)"
                               if getattr(leg.gstep, "_x_static", None) is not None else
                               "one launch: copies + key-major rows for the dedup plan (tt_batch_ingest)")
                              if getattr(leg.gstep, "_ingest", None) is not None else "one launch: copies (tt_copy_multi)") if leg.gstep is not None else "none"}
    if leg.gstep is not None and hasattr(leg.gstep, "library_launches"):
        cfg["launches_per_step"] = leg.gstep.library_launches      # kernels of libtwotower_hip.so per step (RCCL's own kernels come on top)
    if leg.sharded and ex is not None and hasattr(ex, "C"):
        cfg.update({"exchange_capacity_rows_per_peer": ex.C,
                    "exchange_bytes_per_rank_fwd": world * ex.C * leg.E * (2 if (x_bf16 and ex.wire_bf16) else 4),
                    "exchange_bytes_per_rank_bwd": world * ex.C * leg.E * (2 if ex.grad_wire_bf16 else 4)})
    cfg["parallelism"] = "single GPU" if not leg.sharded else (
        f"row-wise sharded tables x{world} (dedup-first fixed-capacity all-to-all" + (", RCCL inside the graph" if (leg.gstep is not None and not getattr(leg.gstep, "segmented", False)) else
                                                                                 ", collectives eager between replayed segments" if leg.gstep is not None else "") +
        ") + data parallel towers, " + f"{leg.negatives} in-batch negatives" + (", SyncBN" if leg.sync_bn else "") +
        (" -- REHEARSAL: the ranks share one GPU, collectives staged through gloo (host)" if ctx["staged"] else ""))
    return cfg


def bench_single(args, ctx, sharded):
    import torch
    rows_n = args.rows_notice or 1_000_000
    rows_c = args.rows_company or 1_000_000
    B = args.batch
    leg = Leg(args, ctx, B, rows_n, rows_c, sharded, negatives=args.negatives or "local", sync_bn=args.sync_bn)
    leg.run()
    is_c1 = B == 8192 and args.final_dim == 64 and args.hidden == "128,64" and args.zipf is None and rows_n == 1_000_000 and rows_c == 1_000_000 and args.score_dtype == "bf16"
    out = base_line(args, leg, 1, "weak", B)
    out["config"] = config_of(args, leg, 1, ctx, B, ("configs[1]: " if is_c1 else "variant of configs[1]: ") +
                              f"32+6 real keys, {rows_n}-row notice + {rows_c}-row company tables, batch {B}, E=32, towers [{args.hidden}], "
                              f"final {args.final_dim}, in-batch negatives, dropout 0.1")
    out["roofline"] = roofline_of(args, leg, 1)
    out["final_loss"] = leg.loss
    try:
        out["mfma"] = score_mfma_leg(args, ctx["dev"], B, args.final_dim)
    except Exception as e:                                # an instrumentation leg must not lose the line
        out["mfma"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_h2d and not sharded:
        try:
            out.update(h2d_leg(args, leg, ctx))
            if args.mlp_dtype == "bf16":                      # the same leg with bf16 dense features on the host side (never `value`)
                import torch as _torch
                out.update(h2d_leg(args, leg, ctx, dense_dtype=_torch.bfloat16, key="value_with_h2d_bf16_dense"))
        except Exception as e:
            out["value_with_h2d"] = None
            out["h2d_error"] = f"{type(e).__name__}: {e}"
    if not args.no_h2d and not sharded:
        try:
            out.update(device_store_leg(args, leg, ctx))
        except Exception as e:
            out["value_with_device_store"] = None
            out["device_store_error"] = f"{type(e).__name__}: {e}"
    if args.breakdown:
        from jodalrob_twotower_amd import ops
        t2 = ops.KernelTimer()
        ops.set_timer(t2)
        n = min(args.steps, 20)
        for i in range(n):
            leg.step(10_000 + i, eager=True)
        out["kernel_breakdown"] = {k: {"launches_per_step": v[0] / n, "mean_ms": round(v[1], 5)} for k, v in t2.summary().items()}
        ops.set_timer(None)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(leg, args.cpu_seconds)
    leg.close()
    if is_c1 and not sharded and not args.no_extra_legs and args.mode == "graph":
        for key, fn in (("roofline_hbm_resident", hbm_resident_lookup_leg), ("configs4", configs4_leg)):
            try:
                out[key] = fn(args, ctx)
            except Exception as e:                            # an extra leg must not lose the line
                out[key] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
    return out


def bench_multi(args, ctx):
    import torch
    world, rank = ctx["world"], ctx["rank"]
    rows_n = args.rows_notice or 100_000_000              # configs[2]: over ALL GPUs (row r lives on GPU r mod N)
    rows_c = args.rows_company or 10_000_000
    legs = (args.legs or "strong,weak,one_gpu").split(",")
    Bg = args.batch
    if Bg % world:
        raise SystemExit(f"--batch {Bg} must be a multiple of the {world} ranks")
    cname = "configs[3]" if args.zipf is not None else "configs[2]"
    desc = (f"32+6 real keys, {rows_n}-row notice + {rows_c}-row company tables sharded row-wise over {world} GPUs (row r on GPU r mod {world}), "
            f"E=32, towers [{args.hidden}], final {args.final_dim}, dropout 0.1")
    out, strong = None, None
    if "weak" in legs:
        # `value` at N > 1: every GPU keeps the N = 1 job's batch (8192 pairs PER GPU; per-GPU work fixed: "scaling": "weak"),
        # tables sharded row-wise -- the data-parallel job whose global batch grows with N
        leg = Leg(args, ctx, Bg, rows_n, rows_c, True, negatives=args.negatives or "local", sync_bn=args.sync_bn, label="weak")
        leg.run()
        out = base_line(args, leg, world, "weak", Bg * world)
        #  the top-level number at N > 1 is NOT the batch-8192 job of N = 1 split N ways: say so in the metric text
        # itself, so that nobody compares it with a global-batch figure under the same key; that job is `value_global_batch_8192`
        out["metric"] = (f"training pairs/sec, {Bg} pairs PER GPU (global batch {Bg * world}; weak scaling of the N = 1 job at batch {Bg}) "
                         "(embedding-lookup HBM GB/s in roofline)")
        out["config"] = config_of(args, leg, world, ctx, Bg * world, f"VARIANT of {cname} (weak-scaling leg): {desc}; {Bg} pairs PER GPU (global batch "
                                  f"{Bg * world}), per-rank in-batch negatives and BatchNorm statistics (data-parallel semantics); {cname} as SURVEY 8(d) "
                                  f"states it -- GLOBAL batch {Bg}, global negatives + SyncBN -- is `value_global_batch_{Bg}` / `strong_scaling`")
        out["roofline"] = roofline_of(args, leg, world)
        out["final_loss"] = leg.loss
        leg.close()
    if "strong" in legs:
        # the batch-8192 job split over N GPUs -- global in-batch negatives + SyncBN make it the single-process job at the
        # global batch (tests: test_two_processes_equal_single_process); latency-bound at 8192 / N pairs per GPU
        neg = args.negatives or "global"
        leg = Leg(args, ctx, Bg // world, rows_n, rows_c, True, negatives=neg, sync_bn=(neg == "global") or args.sync_bn, label="strong")
        leg.run()
        st = base_line(args, leg, world, "strong", Bg)
        cfg = config_of(args, leg, world, ctx, Bg, f"{cname}: {desc}; GLOBAL batch {Bg} ({Bg // world} pairs per GPU), global in-batch negatives + SyncBN")
        if out is None:
            out = st
            out["config"], out["roofline"], out["final_loss"] = cfg, roofline_of(args, leg, world), leg.loss
        else:
            strong = {k: st[k] for k in ("value", "unit", "ms_per_step", "device_ms_per_step_median", "host_enqueue_ms_per_step", "scaling")}
            strong["config"] = cfg
            strong["roofline"] = roofline_of(args, leg, world)
            out["strong_scaling"] = strong
            # SURVEY 8(d) C3 at top level too: pairs/s of the single-process batch-8192 job split over the N GPUs
            out[f"value_global_batch_{Bg}"] = strong["value"]
            out[f"ms_per_step_global_batch_{Bg}"] = strong["ms_per_step"]
            out[f"roofline_global_batch_{Bg}"] = strong["roofline"]
        leg.close()
    if "one_gpu" in legs:
        # like-for-like denominator: the unsharded single-GPU step on the SAME tables and batch, rank 0 only
        ref = None
        if rank == 0:
            try:
                solo = dict(ctx, world=1, comm=None, fence=lambda: torch.cuda.synchronize(), max_over_ranks=lambda x: x, staged=False)
                leg = Leg(args, solo, Bg, rows_n, rows_c, False, label="one_gpu")
                leg.run()
                ref = {"value": Bg * args.steps / leg.dt, "unit": "pairs/s", "ms_per_step": leg.dt / args.steps * 1e3,
                       "device_ms_per_step_median": leg.device_ms_median,
                       "config": f"the same {rows_n} + {rows_c}-row tables UNSHARDED on one GPU, batch {Bg}, measured by rank 0 in this run",
                       "roofline": roofline_of(args, leg, 1),           # the lookup over tables the Infinity Cache cannot hold
                       "launches_per_step": getattr(leg.gstep, "library_launches", None)}
                leg.close()
            except Exception as e:
                ref = {"error": f"{type(e).__name__}: {e}"}
        ctx["fence"]()
        if out is not None and ref is not None:
            out["one_gpu_same_tables"] = ref
            if "value" in ref:
                out["speedup_vs_one_gpu_same_tables"] = out["value"] / ref["value"]
                if "strong_scaling" in out:
                    out["strong_scaling"]["speedup_vs_one_gpu_same_tables"] = out["strong_scaling"]["value"] / ref["value"]
    return out


# ------------------------------------------------------------------------------------------------ instrumentation legs
def lookup_dispatch_overhead_us(task, pool, dev, profile, iters: int = 48):
    """What a dispatch costs on top of the kernel body (command-processor launch, end-of-kernel cache write-back): the same
    lookup launched right after the timed region, timed by HIP events around back-to-back replays (profile hook off) and by
    the in-kernel stamps; the difference of the means is added to the in-graph stamp time so that `mean_launch_us` is the
    quantity rocprofv3 --kernel-trace reports for this kernel."""
    import torch
    from jodalrob_twotower_amd import ops
    towers = [task.two_tower_model.notice_tower, task.two_tower_model.company_tower]
    ex = getattr(task, "exchange", None)
    store = None if ex is not None else towers[0].categorical_embedder.store
    place_buf = getattr(ex, "_place_buf", None) if ex is not None else None
    if ex is not None and place_buf is None:
        return None
    sides_per_batch = []
    rows_mode = getattr(getattr(task, "_bench_gstep", None), "_rows_sm", None) is not None
    for i, batch in enumerate(pool):
        sides = []
        for tw, side in zip(towers, ("notice", "company")):
            B = batch[side]["dense"].shape[0]
            x = torch.empty((B, tw.x_width), dtype=tw.x_dtype, device=dev)
            sides.append(tw.categorical_embedder.lookup_side(batch[side]["kjt"].values(), x[:, tw.tower_hidden_dims[0]:]))
        if ex is not None:
            # row-wise sharded tables: the launch that fills the tower inputs gathers out of the exchange's RECEIVE buffer by bucket
            # position (tt_embed_lookup_fwd[place]) -- this rank's shard holds 1/G of the rows and is never indexed by global rows
            M = sum(B * s.K for s in sides)
            g = torch.Generator(device=dev).manual_seed(4321 + i)
            rows = torch.randint(0, place_buf.shape[0] - 1, (M,), dtype=torch.int64, device=dev, generator=g)
        else:
            # the captured step's lookup reads precomputed fused rows: calibrate the same kernel
            rows = ops.embed_lookup(store.weight, sides, B, want_rows=True) if rows_mode else None
        sides_per_batch.append((sides, B, rows))

    def launch(sides, B, rows):
        if ex is not None:
            ex.backend.place_rows(place_buf, rows, sides, B)
        elif rows is not None:
            ops.embed_lookup_rows(store.weight, rows, sides, B)
        else:
            ops.embed_lookup(store.weight, sides, B, want_rows=True)
    K = len(sides_per_batch)
    # (A) K launches back to back inside a small captured graph (eagerly the host cannot issue a 9-us kernel fast enough),
    #     profile hook off, HIP events around the replay: dispatch-to-dispatch time per launch
    profile.close()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for sb in sides_per_batch:
            launch(*sb)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    REP = 4                                                     # 4 x K launches per replay: the replay's own launch cost amortises
    import torch.distributed as _d
    mode = "thread_local" if _d.is_initialized() else "global"       # a process group's watchdog polls events from its own thread
    with torch.cuda.graph(g, capture_error_mode=mode):
        keep = [launch(*sb) for _ in range(REP) for sb in sides_per_batch]
    ev_us = []
    for rep in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        ev_us.append(a.elapsed_time(b) * 1e3 / (REP * K))
    del g, keep
    # (B) the same K launches (eagerly) with the in-kernel stamps
    profile.reopen()
    profile.reset()
    for rep in range(3):
        for sb in sides_per_batch:
            launch(*sb)
    torch.cuda.synchronize()
    st_us = profile.durations_us()
    if not ev_us or not st_us:
        return None
    ev_us.sort()
    return max(0.0, ev_us[len(ev_us) // 2] - sum(st_us) / len(st_us))      # upper median of the replays: min() under-reports


def hbm_resident_lookup_leg(args, ctx, rows_n: int = 100_000_000, rows_c: int = 10_000_000, B: int = 8192):
    """The embedding lookup ALONE on tables the caches cannot hold (BASELINE configs[2] sizes on one GPU: 100 M + 10 M rows of
    128 B = 14 GB; tables only, no model, no optimiser state): north_star's ">= 60 % of the HBM roofline at batch 8192" where HBM
    really serves the rows.  Never `value`.  Three forms of the product path over a pool of 8 id batches:
      lookup_rows  tt_embed_lookup_rows_fwd (the captured step's lookup: fused rows precomputed by the hand-over launch) -- the leg's headline
      lookup       tt_embed_lookup_fwd (from int64 ids: eager steps, evaluation)
      handover     tt_batch_ingest_lookup (hand-over + lookup as ONE launch: + dense features and ids copied; optional form)
    each timed (a) back to back inside a small captured graph with HIP events around the replay: mean LAUNCH time incl. the
    kernel boundary (what rocprofv3 sums per kernel plus the gap), and with its workgroups' device-clock stamps: mean BODY;
    (b) `cold`: every launch behind a 256 MB streaming copy (what the other kernels of a step do to L2 / Infinity Cache), HIP
    events around the single launch."""
    import torch
    from jodalrob_twotower_amd import ops, synthetic
    dev = ctx["dev"]
    schema = synthetic.load_real_schema(ROOT / "jodalrob-twotower_amd" / "schema_real.json")
    kn, kc = schema["notice"]["categorical"], schema["company"]["categorical"]
    vn, vc = synthetic.scale_vocabs(schema["notice"]["vocab_sizes"], rows_n), synthetic.scale_vocabs(schema["company"]["vocab_sizes"], rows_c)
    E, h0, din_n, din_c = 32, 128, 256, 128
    R = sum(vn) + sum(vc)
    g = torch.Generator(device=dev).manual_seed(99)
    table = torch.empty((R, E), dtype=torch.float32, device=dev)
    for lo in range(0, R, 1 << 24):                              # filled in slices: no 14 GB temporaries
        table[lo:lo + (1 << 24)].normal_(generator=g)
    offs_n = torch.tensor([sum(vn[:i]) for i in range(len(vn))], dtype=torch.int64, device=dev)
    offs_c = torch.tensor([sum(vn) + sum(vc[:i]) for i in range(len(vc))], dtype=torch.int64, device=dev)
    voc_n, voc_c = torch.tensor(vn, dtype=torch.int64, device=dev), torch.tensor(vc, dtype=torch.int64, device=dev)
    xn = torch.zeros((B, h0 + len(kn) * E), dtype=torch.bfloat16, device=dev)
    xc = torch.zeros((B, h0 + len(kc) * E), dtype=torch.bfloat16, device=dev)
    pool = [synthetic.make_batch(B, vn, vc, kn, kc, din_n, din_c, dev, seed=500 + i) for i in range(8)]
    stat = {"dn": torch.empty((B, din_n), device=dev), "dc": torch.empty((B, din_c), device=dev),
            "in": torch.empty(B * len(kn), dtype=torch.int64, device=dev), "ic": torch.empty(B * len(kc), dtype=torch.int64, device=dev),
            "km": torch.empty(B * (len(kn) + len(kc)), dtype=torch.int32, device=dev)}
    K_tot = len(kn) + len(kc)
    lookup_bytes = B * K_tot * (E * 4 + 8 + E * 2)
    rows_bytes = B * K_tot * (E * 4 + 4 + E * 2)
    handover_bytes = lookup_bytes + B * (2 * 4 * (din_n + din_c) + K_tot * (8 + 8 + 4))
    tiles = ops.ingest_lookup_tiles(B, [len(kn), len(kc)])

    def sides(b):
        return [ops.LookupSide(b["notice"]["kjt"].values(), offs_n, voc_n, xn[:, h0:], len(kn)),
                ops.LookupSide(b["company"]["kjt"].values(), offs_c, voc_c, xc[:, h0:], len(kc))]

    def run_lookup(b):
        ops.embed_lookup(table, sides(b), B, want_rows=False)

    rows_of = {id(b): ops.embed_lookup(table, sides(b), B, want_rows=True) for b in pool}      # slot-order fused rows (what the hand-over leaves)

    def run_lookup_rows(b):
        ops.embed_lookup_rows(table, rows_of[id(b)], sides(b), B)

    def run_handover(b):
        ops.batch_ingest([(stat["dn"], b["notice"]["dense"]), (stat["in"], b["notice"]["kjt"].values()), (stat["dc"], b["company"]["dense"]),
                          (stat["ic"], b["company"]["kjt"].values())], sides(b), B, stat["km"], table=table)

    flush_src = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    flush_dst = torch.empty_like(flush_src)
    prof = ops.LookupProfile(dev, n_slots=64)
    import torch.distributed as _d
    mode = "thread_local" if _d.is_initialized() else "global"
    out = {"tables": f"{rows_n} + {rows_c} rows x {E} f32 = {R * E * 4 / 1e9:.1f} GB, uniform ids over the real 32 + 6 key vocabularies scaled to these sizes",
           "batch": B}
    try:
        for name, fn, nbytes, last in (("lookup_rows", run_lookup_rows, rows_bytes, None), ("lookup", run_lookup, lookup_bytes, None),
                                       ("handover", run_handover, handover_bytes, None)):
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for b in pool:
                    fn(b)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            REP = 4
            # (a) launch time: the kernels as the product launches them -- the stamp hook OFF (a stamped launch ends with a wait for its own
            #     stores and two more stores per workgroup) --, HIP events around the replay; (b) body: a second capture with the hook on
            prof.close()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode=mode):
                for _ in range(REP):
                    for b in pool:
                        fn(b)
            launch_us = []
            for rep in range(12):
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                gr.replay()
                e.record()
                torch.cuda.synchronize()
                if rep >= 2:                                     # (the first replays of a graph are slower: warm-up)
                    launch_us.append(a.elapsed_time(e) * 1e3 / (REP * len(pool)))
            del gr
            prof.reopen()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode=mode):
                for b in pool:
                    fn(b)
            prof.reset()
            for rep in range(4):
                gr.replay()
            body = prof.durations_us()
            phase = prof.durations_us(0, tiles) if name == "handover" else body
            del gr
            cold = []
            for i in range(24):
                flush_dst.copy_(flush_src)
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn(pool[i % len(pool)])
                e.record()
                cold.append((a, e))
            torch.cuda.synchronize()
            cold_us = sorted(a.elapsed_time(e) * 1e3 for a, e in cold[4:])
            launch_us.sort()
            ml = launch_us[len(launch_us) // 2]
            rec = {"entry": {"lookup": "tt_embed_lookup_fwd", "lookup_rows": "tt_embed_lookup_rows_fwd", "handover": "tt_batch_ingest_lookup"}[name],
                   "algorithmic_bytes_per_launch": nbytes,
                   "mean_launch_us": ml, "mean_body_us": (sum(body) / len(body)) if body else None,
                   "achieved": nbytes / (ml * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": nbytes / (ml * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                   "cold_cache_launch_us": cold_us[len(cold_us) // 2], "cold_cache_frac": nbytes / (cold_us[len(cold_us) // 2] * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                   "launches_timed": REP * len(pool) * 10}
            if name == "handover" and phase:
                pu = sum(phase) / len(phase)
                rec["lookup_phase"] = {"mean_us": pu, "algorithmic_bytes": lookup_bytes, "frac": lookup_bytes / (pu * 1e-6) / 1e9 / HBM_PEAK_GBPS}
            out[name] = rec
        out["frac"] = out["lookup_rows"]["frac"]                  # the headline of this leg: the captured step's lookup launch
        out["mean_launch_us"], out["mean_body_us"] = out["lookup_rows"]["mean_launch_us"], out["lookup_rows"]["mean_body_us"]
        out["how"] = ("mean_launch_us: 32 launches over 8 id batches back to back in a captured graph, stamp hook off, HIP events around the replay, "
                      "median of 10 replays behind 2 (kernel + boundary); mean_body_us: min start .. max end of the workgroups' s_memrealtime stamps in a "
                      "second capture of the same launches with the hook on; "
                      "cold_cache_*: each launch behind a 256 MB streaming copy, HIP events around the single launch")
    finally:
        prof.close()
        del table
        torch.cuda.empty_cache()
    return out


def configs4_leg(args, ctx):
    """BASELINE configs[4] in the default run (never `value`): final_embedding_dim 256, batch 65536, fp8 (e4m3) MFMA score matrix,
    fused sparse Adam on the embedding rows; 1 M + 1 M-row tables; a few replayed steps + the score kernels' MFMA figure."""
    import copy
    a4 = copy.copy(args)
    # (40 steps behind 10: 0.35 s of GPU time; the five steps behind two of rounds 3-4 read a cold process, like configs[1]'s 100-step region)
    a4.batch, a4.final_dim, a4.score_dtype, a4.steps, a4.warmup, a4.pool, a4.no_lookup_profile = 65536, 256, "fp8", 40, 10, 2, True
    a4.hidden, a4.optimizer, a4.mlp_dtype, a4.zipf = "128,64", "fused_sparse", "bf16", None
    leg = Leg(a4, ctx, a4.batch, 1_000_000, 1_000_000, False)
    try:
        leg.run()
        out = {"value": a4.batch * a4.steps / leg.dt, "unit": "pairs/s", "ms_per_step": leg.dt / a4.steps * 1e3,
               "device_ms_per_step_median": leg.device_ms_median, "steps": a4.steps, "warmup": a4.warmup, "final_loss": leg.loss,
               "config": {"workload": "configs[4]: 32+6 real keys, 1000000 + 1000000 table rows, batch 65536, E=32, towers [128,64], final 256, "
                                      "fp8 (e4m3) score operands, fused sparse Adam", "fp8_grad": a4.fp8_grad,
                          "launches_per_step": getattr(leg.gstep, "library_launches", None)}}
        try:
            out["mfma"] = score_mfma_leg(a4, ctx["dev"], a4.batch, a4.final_dim, reps=2)
        except Exception as e:
            out["mfma"] = {"error": f"{type(e).__name__}: {e}"}
    finally:
        leg.close()
    return out


def score_mfma_leg(args, dev, B, D, reps: int = 8):
    """The score GEMMs (the MFMA users north_star names) on their own: forward + loss finish + backward of the product path's
    autograd node on random unit rows, `reps` back-to-back iterations per graph replay, HIP events around the replays.
    FLOPs per SURVEY 8(d): forward 2 B^2 D, backward 4 B^2 D (recomputed score tiles are overhead, not credited)."""
    import torch
    from jodalrob_twotower_amd.two_tower_train_task import _ScoreCEFn
    if args.score_dtype == "fp32":
        return None
    g = torch.Generator(device=dev).manual_seed(7)
    n = torch.nn.functional.normalize(torch.randn((B, D), generator=g, device=dev), dim=1).requires_grad_(True)
    c = torch.nn.functional.normalize(torch.randn((B, D), generator=g, device=dev), dim=1).requires_grad_(True)
    ones = torch.ones((), device=dev)

    def body():
        n.grad = c.grad = None
        loss, _, _ = _ScoreCEFn.apply(n, c, 1.0, args.score_dtype, False, False)
        loss.backward(ones)
        return loss

    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    import torch.distributed as _d
    with torch.cuda.graph(gr, capture_error_mode="thread_local" if _d.is_initialized() else "global"):
        for _ in range(reps):
            body()
    times = []
    for _ in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        gr.replay()
        b.record()
        torch.cuda.synchronize()
        times.append(a.elapsed_time(b) * 1e3 / reps)
    times.sort()
    us = times[len(times) // 2]
    flops = 6.0 * B * B * D
    peak = MFMA_FP8_PEAK_TFLOPS if args.score_dtype == "fp8" else MFMA_BF16_PEAK_TFLOPS
    ach = flops / (us * 1e-6) / 1e12
    del gr
    return {"kernel": f"score forward + backward ({args.score_dtype} operands; incl. the operand pack and loss-finish launches)", "bound": "mfma",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "us_per_iteration": us,
            "algorithmic_flops_per_iteration": flops, "flop_count": "2 B^2 D forward + 4 B^2 D backward (SURVEY 8d); recomputation not credited"}


def h2d_leg(args, leg, ctx, dense_dtype=None, key="value_with_h2d"):
    """The same step fed from PINNED HOST batches (what the reference pays every step: scripts/train.py:261-273, 315-316):
    ids + dense features cross PCIe into one of two device staging sets on a side stream while the previous step computes;
    the step waits for its batch's copy event.  Returns value_with_h2d = pairs/s of K such steps."""
    import torch
    dev, B = ctx["dev"], leg.B
    # dense_dtype=torch.bfloat16: the loader keeps the dense features in bf16 (what the projection GEMM of the bf16 MLP rounds them
    # to anyway: identical results) -- half of the 12.6 MB of dense features per step on the PCIe link
    cast = (lambda t: t.to(dense_dtype)) if dense_dtype is not None else (lambda t: t)
    host = [{s: {"dense": cast(b[s]["dense"].cpu()).pin_memory(), "ids": b[s]["kjt"].values().cpu().pin_memory()} for s in ("notice", "company")}
            for b in leg.pool]
    from jodalrob_twotower_amd.kjt import KeyedJaggedTensor
    stage = [{s: {"dense": torch.empty_like(cast(leg.pool[0][s]["dense"])),
                  "kjt": KeyedJaggedTensor(leg.pool[0][s]["kjt"].keys(), torch.empty_like(leg.pool[0][s]["kjt"].values()))}
              for s in ("notice", "company")} for _ in range(2)]
    copy_stream = torch.cuda.Stream(device=dev)
    ready = [torch.cuda.Event() for _ in range(2)]
    freed = [torch.cuda.Event() for _ in range(2)]
    main = torch.cuda.current_stream(dev)

    def upload(i):
        slot = i % 2
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(freed[slot])                   # the step that last read this staging set has run
            h = host[i % len(host)]
            for s in ("notice", "company"):
                stage[slot][s]["dense"].copy_(h[s]["dense"], non_blocking=True)
                stage[slot][s]["kjt"].values().copy_(h[s]["ids"], non_blocking=True)
            ready[slot].record(copy_stream)

    for e in freed:
        e.record(main)
    K = args.steps
    upload(0)
    for i in range(min(10, K)):                                   # settle
        upload(i + 1)
        main.wait_event(ready[i % 2])
        leg.step(i, batch=stage[i % 2])
        freed[i % 2].record(main)
    torch.cuda.synchronize()
    base = min(10, K)
    t0 = time.perf_counter()
    for j in range(K):
        i = base + j
        upload(i + 1)
        main.wait_event(ready[i % 2])
        leg.step(i, batch=stage[i % 2])
        freed[i % 2].record(main)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = sum(h[s]["dense"].numel() * h[s]["dense"].element_size() + h[s]["ids"].numel() * 8 for h in host[:1] for s in ("notice", "company"))
    return {key: B * K / dt, "ms_per_step" + key[5:]: dt / K * 1e3,
            key[11:]: {"bytes_per_step": nbytes, "how": "pinned host batches -> two device staging sets on a copy stream, the step waits for its batch's copy event "
                                                     "(the copy of step i+1 overlaps the compute of step i)"}}


def device_store_leg(args, leg, ctx, entities: int = 262_144, key: str = "value_with_device_store"):
    """The captured step fed from DEVICE-RESIDENT feature stores (never `value`): per step ONE hand-over launch gathers the batch's
    pairs -> entity rows -> dense features + ids + key-major fused rows into the graph's static buffers
    (GraphedTrainStep.step_from_store / tt_batch_ingest_store), then the replay -- the loop scripts/train.py --fast runs.  Nothing
    per step crosses PCIe.  Stores: `entities` notice and company rows (dense f32 [N, 256] / [N, 128], ids int64 [N, 32] / [N, 6],
    uniform over the leg's vocabularies), 64 batches of shuffled pairs."""
    import torch
    from jodalrob_twotower_amd.data_loader import DeviceFeatureStore
    dev, B = ctx["dev"], leg.B
    if leg.gstep is None or not hasattr(leg.gstep, "step_from_store"):
        return {key: None}
    g = torch.Generator(device=dev)
    g.manual_seed(4321)

    def make(n, din, vocab, keys):
        cat = torch.stack([torch.randint(0, v, (n,), generator=g, device=dev, dtype=torch.int64) for v in vocab], dim=1)
        return DeviceFeatureStore({"dense_projected": torch.randn((n, din), generator=g, device=dev), "categorical": cat}, keys, dev)
    ns, cs = make(entities, leg.din_n, leg.vocab_n, leg.keys_n), make(entities, leg.din_c, leg.vocab_c, leg.keys_c)
    P = 64 * B
    pairs = torch.randint(0, entities, (P, 2), generator=g, device=dev, dtype=torch.int64)
    order = torch.randperm(P, generator=g, device=dev)
    K = args.steps

    def one(j):
        res = leg.gstep.step_from_store(ns, cs, pairs, order, (j % 64) * B)
        leg.sched.step()
        return res
    U = int(getattr(leg.gstep, "unroll", 1))

    def many(j0, count):                                          # (an unrolled captured step: U batches per graph launch)
        res, i = None, 0
        while i < count:
            if U > 1 and count - i >= U:
                res = leg.gstep.steps_from_store(ns, cs, pairs, order, [((j0 + i + u) % 64) * B for u in range(U)], after_each=leg.sched.step)[-1]
                i += U
            else:
                res = one(j0 + i)
                i += 1
        return res
    many(0, min(10, K))
    if U > 1 and K % U:
        leg.gstep._single()
    ctx["fence"]()
    t0 = time.perf_counter()
    res = many(10, K)
    ctx["fence"]()
    dt = ctx["max_over_ranks"](time.perf_counter() - t0)
    loss = float(res["loss"].detach())
    del ns, cs, pairs, order
    return {key: B * K / dt, "ms_per_step" + key[5:]: dt / K * 1e3,
            key[11:]: {"entities_per_store": entities, "pairs": P, "final_loss": loss,
                       "how": "GraphedTrainStep.step_from_store: tt_batch_ingest_store (pair -> entity row -> dense features, ids, key-major "
                              "fused rows, step scalars; one launch) + graph replay; stores and pair list resident in HBM"}}


def _physical_cores() -> int:
    try:
        ids = set()
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    ids.add((phys, core))
                phys = core = None
        n = len(ids) or (os.cpu_count() or 1)
    except Exception:
        n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(leg, seconds: float):
    """oracle/oracle_torch.py -- the vectorised torch-CPU restatement of the reference step (forward + autograd backward +
    torch.optim.Adam over every parameter, dense table gradients: reference semantics) -- timed on the host's physical cores
    for about `seconds` of CPU work on steps of the same batch shape."""
    import torch
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle_torch as OT
    cores = _physical_cores()
    prev = torch.get_num_threads()
    torch.set_num_threads(cores)
    try:
        B = leg.B
        state = OT.make_state({k: v.detach().cpu().numpy() for k, v in leg.task.state_dict().items()})
        b0 = leg.pool[0]
        batch = {"notice_ids": b0["notice"]["kjt"].values().cpu().view(B, -1), "company_ids": b0["company"]["kjt"].values().cpu().view(B, -1),
                 "notice_dense": b0["notice"]["dense"].cpu(), "company_dense": b0["company"]["dense"].cpu()}
        params = OT.parameters(state)
        opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5)

        def one():
            opt.zero_grad(set_to_none=True)
            loss, _ = OT.task_loss(state, batch, leg.keys_n, leg.keys_c, leg.vocab_n, leg.vocab_c, 1.0, True)
            loss.backward()
            opt.step()

        one()                                                     # first call: allocator, thread pool
        t0 = time.perf_counter()
        n = 0
        while True:
            one()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 200:
                break
    finally:
        torch.set_num_threads(prev)
    return {"value": B * n / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} steps of batch {B} in {dt:.1f} s (forward + autograd backward + torch.optim.Adam over all "
                      f"{sum(p.numel() for p in params)} parameters, dense table gradients), oracle/oracle_torch.py on {cores} threads "
                      f"(physical cores available to this process; host reports {os.cpu_count()} logical). Vectorised id unpack: an "
                      f"optimistic stand-in for the reference, whose per-id Python loop alone takes ~1.2 s per step at this batch (SURVEY section 6)"}


def main():
    args = parse()
    if os.environ.get("TT_BENCH_LAUNCH_ONLY"):            # plumbing test of the self-launch (no GPU, no package import)
        if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
            raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
        info = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        print(json.dumps({"launch_only": True, "gpus": args.gpus, **info}), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    run(args)


if __name__ == "__main__":
    main()
