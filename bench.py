#!/usr/bin/env python3
"""bench.py -- training pairs/sec of the two-tower step on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Step = TwoTowerTrainTask forward (return_metrics=True) + loss.backward() + optimiser step + LR schedule
on one synthetic batch whose ids and dense features are already resident in HBM.  Workload at N=1 is
BASELINE.json configs[1]: real 32+6 key schema, per-key vocabularies scaled to 1 M rows per tower,
E=32, towers [128,64], final 64, batch 8192, in-batch negatives.

Prints ONE JSON line (rank 0).  `roofline` is the embedding-lookup kernel (the kernel BASELINE.json's
metric names): algorithmic bytes per launch / mean launch duration measured with HIP events on the
launch stream inside the timed region.  `cpu_baseline` times the numpy oracle (checker code, used here
only as the reported baseline) on a bounded sample of the same workload on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3      # v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8192, help="pairs per GPU per step")
    ap.add_argument("--rows-notice", type=int, default=1_000_000, help="table rows per GPU, notice tower")
    ap.add_argument("--rows-company", type=int, default=1_000_000, help="table rows per GPU, company tower")
    ap.add_argument("--zipf", type=float, default=None, help="Zipf alpha for ids (default uniform)")
    ap.add_argument("--final-dim", type=int, default=64, help="final_embedding_dim (BASELINE configs[4] uses 256 at --batch 65536)")
    ap.add_argument("--optimizer", choices=["fused_sparse", "fused_dense", "torch_adam"], default="fused_sparse")
    ap.add_argument("--score-dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--mlp-dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--pool", type=int, default=8, help="distinct pre-generated batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8, help="oracle steps timed for cpu_baseline (~1.4 s each on the box)")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="graph: whole step replayed from one captured HIP graph; eager: launched from Python")
    ap.add_argument("--force-dist", action="store_true", help="use the sharded-table path even on one GPU")
    ap.add_argument("--negatives", choices=["local", "global"], default="local",
                    help="multi-GPU: in-batch negatives of the rank's own batch (data-parallel default) or of the GLOBAL batch")
    ap.add_argument("--sync-bn", action="store_true", help="multi-GPU: BatchNorm statistics over all ranks' rows")
    ap.add_argument("--breakdown", action="store_true", help="extra instrumented pass: per-kernel HIP-event times")
    return ap.parse_args()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run); got {world}")
    import jodalrob_twotower_amd as tt
    from jodalrob_twotower_amd import ops, synthetic
    from jodalrob_twotower_amd.optim import FusedAdam

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if os.environ.get("TT_BENCH_INIT_PG"):          # fault hunting: a process group exists, the task stays single-GPU
        import torch.distributed as _d
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29545")
        _d.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        _t = torch.ones(4, device=dev); _d.all_reduce(_t); torch.cuda.synchronize()
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29544")
        # RCCL and gloo print banners to stdout when their communicators come up: park fd 1 on stderr until the JSON line
        sys.stdout.flush()
        _saved_stdout = os.dup(1)
        os.dup2(2, 1)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # barriers / the max-over-ranks of the timing go through a gloo group: an eager RCCL collective issued between
        # replays of a graph that CONTAINS RCCL collectives faulted the GPU here (the replayed kernels read work
        # descriptors that the eager launch had recycled)
        cpu_group = dist.new_group(backend="gloo")

    schema = synthetic.load_real_schema(ROOT / "jodalrob-twotower_amd" / "schema_real.json")
    keys_n, keys_c = schema["notice"]["categorical"], schema["company"]["categorical"]
    # weak scaling: every GPU brings its own 1 M + 1 M rows and its own 8192 pairs
    vocab_n = synthetic.scale_vocabs(schema["notice"]["vocab_sizes"], args.rows_notice * world)
    vocab_c = synthetic.scale_vocabs(schema["company"]["vocab_sizes"], args.rows_company * world)
    E, hidden, D, din_n, din_c = 32, [128, 64], args.final_dim, 256, 128
    B = args.batch
    tmp = tempfile.mkdtemp(prefix="tt_bench_")
    meta = synthetic.write_metadata(Path(tmp) / "metadata.csv", {"notice": dict(zip(keys_n, vocab_n)),
                                                                 "company": dict(zip(keys_c, vocab_c))})
    torch.manual_seed(1234)
    grad_mode = "sparse" if args.optimizer == "fused_sparse" else "dense"
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        if dist is not None:
            from jodalrob_twotower_amd.distributed import create_distributed_train_task
            task = create_distributed_train_task(keys_n, keys_c, metadata_path=str(meta), categorical_embedding_dim=E,
                                                 notice_dense_input_dim=din_n, company_dense_input_dim=din_c,
                                                 tower_hidden_dims=hidden, final_embedding_dim=D, dropout_rate=0.1,
                                                 temperature=1.0, device=dev, embedding_grad=grad_mode, score_dtype=args.score_dtype,
                                                 mlp_dtype=args.mlp_dtype,
                                                 # dedup-first, fixed-capacity all-to-alls: the whole step incl. RCCL is one graph replay
                                                 exchange="padded" if grad_mode == "sparse" else "exact",
                                                 negatives=args.negatives, sync_bn=args.sync_bn)
        else:
            task = tt.create_two_tower_train_task(keys_n, keys_c, metadata_path=str(meta), categorical_embedding_dim=E,
                                                  notice_dense_input_dim=din_n, company_dense_input_dim=din_c,
                                                  tower_hidden_dims=hidden, final_embedding_dim=D, dropout_rate=0.1,
                                                  temperature=1.0, device=dev, embedding_grad=grad_mode,
                                                  score_dtype=args.score_dtype, mlp_dtype=args.mlp_dtype)
    task.train()
    task._pair_check_done = True            # skip the first-call diagnostic printout (host sync)
    if args.optimizer == "torch_adam":
        opt = torch.optim.Adam(task.parameters(), lr=1e-3, weight_decay=1e-5)
    else:
        opt = FusedAdam.for_task(task, lr=1e-3, weight_decay=1e-5)
    total_steps = args.steps + args.warmup
    warm = max(1, int(total_steps * 0.05))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: s / warm if s < warm else 1.0, last_epoch=-1)
    pool = [synthetic.make_batch(B, vocab_n, vocab_c, keys_n, keys_c, din_n, din_c, dev, seed=1234 + 7919 * (rank * args.pool + i),
                                 zipf_alpha=args.zipf) for i in range(args.pool)]

    # Sharded step: the fixed-capacity exchange makes it capturable (RCCL all-to-alls and the dense all-reduce inside the
    # graph); TT_DIST_EAGER=1 launches it eagerly instead.
    use_graph = (args.mode == "graph" and args.optimizer == "fused_sparse" and not os.environ.get("TT_DIST_EAGER")) \
        if dist is not None else (args.mode == "graph" and args.optimizer != "torch_adam")
    gstep = None
    profile = ops.LookupProfile(dev) if (use_graph and not os.environ.get("TT_BENCH_NO_PROFILE")) else None   # device-clock stamps: work inside a graph
    if dist is not None and getattr(task, "exchange", None) is not None and hasattr(task.exchange, "reset_capacity"):
        # fixed-capacity exchange: size the buckets for the largest need over the whole batch pool (one forward per
        # batch, outside the timed region), not just for the batch the capture happens to see
        caps = []
        with torch.no_grad():
            for b in pool:
                task.exchange.reset_capacity()
                task(b, return_metrics=False)
                caps.append(task.exchange.C)
        task.exchange.C = max(caps)
        torch.cuda.synchronize()
    if use_graph:
        from jodalrob_twotower_amd.graph import GraphedTrainStep
        try:
            gstep = GraphedTrainStep(task, opt, pool[0], return_metrics=True, warmup=3)
        except Exception as e:                      # a capture that fails on some RCCL / world size must not lose the run
            if dist is None:
                raise
            print(f"[bench] rank {rank}: graph capture of the sharded step failed ({type(e).__name__}: {e}); running eagerly",
                  file=sys.stderr, flush=True)
            if profile is not None:
                profile.close()
            gstep, profile = None, None
            torch.cuda.synchronize()
        if os.environ.get("TT_BENCH_TRACE"):
            torch.cuda.synchronize(); print("[bench] captured", file=sys.stderr, flush=True)

    same_batch = bool(os.environ.get("TT_BENCH_SAME_BATCH"))      # fault hunting: every step on pool[0]

    def step(i, eager=False):
        if same_batch:
            i = 0
        if gstep is not None and not eager:
            res = gstep.step(pool[i % args.pool])
            sched.step()
            return res
        opt.zero_grad()
        res = task(pool[i % args.pool], return_metrics=True)
        res["loss"].backward()
        opt.step()
        sched.step()
        return res

    def fence():
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier(group=cpu_group)
        torch.cuda.synchronize()

    for i in range(args.warmup):
        res = step(i)
        if os.environ.get("TT_BENCH_TRACE"):
            torch.cuda.synchronize(); print(f"[bench] warm-up replay {i} ok", file=sys.stderr, flush=True)
    fence()
    # In eager mode the lookup launches are timed with HIP events inside the timed region.  A graph replay
    # has no per-kernel host call to bracket, so in graph mode the same launches are timed in an eager pass
    # of the same steps right after the timed region (same kernel, same batches, same stream).
    lookup_name = "tt_embed_lookup_fwd" if dist is None else "tt_embed_lookup_fwd[place]"   # sharded: the launch that fills the tower inputs
    timer = ops.KernelTimer(names=[lookup_name])
    if gstep is None:
        ops.set_timer(timer)
    else:
        if profile is not None:
            profile.reset()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(args.warmup + i)
    t_enqueue = time.perf_counter() - t0                     # host time to issue the K steps (no sync inside)
    fence()
    dt = time.perf_counter() - t0
    ops.set_timer(None)
    lookup_us = profile.durations_us() if profile is not None else []
    dispatch_us = None
    if profile is not None:
        # (sharded runs: the same procedure on the local shard -- the overhead is a property of the dispatch, not of the rows)
        dispatch_us = lookup_dispatch_overhead_us(task, pool, dev, profile)
    if profile is not None:
        profile.close()
    if dist is not None and gstep is not None:
        # sharded step: the one stamped lookup launch per replay is the PLACE launch (pooled rows -> tower inputs; the same
        # bytes as the single-GPU lookup); the owner-side gathers run in tt_gather_rows
        if task.exchange.overflowed():
            raise RuntimeError("bench: a fixed-capacity bucket overflowed during the timed region -- result invalid")
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=cpu_group)
        dt = float(tmax.item())
    loss_val = float(res["loss"])
    ksum = timer.summary()
    n_launch, lookup_ms = ksum.get(lookup_name, (0, float("nan")))
    body_us = None
    if lookup_us:
        body_us = sum(lookup_us) / len(lookup_us)
        n_launch, lookup_ms = len(lookup_us), (body_us + (dispatch_us or 0.0)) * 1e-3

    breakdown = None
    if args.breakdown and rank == 0:
        t2 = ops.KernelTimer()
        ops.set_timer(t2)
        for i in range(min(args.steps, 20)):
            step(total_steps + i, eager=True)
        breakdown = {k: {"launches_per_step": v[0] / min(args.steps, 20), "mean_ms": round(v[1], 5)} for k, v in t2.summary().items()}
        ops.set_timer(None)

    if rank != 0:
        if dist is not None:
            gstep = None
            import gc
            gc.collect()
            torch.cuda.synchronize()
            dist.destroy_process_group()
        return
    K_tot = len(keys_n) + len(keys_c)
    x_bf16 = args.mlp_dtype == "bf16" and os.environ.get("TT_TOWER_IO_DTYPE", "x") in ("x", "both")
    s_out = 2 if x_bf16 else 4                                # the lookup writes straight into the tower input x
    bytes_per_pair = K_tot * (E * 4 + 8 + E * s_out)         # table row + i64 id + output row (10,032 / 7,600 B)
    if dist is not None and x_bf16 and getattr(getattr(task, "exchange", None), "wire_bf16", False):
        # sharded step: the stamped launch PLACES the exchanged rows, which arrive as bf16 (the f32 table rows are read by
        # tt_gather_rows on their owners): bf16 row in + i64 index + bf16 row out
        bytes_per_pair = K_tot * (E * 2 + 8 + E * 2)
    algo_bytes = B * bytes_per_pair                          # one launch = one batch on this GPU
    achieved = algo_bytes / (lookup_ms * 1e-3) / 1e9 if lookup_ms == lookup_ms and lookup_ms > 0 else None
    traffic = None
    pmc = ROOT / "profiles" / ("lookup_pmc_bf16out.json" if x_bf16 else "lookup_pmc.json")
    if pmc.exists() and dist is None:                        # the PMC passes were taken on the single-GPU lookup launch
        try:
            traffic = json.loads(pmc.read_text()).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "training pairs/sec at batch 8192 (embedding-lookup HBM GB/s in roofline)",
        "value": B * world * args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "host_enqueue_ms_per_step": t_enqueue / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("bf16" if (args.score_dtype == "bf16" and args.mlp_dtype == "bf16") else
                  f"score {args.score_dtype} / mlp {args.mlp_dtype}") + " MFMA operands, f32 accumulate; f32 tables, master weights and activations" +
                 (" (the tower input x, which the GEMMs round to bf16 anyway, is stored bf16)" if x_bf16 else ""),
        "data": "synthetic",
        "config": {"workload": ("configs[1]: " if (B == 8192 and D == 64 and args.zipf is None and args.rows_notice == 1_000_000 and args.rows_company == 1_000_000)
                                else "variant of configs[1]: ") +
                               f"32+6 real keys, {sum(vocab_n)}-row notice + {sum(vocab_c)}-row company tables per GPU, batch {B} per GPU, "
                               f"E=32, towers [128,64], final {D}, in-batch negatives, dropout 0.1",
                   "batch_per_gpu": B, "global_batch": B * world, "rows_notice": sum(vocab_n), "rows_company": sum(vocab_c),
                   "ids": "uniform" if args.zipf is None else f"zipf({args.zipf})", "optimizer": args.optimizer,
                   "score_dtype": args.score_dtype, "mlp_dtype": args.mlp_dtype, "launch": "hip graph replay" if gstep is not None else "eager",
                   **({} if dist is None or not hasattr(task.exchange, "C") else
                      {"exchange_capacity_rows_per_peer": task.exchange.C,
                       "exchange_bytes_per_rank_fwd": world * task.exchange.C * E * (2 if (x_bf16 and task.exchange.wire_bf16) else 4),
                       "exchange_bytes_per_rank_bwd": world * task.exchange.C * E * (2 if task.exchange.grad_wire_bf16 else 4)}),
                   "parallelism": ("single GPU" if dist is None else f"row-wise sharded tables x{world} (dedup-first fixed-capacity all-to-all" +
                                   (", RCCL inside the graph" if gstep is not None else "") + ") + data parallel towers" +
                                   (f", {args.negatives} in-batch negatives" + (", SyncBN" if args.sync_bn else "")))},
        "roofline": {"kernel": "lookup_kernel (tt_embed_lookup_fwd)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                     "algorithmic_bytes_per_launch": algo_bytes, "launches_timed": n_launch, "mean_launch_us": lookup_ms * 1e3,
                     "mean_body_us": body_us, "dispatch_overhead_us": dispatch_us,
                     "timed_in": "timed region, HIP events on the launch stream" if gstep is None else
                                 "timed region (graph replay), every launch: mean_launch_us = mean_body_us + dispatch_overhead_us. mean_body_us: device-clock (s_memrealtime, 100 MHz) stamps, min start .. "
                                 "max end over the kernel's workgroups, ring of per-launch slots read after the region "
                                 "(HIP events cannot bracket one kernel inside a replayed graph)"},
        "final_loss": loss_val,
    }
    if breakdown is not None:
        out["kernel_breakdown"] = breakdown
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(task, pool[0], keys_n, keys_c, vocab_n, vocab_c, B, args.cpu_steps)
    if dist is not None:
        sys.stdout.flush()
        os.dup2(_saved_stdout, 1)
    print(json.dumps(out), flush=True)
    if dist is not None:
        os.dup2(2, 1)
        gstep = None                    # the captured graph holds the communicator: drop it before the process group
        import gc
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()


def lookup_dispatch_overhead_us(task, pool, dev, profile, iters: int = 48):
    """What a dispatch costs on top of the kernel body (command-processor launch, end-of-kernel cache write-back): the same
    lookup launched eagerly right after the timed region, alternately timed by HIP events on the launch stream (profile hook
    off) and by the in-kernel stamps; the difference of the means is added to the in-graph stamp time so that
    `mean_launch_us` is the quantity rocprofv3 --kernel-trace reports for this kernel."""
    from jodalrob_twotower_amd import ops
    towers = [task.two_tower_model.notice_tower, task.two_tower_model.company_tower]
    store = task.sharded_store if hasattr(task, "sharded_store") else towers[0].categorical_embedder.store
    sides_per_batch = []
    for batch in pool:
        sides = []
        for tw, side in zip(towers, ("notice", "company")):
            B = batch[side]["dense"].shape[0]
            x = torch.empty((B, tw.x_width), dtype=tw.x_dtype, device=dev)
            sides.append(tw.categorical_embedder.lookup_side(batch[side]["kjt"].values(), x[:, tw.tower_hidden_dims[0]:]))
        sides_per_batch.append((sides, B))
    K = len(sides_per_batch)
    # (A) K launches back to back inside a small captured graph (eagerly the host cannot issue a 9-us kernel fast enough),
    #     profile hook off, HIP events around the replay: dispatch-to-dispatch time per launch
    profile.close()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for sides, B in sides_per_batch:
            ops.embed_lookup(store.weight, sides, B, want_rows=True)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    REP = 4                                                     # 4 x K launches per replay: the replay's own launch cost amortises
    import torch.distributed as _d
    mode = "thread_local" if _d.is_initialized() else "global"       # a process group's watchdog polls events from its own thread
    with torch.cuda.graph(g, capture_error_mode=mode):
        keep = [ops.embed_lookup(store.weight, sides, B, want_rows=True) for _ in range(REP) for sides, B in sides_per_batch]
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
        for sides, B in sides_per_batch:
            ops.embed_lookup(store.weight, sides, B, want_rows=True)
    torch.cuda.synchronize()
    st_us = profile.durations_us()
    if not ev_us or not st_us:
        return None
    if os.environ.get("TT_BENCH_TRACE"):
        print(f"[bench] dispatch calibration: replays {['%.2f' % v for v in ev_us]} us/launch, stamps {sum(st_us) / len(st_us):.2f} us", file=sys.stderr)
    ev_us.sort()
    return max(0.0, ev_us[len(ev_us) // 2] - sum(st_us) / len(st_us))      # upper median of the replays: min() under-reports


def cpu_baseline(task, batch, keys_n, keys_c, vocab_n, vocab_c, B, n_steps):
    """Oracle (numpy restatement of the reference step: forward + backward + dense Adam over every
    parameter, reference semantics) timed on the host cores on `n_steps` steps of the same batch shape."""
    import numpy as np
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle_np as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    state = {k: v.detach().cpu().numpy().copy() for k, v in task.state_dict().items()}
    b = {"notice_ids": batch["notice"]["kjt"].values().cpu().numpy().reshape(B, len(keys_n)),
         "company_ids": batch["company"]["kjt"].values().cpu().numpy().reshape(B, len(keys_c)),
         "notice_dense": batch["notice"]["dense"].cpu().numpy(), "company_dense": batch["company"]["dense"].cpu().numpy()}
    pkeys = [k for k in state if "running" not in k and "num_batches" not in k]
    m = {k: np.zeros_like(state[k]) for k in pkeys}
    v = {k: np.zeros_like(state[k]) for k in pkeys}
    t0 = time.perf_counter()
    for s in range(n_steps):
        out = O.task_step(state, b, keys_n, keys_c, vocab_n, vocab_c, 1.0, True)
        for k in pkeys:
            O.adam_step(state[k], out["grads"][k], m[k], v[k], s + 1, 1e-3, wd=1e-5)
        state.update(out["bn_updates"])
    dt = time.perf_counter() - t0
    return {"value": B * n_steps / dt, "unit": "pairs/s", "cores": threads, "kind": "port",
            "sample": f"{n_steps} steps of batch {B} (forward + backward + dense Adam over all {sum(state[k].size for k in pkeys)} "
                      f"parameters), numpy oracle: BLAS-threaded matmuls on {threads} threads, single-threaded elementwise; "
                      f"{dt:.1f} s of CPU work; host has {os.cpu_count()} logical cores"}


if __name__ == "__main__":
    main()
