"""Worker of test_two_processes_equal_single_process (not collected by pytest: no test_ prefix).

TWO processes share the one GPU of the box; their collectives go through gloo with the tensors staged on the host (RCCL
cannot put two ranks on one device).  Everything else is the product path: row-wise sharded tables behind the
fixed-capacity exchange (HIP routing kernels), global in-batch negatives, SyncBN, dropout masks drawn per global row.
Claim checked: one training step of the 2-rank job on a global batch of 2 x B pairs == the single-process task on the
same 2 x B pairs -- global loss, BN running statistics, every dense gradient (after the data-parallel sum), and the table
rows touched by the step after one Adam update.  Prints DIST_WORLD2_OK."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tests", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import jodalrob_twotower_amd as tt  # noqa: E402
from jodalrob_twotower_amd.distributed import HostStagedComm, create_distributed_train_task  # noqa: E402
from jodalrob_twotower_amd.optim import FusedAdam  # noqa: E402
from params_init import init_state_numpy, synth_batch_numpy  # noqa: E402

DEV = "cuda:0"
GOLD = ROOT / "tests" / "golden"


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = dict(json.load(open(GOLD / "manifest.json"))["cases"]["wide_b40"])        # towers [32, 24] -> 16, E = 16
    if os.environ.get("TT_W2_SEGMENTED") == "1":
        segmented_equals_eager(rank, world, cfg)
        dist.barrier()
        dist.destroy_process_group()
        print("DIST_WORLD2_SEGMENTED_OK", flush=True)
        return
    if os.environ.get("TT_W2_WIDE") == "1":                  # wider than the fused tail takes: the separate kernels' SyncBN cut
        cfg["hidden"], cfg["D"] = [48, 80], 72
    Bl, drop = 96, 0.1
    B = world * Bl
    b = synth_batch_numpy(B, cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 881, oob=True)

    def to_batch(sl):
        return {"notice": {"dense": torch.from_numpy(b["notice_dense"][sl]).to(DEV),
                           "kjt": tt.build_batch_kjt(torch.from_numpy(b["notice_ids"][sl]), cfg["keys_n"]).to(DEV)},
                "company": {"dense": torch.from_numpy(b["company_dense"][sl]).to(DEV),
                            "kjt": tt.build_batch_kjt(torch.from_numpy(b["company_ids"][sl]), cfg["keys_c"]).to(DEV)}}

    common = dict(metadata_path=str(GOLD / "synthetic_metadata.csv"), categorical_embedding_dim=cfg["E"],
                  notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"], tower_hidden_dims=list(cfg["hidden"]),
                  final_embedding_dim=cfg["D"], dropout_rate=drop, temperature=cfg["T"], device=DEV, embedding_grad="sparse",
                  mlp_dtype="bf16", score_dtype="bf16")
    sync = os.environ.get("TT_W2_NO_SYNC") != "1"            # negative control of the test: without SyncBN the claim must fail
    td = create_distributed_train_task(cfg["keys_n"], cfg["keys_c"], exchange="padded", negatives="global", sync_bn=sync,
                                       comm=HostStagedComm(), **common)
    shapes = {k: tuple(v.shape) for k, v in td.full_state_dict().items()}
    state = {k: torch.from_numpy(np.asarray(v)) for k, v in init_state_numpy(shapes, 882).items()}
    td.load_full_state_dict(state)
    ts = tt.create_two_tower_train_task(cfg["keys_n"], cfg["keys_c"], **common)
    ts.load_state_dict(state)
    for task in (td, ts):
        task.train()
        for tw in (task.two_tower_model.notice_tower, task.two_tower_model.company_tower):
            tw._seed_override = 4242
    od, os_ = FusedAdam.for_task(td, lr=1e-2, weight_decay=1e-5), FusedAdam.for_task(ts, lr=1e-2, weight_decay=1e-5)

    od.zero_grad()
    rd = td(to_batch(slice(rank * Bl, (rank + 1) * Bl)), return_metrics=True)
    rd["loss"].backward()
    assert not td.exchange.overflowed()
    os_.zero_grad()
    rs = ts(to_batch(slice(0, B)), return_metrics=True)
    rs["loss"].backward()
    torch.cuda.synchronize()

    loss = torch.tensor([float(rd["loss"])], dtype=torch.float64)
    dist.all_reduce(loss)                                                # objective = mean over ranks of the local means
    assert abs(loss.item() / world - float(rs["loss"])) < 2e-4 * abs(float(rs["loss"])), (loss.item() / world, float(rs["loss"]))
    close = lambda a, ref, tol: float(torch.linalg.norm(a - ref)) <= tol * float(torch.linalg.norm(ref)) + 1e-7
    dn = dict(td.named_parameters())
    for name, p in ts.named_parameters():
        if "embeddings" in name:
            continue
        g = dn[name].grad
        assert g is not None and close(g, p.grad, 1e-2), (name, float(torch.linalg.norm(g - p.grad)), float(torch.linalg.norm(p.grad)))
    od.step(); os_.step()
    torch.cuda.synchronize()
    full_d, full_s = td.full_state_dict(), ts.state_dict()
    for k, v in full_s.items():
        a = full_d[k].to(v.device)
        if "num_batches" in k:
            assert int(a) == int(v), k
        elif "running" in k:
            assert torch.allclose(a, v, rtol=1e-5, atol=1e-7), k
        else:
            # Adam's first step moves every touched weight by +-lr whatever the gradient's size: compare where the
            # gradient is not a rounding-level number (its sign is then the same in both runs)
            assert float((a - v).abs().max()) <= 2.1e-2, (k, float((a - v).abs().max()))
            assert float(((a - v).abs() > 1e-4).float().mean()) < 0.02 + 2.0 / a.numel(), (k, float(((a - v).abs() > 1e-4).float().mean()))
    # the overflow path with real peers (untested on hardware): a capacity far below the need -> rows that do not fit go
    # to the spare row, the sticky flag rises, check_overflow raises on every rank; after reset_capacity the next step is clean again
    from jodalrob_twotower_amd.distributed import ExchangeOverflowError
    mine = to_batch(slice(rank * Bl, (rank + 1) * Bl))
    td.exchange.reset_capacity()
    td.exchange.C = 16
    od.zero_grad()
    td(mine, return_metrics=True)["loss"].backward()
    torch.cuda.synchronize()
    try:
        td.exchange.check_overflow()
        raise AssertionError("a 16-row bucket cannot hold this batch's rows: the overflow flag did not rise")
    except ExchangeOverflowError:
        pass
    td.exchange.reset_capacity()                             # (recalibrates on the next forward)
    od.zero_grad()
    r2 = td(mine, return_metrics=True)
    r2["loss"].backward()
    torch.cuda.synchronize()
    assert not td.exchange.overflowed() and bool(torch.isfinite(r2["loss"]))
    del od, os_, td, ts
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()                             # ordinary teardown and interpreter exit, as the world-1 worker
    print("DIST_WORLD2_OK", flush=True)


def segmented_equals_eager(rank, world, cfg):
    """SegmentedTrainStep with REAL peers: each rank trains on its own batches, rows and row gradients cross ranks in every step, the
    collectives run eagerly (host-staged gloo) between the replayed segments.  Claim: on every rank the losses of 4 steps and the
    whole state afterwards (this rank's table shard, towers, BN statistics) == the same steps issued launch by launch, bit for bit --
    on the default semantics (per-rank negatives and BN statistics) and on global negatives + SyncBN."""
    from jodalrob_twotower_amd.segmented import SegmentedTrainStep
    Bl = 128

    def to_batch(b):
        return {"notice": {"dense": torch.from_numpy(b["notice_dense"]).to(DEV),
                           "kjt": tt.build_batch_kjt(torch.from_numpy(b["notice_ids"]), cfg["keys_n"]).to(DEV)},
                "company": {"dense": torch.from_numpy(b["company_dense"]).to(DEV),
                            "kjt": tt.build_batch_kjt(torch.from_numpy(b["company_ids"]), cfg["keys_c"]).to(DEV)}}

    batches = [to_batch(synth_batch_numpy(Bl, cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 3000 + 17 * rank + i, oob=True))
               for i in range(4)]
    for negatives, sync_bn in (("local", False), ("global", True)):
        finals, state = {}, None
        for mode in ("eager", "segmented"):
            t = create_distributed_train_task(
                cfg["keys_n"], cfg["keys_c"], metadata_path=str(GOLD / "synthetic_metadata.csv"), categorical_embedding_dim=cfg["E"],
                notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"], tower_hidden_dims=list(cfg["hidden"]),
                final_embedding_dim=cfg["D"], dropout_rate=0.0, temperature=cfg["T"], device=DEV, embedding_grad="sparse",
                mlp_dtype="bf16", score_dtype="bf16", exchange="padded", negatives=negatives, sync_bn=sync_bn, comm=HostStagedComm())
            if state is None:
                shapes = {k: tuple(v.shape) for k, v in t.full_state_dict().items()}
                state = {k: torch.from_numpy(np.asarray(v)) for k, v in init_state_numpy(shapes, 556).items()}
            t.load_full_state_dict(state)
            t.train()
            t._pair_check_done = True
            o = FusedAdam.for_task(t, lr=1e-2, weight_decay=1e-5)
            losses, gs = [], None
            if mode == "segmented":
                gs = SegmentedTrainStep(t, o, batches[0], warmup=1, preserve_state=False)
                for bt in batches:
                    losses.append(gs.step(bt)["loss"].item())
            else:
                o.zero_grad(); t(batches[0], return_metrics=True)["loss"].backward(); o.step()     # the capture's warm-up step
                for bt in batches:
                    o.zero_grad()
                    r = t(bt, return_metrics=True)
                    r["loss"].backward()
                    o.step()
                    losses.append(r["loss"].item())
            torch.cuda.synchronize()
            assert not t.exchange.overflowed()
            local = {k: v.detach().cpu().clone() for k, v in t.state_dict().items()}          # (this rank's shard + replicated parts)
            finals[mode] = (losses, local, None if gs is None else (gs.collectives_per_step(), sum(g is not None for g in gs._segments)))
            if gs is not None:
                gs.close()
            del gs, o, t
        assert finals["segmented"][0] == finals["eager"][0], (negatives, finals["segmented"][0], finals["eager"][0])
        assert len(set(finals["eager"][0])) > 1
        for k, v in finals["eager"][1].items():
            assert torch.equal(v, finals["segmented"][1][k]), (negatives, k)
        n_coll, n_seg = finals["segmented"][2]
        print(f"[rank {rank}] segmented, {negatives} negatives, sync_bn={sync_bn}: {n_coll} eager collectives between {n_seg} replayed segments; "
              f"losses {finals['segmented'][0]}", flush=True)
        assert n_coll >= (4 if negatives == "local" else 6), n_coll


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)                                          # a failed rank must not wait in gloo's teardown for its peer
    sys.stdout.flush()
