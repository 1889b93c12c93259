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
    del od, os_, td, ts
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()                             # ordinary teardown and interpreter exit, as the world-1 worker
    print("DIST_WORLD2_OK", flush=True)


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)                                          # a failed rank must not wait in gloo's teardown for its peer
    sys.stdout.flush()
