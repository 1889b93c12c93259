"""Worker of test_padded_exchange_graph_world1_subprocess (not collected by pytest: no test_ prefix).

exact-size exchange (eager) == fixed-capacity exchange (eager) == fixed-capacity exchange captured as ONE graph with the RCCL
all-to-alls / all-reduce inside, over a warm-up step + 4 steps with different batches (bf16 and fp32 MLP): same losses, final
state equal to 1e-6; a sharded run resumed from (full_state_dict, FusedAdam.state_dict()) continues bit for bit.
Teardown is the ordinary one, in the order a graph with RCCL nodes needs: GraphedTrainStep.close() (drops the graph and its
pool) -> tasks / optimisers released -> synchronize -> destroy_process_group() -> normal interpreter exit (rc 0).
Between the replays of the graph that contains RCCL collectives it issues an EAGER RCCL all-reduce on the SAME communicator
(the pattern a checkpoint all-gather inside a training loop produces; round 1 suspected this of faulting the GPU -- it was
the memset-node bug of DESIGN.md section 7, fixed since)."""
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
from jodalrob_twotower_amd.distributed import create_distributed_train_task  # noqa: E402
from jodalrob_twotower_amd.graph import GraphedTrainStep  # noqa: E402
from jodalrob_twotower_amd.segmented import SegmentedTrainStep  # noqa: E402
from jodalrob_twotower_amd.optim import FusedAdam  # noqa: E402
from params_init import init_state_numpy, synth_batch_numpy  # noqa: E402

DEV = "cuda:0"
GOLD = ROOT / "tests" / "golden"


def main():
    import socket
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)     # never the parent's rendezvous port
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    cfg = dict(json.load(open(GOLD / "manifest.json"))["cases"]["wide_b40"])
    cfg["B"] = 256

    def to_batch(b):
        return {"notice": {"dense": torch.from_numpy(b["notice_dense"]).to(DEV),
                           "kjt": tt.build_batch_kjt(torch.from_numpy(b["notice_ids"]), cfg["keys_n"]).to(DEV)},
                "company": {"dense": torch.from_numpy(b["company_dense"]).to(DEV),
                            "kjt": tt.build_batch_kjt(torch.from_numpy(b["company_ids"]), cfg["keys_c"]).to(DEV)}}

    batches = [to_batch(synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 790 + i, oob=True))
               for i in range(4)]
    for mlp in ("fp32", "bf16"):
        finals, state = {}, None
        for mode in ("exact", "padded", "padded-graph", "padded-segmented"):
            t = create_distributed_train_task(
                cfg["keys_n"], cfg["keys_c"], metadata_path=str(GOLD / "synthetic_metadata.csv"), categorical_embedding_dim=cfg["E"],
                notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"], tower_hidden_dims=list(cfg["hidden"]),
                final_embedding_dim=cfg["D"], dropout_rate=0.0, temperature=cfg["T"], device=DEV, embedding_grad="sparse",
                mlp_dtype=mlp, exchange="exact" if mode == "exact" else "padded")
            if state is None:
                shapes = {k: tuple(v.shape) for k, v in t.full_state_dict().items()}
                state = {k: torch.from_numpy(np.asarray(v)) for k, v in init_state_numpy(shapes, 777).items()}
            t.load_full_state_dict(state)
            t.train()
            o = FusedAdam.for_task(t, lr=1e-2, weight_decay=1e-5)
            losses = []
            gs = None
            if mode in ("padded-graph", "padded-segmented"):
                # padded-segmented: the fallback for a step whose collectives cannot be captured -- the compute between two collectives
                # as graphs of its own, the collectives eager between the replays (segmented.SegmentedTrainStep)
                Step = GraphedTrainStep if mode == "padded-graph" else SegmentedTrainStep
                gs = Step(t, o, batches[0], warmup=1, preserve_state=False)   # the eager warm-up step (a real step here, as in the eager leg) calibrates the bucket capacity
                if mode == "padded-segmented":
                    print(f"[segmented] {mlp}: {gs.collectives_per_step()} collectives, {sum(g is not None for g in gs._segments)} graph segments "
                          f"of {len(gs._segments)}", flush=True)
                    assert gs.collectives_per_step() == 4, gs.collectives_per_step()      # ids, rows, row gradients (all-to-all), dense gradients (all-reduce)
                for bt in batches:
                    losses.append(gs.step(bt)["loss"].item())
                    probe = torch.full((1024,), 3.0, device=DEV)
                    dist.all_reduce(probe)                              # eager, same communicator, between two replays
                    assert float(probe.sum().item()) == 3.0 * 1024
            else:
                o.zero_grad(); t(batches[0], return_metrics=True)["loss"].backward(); o.step()     # the same warm-up step
                for bt in batches:
                    o.zero_grad()
                    r = t(bt, return_metrics=True)
                    r["loss"].backward()
                    o.step()
                    losses.append(r["loss"].item())
            if mode != "exact":
                assert not t.exchange.overflowed()
            finals[mode] = (losses, {k: v.detach().cpu().numpy().copy() for k, v in t.full_state_dict().items()})
            if gs is not None:
                gs.close()                                              # graph + pool go before the communicator does
            del gs, o, t
        for mode in ("padded", "padded-graph", "padded-segmented"):
            assert finals[mode][0] == finals["exact"][0], (mlp, mode, finals[mode][0], finals["exact"][0])
            for k, v in finals["exact"][1].items():
                np.testing.assert_allclose(finals[mode][1][k], v, rtol=1e-6, atol=1e-7, err_msg=f"{mlp}:{mode}:{k}")
    resume_check(cfg, batches)
    deferred_slabs_before_all_reduce()
    segmented_global_negatives(cfg, batches)
    import gc
    gc.collect()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("DIST_WORLD1_OK", flush=True)


def segmented_global_negatives(cfg, batches):
    """The segmented fallback on the SURVEY-C3 semantics (global in-batch negatives + SyncBN: all-gathers back to back leave empty
    segments, the SyncBN tail runs in two phases with a collective between them) == the same steps eager, bit for bit."""
    finals, state = {}, None
    for mode in ("eager", "segmented"):
        t = create_distributed_train_task(
            cfg["keys_n"], cfg["keys_c"], metadata_path=str(GOLD / "synthetic_metadata.csv"), categorical_embedding_dim=cfg["E"],
            notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"], tower_hidden_dims=list(cfg["hidden"]),
            final_embedding_dim=cfg["D"], dropout_rate=0.0, temperature=cfg["T"], device=DEV, embedding_grad="sparse",
            mlp_dtype="bf16", score_dtype="bf16", exchange="padded", negatives="global", sync_bn=True)
        if state is None:
            shapes = {k: tuple(v.shape) for k, v in t.full_state_dict().items()}
            state = {k: torch.from_numpy(np.asarray(v)) for k, v in init_state_numpy(shapes, 555).items()}
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
        finals[mode] = (losses, {k: v.detach().cpu().clone() for k, v in t.full_state_dict().items()}, None if gs is None else gs.collectives_per_step())
        if gs is not None:
            gs.close()
        del gs, o, t
    assert finals["segmented"][0] == finals["eager"][0], (finals["segmented"][0], finals["eager"][0])
    for k, v in finals["eager"][1].items():
        assert torch.equal(v, finals["segmented"][1][k]), k
    print(f"[segmented, global negatives + SyncBN] {finals['segmented'][2]} collectives per step", flush=True)
    assert finals["segmented"][2] >= 4, finals["segmented"][2]       # (at world 1 the global-negative / SyncBN gathers are short-circuited: 4 cuts, as the default semantics)


def deferred_slabs_before_all_reduce():
    """GraphedTrainStep lets the towers' slab reduction (which writes the first block's and the
    projection's weight gradients) ride in the embedding gradient's launch -- but on the sharded task the dense gradients are
    all-reduced BEFORE that launch.  The queue must have run by then.  Two ranks cannot share this box's one GPU under RCCL
    and a host-staged wire cannot be captured, so the order is pinned at world 1 with a communicator whose all-reduce also
    doubles its argument (two ranks holding the same gradient): replayed and eager steps must then agree bit for bit on every
    dense parameter, nothing may be pending when the all-reduce is issued, and the deferral must have been on (else the test
    proves nothing).  Reference-shaped bf16 towers [128, 64] -> 64 on the real 32 + 6 key schema: the shapes that take the
    one-launch first-block backward whose slabs are deferred."""
    from jodalrob_twotower_amd import _lib as L
    from jodalrob_twotower_amd.distributed import DistComm
    real = json.load(open(GOLD / "schema_real.json"))
    kn, kc = real["notice"]["categorical"], real["company"]["categorical"]
    vn, vc = real["notice"]["vocab_sizes"], real["company"]["vocab_sizes"]
    seen = {"pending": [], "defer_on": []}

    class DoublingComm(DistComm):
        def all_reduce_sum(self, t):
            seen["pending"].append(int(L.load().tt_deferred_pending(L.ctx(t.device))) & 1)
            seen["defer_on"].append(bool(L._defer_on))
            super().all_reduce_sum(t)
            return t.mul_(2.0)

    def to_batch(b):
        return {"notice": {"dense": torch.from_numpy(b["notice_dense"]).to(DEV), "kjt": tt.build_batch_kjt(torch.from_numpy(b["notice_ids"]), kn).to(DEV)},
                "company": {"dense": torch.from_numpy(b["company_dense"]).to(DEV), "kjt": tt.build_batch_kjt(torch.from_numpy(b["company_ids"]), kc).to(DEV)}}

    batches = [to_batch(synth_batch_numpy(256, vn, vc, 256, 128, 990 + i, oob=False)) for i in range(4)]
    finals, state = {}, None
    for mode in ("eager", "graph"):
        t = create_distributed_train_task(kn, kc, metadata_path=str(GOLD / "real_vocab_metadata.csv"), categorical_embedding_dim=32,
                                          notice_dense_input_dim=256, company_dense_input_dim=128, tower_hidden_dims=[128, 64],
                                          final_embedding_dim=64, dropout_rate=0.0, temperature=1.0, device=DEV, embedding_grad="sparse",
                                          mlp_dtype="bf16", score_dtype="bf16", exchange="padded", comm=DoublingComm())
        if state is None:
            shapes = {k: tuple(v.shape) for k, v in t.full_state_dict().items()}
            state = {k: torch.from_numpy(np.asarray(v)) for k, v in init_state_numpy(shapes, 991).items()}
        t.load_full_state_dict(state)
        t.train()
        o = FusedAdam.for_task(t, lr=1e-2, weight_decay=1e-5)
        gs = None
        if mode == "graph":
            seen["pending"].clear(); seen["defer_on"].clear()
            gs = GraphedTrainStep(t, o, batches[0], warmup=1, preserve_state=False)
            assert any(seen["defer_on"]), "the slab deferral was never on while the dense all-reduce was issued: nothing tested"
            assert not any(seen["pending"]), "a slab reduction was still queued when the dense gradients were all-reduced"
            for bt in batches:
                gs.step(bt)
        else:
            o.zero_grad(); t(batches[0], return_metrics=True)["loss"].backward(); o.step()       # the capture's warm-up step
            for bt in batches:
                o.zero_grad()
                t(bt, return_metrics=True)["loss"].backward()
                o.step()
        torch.cuda.synchronize()
        finals[mode] = {k: v.detach().cpu().clone() for k, v in t.full_state_dict().items()}
        if gs is not None:
            gs.close()
        del gs, o, t
    for k, v in finals["eager"].items():
        if "embeddings" in k or "num_batches" in k:
            continue
        assert torch.equal(finals["graph"][k], v), (k, float((finals["graph"][k] - v).abs().max()))


def resume_check(cfg, batches):
    """save -> load -> step on the sharded path: model through full_state_dict / load_full_state_dict, optimiser through
    FusedAdam.state_dict / load_state_dict (the shard's Adam moments and step count must survive: ADVICE round 1)."""
    import copy

    def make():
        t = create_distributed_train_task(
            cfg["keys_n"], cfg["keys_c"], metadata_path=str(GOLD / "synthetic_metadata.csv"), categorical_embedding_dim=cfg["E"],
            notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"], tower_hidden_dims=list(cfg["hidden"]),
            final_embedding_dim=cfg["D"], dropout_rate=0.0, temperature=cfg["T"], device=DEV, embedding_grad="sparse", mlp_dtype="bf16")
        t.train()
        return t, FusedAdam.for_task(t, lr=1e-2, weight_decay=1e-5)

    def run(t, o, bs):
        out = []
        for bt in bs:
            o.zero_grad()
            r = t(bt, return_metrics=True)
            r["loss"].backward()
            o.step()
            out.append(r["loss"].item())
        return out

    ta, oa = make()
    shapes = {k: tuple(v.shape) for k, v in ta.full_state_dict().items()}
    ta.load_full_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in init_state_numpy(shapes, 778).items()})
    run(ta, oa, batches[:2])
    snap_model = {k: v.detach().clone() for k, v in ta.full_state_dict().items()}
    snap_opt = copy.deepcopy(oa.state_dict())
    la = run(ta, oa, batches[2:])
    tb, ob = make()
    tb.load_full_state_dict(snap_model)
    ob.load_state_dict(snap_opt)
    shard_state = ob.state[tb.embedding_shard]
    assert float(shard_state["step"]) == 2.0 and shard_state["exp_avg"].abs().sum().item() > 0      # moments arrived
    assert shard_state["exp_avg"].data_ptr() == ob._store_state[id(tb.sharded_store)]["m"].data_ptr()   # and alias the kernels' buffers
    lb = run(tb, ob, batches[2:])
    assert la == lb, (la, lb)
    fa, fb = ta.full_state_dict(), tb.full_state_dict()
    for k in fa:
        assert torch.equal(fa[k], fb[k]), k
    assert torch.equal(oa.state[ta.embedding_shard]["exp_avg_sq"], ob.state[tb.embedding_shard]["exp_avg_sq"])


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)
    sys.stdout.flush()
