import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
import torch.distributed as dist
import jodalrob_twotower_amd as tt
from jodalrob_twotower_amd.distributed import create_distributed_train_task
from jodalrob_twotower_amd.optim import FusedAdam
from jodalrob_twotower_amd.graph import GraphedTrainStep
from conftest import GOLD
from params_init import init_state_numpy, synth_batch_numpy
DEV = "cuda:0"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
manifest = json.load(open(GOLD / "manifest.json"))
cfg = dict(manifest["cases"]["wide_b40"]); cfg["B"] = 256
modes = sys.argv[1].split(",")
def to_batch(b):
    return {"notice": {"dense": torch.from_numpy(b["notice_dense"]).to(DEV), "kjt": tt.build_batch_kjt(torch.from_numpy(b["notice_ids"]), cfg["keys_n"]).to(DEV)},
            "company": {"dense": torch.from_numpy(b["company_dense"]).to(DEV), "kjt": tt.build_batch_kjt(torch.from_numpy(b["company_ids"]), cfg["keys_c"]).to(DEV)}}
batches = [to_batch(synth_batch_numpy(cfg["B"], cfg["vocab_n"], cfg["vocab_c"], cfg["din_n"], cfg["din_c"], 790 + i, oob=True)) for i in range(4)]
state = None
for mode in modes:
    print("mode", mode, flush=True)
    t = create_distributed_train_task(cfg["keys_n"], cfg["keys_c"], metadata_path=str(GOLD / "synthetic_metadata.csv"), categorical_embedding_dim=cfg["E"],
        notice_dense_input_dim=cfg["din_n"], company_dense_input_dim=cfg["din_c"], tower_hidden_dims=list(cfg["hidden"]),
        final_embedding_dim=cfg["D"], dropout_rate=0.0, temperature=cfg["T"], device=DEV, embedding_grad="sparse",
        exchange="exact" if mode == "exact" else "padded")
    if state is None:
        state = {k: v.detach().clone() for k, v in t.full_state_dict().items()}
    t.load_full_state_dict(state)
    t.train()
    o = FusedAdam.for_task(t, lr=1e-2, weight_decay=1e-5)
    losses = []
    if mode == "padded-graph":
        gs = GraphedTrainStep(t, o, batches[0], warmup=1)
        print("captured; checksum after warm-up", float(sum(p.double().sum() for p in t.parameters())), flush=True)
        for bt in batches:
            losses.append(gs.step(bt)["loss"].item()); print("step", losses[-1], flush=True)
    else:
        o.zero_grad(); r0 = t(batches[0], return_metrics=True); r0["loss"].backward(); o.step()
        print("warm ok loss", r0["loss"].item(), "checksum after warm-up", float(sum(p.double().sum() for p in t.parameters())), flush=True)
        for bt in batches:
            o.zero_grad(); r = t(bt, return_metrics=True); r["loss"].backward(); o.step(); losses.append(r["loss"].item()); print("step", losses[-1], flush=True)
    fs = t.full_state_dict()
    print("full_state_dict ok", len(fs), flush=True)
    if mode != "exact":
        print("overflow", t.exchange.overflowed(), "C", t.exchange.C, flush=True)
dist.destroy_process_group()
