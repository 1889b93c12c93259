#!/usr/bin/env python3
"""Training driver on the MI355X path -- the counterpart of the reference's scripts/train.py
(config dict :84-134, Adam + LambdaLR warm-up :231-242, loop :313-352, validation :367-423, evaluator
:444-452, checkpoints :497-551, results CSV :24-75).  The reference driver needs PostgreSQL
(DatabaseConnector(), :159); this one takes the same config keys and a synthetic feature/pair source.

    python scripts/train.py [--resume PATH] [--steps N] [--batch-size B] [--fast]

--fast: the loop the MI355X path is built for -- bf16 towers and score operands, row-sparse table gradients, FusedAdam, the
whole step replayed from ONE captured HIP graph, every full batch gathered out of the device-resident feature stores by the
step's own hand-over launch (DevicePairLoader.step_batches -> GraphedTrainStep.step_from_store -> tt_batch_ingest_store): no
batch tensors, nothing per step over PCIe.  Without the flag the loop is the reference's own, statement for statement
(eager forward / backward / optimiser step on the loader's batches, f32 parity kernels).  Either way the run ends like the
reference driver: validation, evaluator report, prediction demo, checkpoints, a results-CSV row; and prints pairs/s over the
whole epoch including evaluation.
"""
from __future__ import annotations

import argparse
import csv
import sys
import tempfile
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "jodalrob-twotower_amd" / "dropin"))

from src.evaluation.evaluator import TwoTowerEvaluator                       # noqa: E402  (same imports as the reference driver)
from src.torchrec_preprocess.schema import build_torchrec_schema_from_meta    # noqa: E402
from src.towers.pairs.unified_bid_data_loader import create_unified_bid_dataloaders  # noqa: E402
from src.towers.two_tower_train_task import create_two_tower_train_task       # noqa: E402
from jodalrob_twotower_amd import synthetic                                   # noqa: E402
from jodalrob_twotower_amd.optim import FusedAdam                             # noqa: E402

RESULT_COLUMNS = ["timestamp", "batch_size", "tower_hidden_dims", "final_embedding_dim", "categorical_embedding_dim",
                  "learning_rate", "num_epochs", "train_batches", "final_train_loss", "final_train_accuracy", "val_loss",
                  "val_accuracy", "recall@5", "recall@10", "mrr", "similarity_gap", "total_params", "train_seconds"]


def save_checkpoint(model, optimizer, epoch, loss, save_dir, is_best=False, is_final=False):
    save_dir = Path(save_dir)
    save_dir.mkdir(parents=True, exist_ok=True)
    ckpt = {"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict(), "loss": loss}
    if not is_final:
        torch.save(ckpt, save_dir / f"checkpoint_epoch_{epoch + 1}.pt")
    if is_best:
        torch.save(ckpt, save_dir / "best_model.pt")
    if is_final:
        torch.save(ckpt, save_dir / "final_model.pt")
        torch.save(model.state_dict(), save_dir / "model_weights.pt")


def load_checkpoint(model, optimizer, checkpoint_path):
    ckpt = torch.load(checkpoint_path, map_location="cuda:0", weights_only=True)
    model.load_state_dict(ckpt["model_state_dict"])
    optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return ckpt["epoch"] + 1, ckpt["loss"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--resume", default=None)
    ap.add_argument("--batch-size", type=int, default=256)
    ap.add_argument("--entities", type=int, default=10_000)
    ap.add_argument("--pairs", type=int, default=100_000)
    ap.add_argument("--steps", type=int, default=None, help="stop after this many training steps")
    ap.add_argument("--output-dir", default="output/models")
    ap.add_argument("--fast", action="store_true", help="captured step fed from the device stores, bf16 operands, sparse gradients")
    ap.add_argument("--hidden", default="128,64", help="tower_hidden_dims (the reference driver trains 512,256: scripts/train.py:106)")
    ap.add_argument("--final-dim", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=1)
    a = ap.parse_args()
    config = {"batch_size": a.batch_size, "test_split": 0.2, "shuffle_seed": 42, "pair_limit": a.pairs,
              "categorical_embedding_dim": 32, "notice_dense_input_dim": 256, "company_dense_input_dim": 128,
              "tower_hidden_dims": [int(h) for h in a.hidden.split(",")], "final_embedding_dim": a.final_dim, "dropout_rate": 0.1,
              "temperature": 1.0, "loss_type": "cross_entropy", "learning_rate": 1e-3, "weight_decay": 1e-5, "num_epochs": a.epochs,
              "warmup_ratio": 0.05,
              "log_interval": 20, "output_dir": a.output_dir}
    device = torch.device("cuda:0")
    real = synthetic.load_real_schema(ROOT / "jodalrob-twotower_amd" / "schema_real.json")
    tmp = Path(tempfile.mkdtemp(prefix="tt_train_"))
    meta_rows = [synthetic.META_HEADER]
    for side in ("notice", "company"):
        for c in real[side]["pk_cols"]:
            meta_rows.append(f"{side},{c},text,Y,,,,0,,Y,Y,,")
        for c in real[side]["numeric"]:
            meta_rows.append(f"{side},{c},numeric,Y,,,,0,,,,,")
        for c, v in zip(real[side]["categorical"], real[side]["vocab_sizes"]):
            meta_rows.append(f"{side},{c},text,Y,,Y,{v - 10},0,,,,,")
        for c in real[side]["text"]:
            meta_rows.append(f"{side},{c},text,Y,,N,,0,,,,,")
    meta = tmp / "metadata.csv"
    meta.write_text("\n".join(meta_rows) + "\n", encoding="utf-8")
    schema = build_torchrec_schema_from_meta(notice_table="notice", company_table="company", pair_table="bid_two_tower",
                                             pair_notice_id_cols=["bidntceno", "bidntceord"], pair_company_id_cols=["bizno"],
                                             metadata_path=str(meta))
    source = synthetic.SyntheticSource(a.entities, a.entities, a.pairs, real["notice"]["vocab_sizes"], real["company"]["vocab_sizes"])
    train_loader, test_loader = create_unified_bid_dataloaders(source, schema, batch_size=config["batch_size"],
                                                               test_split=config["test_split"], shuffle_seed=config["shuffle_seed"],
                                                               test_mode=True, pair_limit=config["pair_limit"], device=device)
    train_task = create_two_tower_train_task(schema.notice.categorical, schema.company.categorical, metadata_path=str(meta),
                                             categorical_embedding_dim=config["categorical_embedding_dim"],
                                             notice_dense_input_dim=config["notice_dense_input_dim"],
                                             company_dense_input_dim=config["company_dense_input_dim"],
                                             tower_hidden_dims=config["tower_hidden_dims"],
                                             final_embedding_dim=config["final_embedding_dim"], dropout_rate=config["dropout_rate"],
                                             temperature=config["temperature"], loss_type=config["loss_type"], device=device,
                                             **(dict(embedding_grad="sparse", score_dtype="bf16", mlp_dtype="bf16") if a.fast else {}))
    optimizer = FusedAdam.for_task(train_task, lr=config["learning_rate"], weight_decay=config["weight_decay"])
    warmup_steps = max(1, int(len(train_loader) * config["warmup_ratio"]))
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: s / warmup_steps if s < warmup_steps else 1.0, last_epoch=-1)
    start_epoch = 0
    if a.resume:
        start_epoch, _ = load_checkpoint(train_task, optimizer, a.resume)
    total_params = sum(p.numel() for p in train_task.parameters())
    print(f"params: {total_params:,}  train batches: {len(train_loader)}  warm-up steps: {warmup_steps}")
    evaluator = TwoTowerEvaluator(device=device)
    t0, steps, best = time.time(), 0, float("inf")
    train_losses, train_accs, val = [], [], {}
    if start_epoch >= config["num_epochs"]:
        config["num_epochs"] = start_epoch + 1           # resumed after the last epoch: run one more

    def eager_step(batch):
        optimizer.zero_grad()
        result = train_task(batch, return_metrics=True)
        result["loss"].backward()
        optimizer.step()
        return result

    graphed = None
    if a.fast and len(train_loader) > 1:                 # (a loader of one ragged batch has nothing to capture)
        from jodalrob_twotower_amd.graph import GraphedTrainStep
        train_task.train()
        state = train_loader._gen.get_state()
        example = next(iter(train_loader))               # shapes + the capture's one eager warm-up step: runs at the schedule's
        train_loader._gen.set_state(state)               # first learning rate, which is 0 (LambdaLR warm-up: scripts/train.py:236-240)
        graphed = GraphedTrainStep(train_task, optimizer, example, warmup=1)
    if a.fast:                                           # one-off: capture the evaluation pass too (outside the epoch clock, like the step's capture)
        evaluator._fast_eval(train_task, test_loader)
    torch.cuda.synchronize()
    pairs_seen, t_loop, t_train, t_eval = 0, time.time(), 0.0, 0.0
    for epoch in range(start_epoch, config["num_epochs"]):
        t_e0 = time.time()
        train_task.train()
        results = train_loader.step_batches(graphed, eager_step) if graphed is not None else (eager_step(b) for b in train_loader)
        for result in results:
            scheduler.step()
            steps += 1
            pairs_seen += config["batch_size"]
            if steps % config["log_interval"] == 0:
                train_losses.append(result["loss"].item())
                train_accs.append(result["accuracy"].item())
                print(f"step {steps:5d}  loss {train_losses[-1]:.4f}  acc {train_accs[-1]:.4f}  "
                      f"pos {result['positive_similarity_mean'].item():.3f}  neg {result['negative_similarity_mean'].item():.3f}")
            if a.steps and steps >= a.steps:
                break
        torch.cuda.synchronize()
        t_e1 = time.time()
        val = evaluator.evaluate_comprehensive(train_task, test_loader, verbose=True, max_batches=50)
        torch.cuda.synchronize()
        t_train, t_eval = t_train + (t_e1 - t_e0), t_eval + (time.time() - t_e1)
        if val.get("loss", float("inf")) < best:
            best = val["loss"]
            save_checkpoint(train_task, optimizer, epoch, best, config["output_dir"], is_best=True)
        else:
            save_checkpoint(train_task, optimizer, epoch, val.get("loss", 0.0), config["output_dir"])
    torch.cuda.synchronize()
    t_epochs = time.time() - t_loop
    print(f"throughput: {pairs_seen / t_epochs:,.0f} pairs/s over {steps} steps incl. evaluation "
          f"({'fast: captured step fed from the device stores' if a.fast else 'eager reference loop'}); "
          f"training {t_train * 1e3:.1f} ms = {pairs_seen / max(t_train, 1e-9):,.0f} pairs/s, evaluation {t_eval * 1e3:.1f} ms "
          f"({val.get('num_batches', 0)} batches), checkpoints {max(t_epochs - t_train - t_eval, 0.0) * 1e3:.1f} ms")
    evaluator.demonstrate_predictions(train_task, next(iter(test_loader)), top_k=10)      # reference driver: scripts/train.py:450-452
    if graphed is not None:
        graphed.close()
    evaluator.close()                                    # (captured evaluation graphs)
    torch.cuda.synchronize()
    save_checkpoint(train_task, optimizer, config["num_epochs"] - 1, best, config["output_dir"], is_final=True)
    row = [time.strftime("%Y-%m-%d %H:%M:%S"), config["batch_size"], str(config["tower_hidden_dims"]), config["final_embedding_dim"],
           config["categorical_embedding_dim"], config["learning_rate"], config["num_epochs"], steps,
           train_losses[-1] if train_losses else "", train_accs[-1] if train_accs else "", val.get("loss", ""), val.get("accuracy", ""),
           val.get("recall@5", ""), val.get("recall@10", ""), val.get("mrr", ""), val.get("similarity_gap", ""), total_params,
           round(time.time() - t0, 2)]
    out_csv = Path(config["output_dir"]).parent / "train_results.csv"
    new = not out_csv.exists()
    with open(out_csv, "a", newline="") as f:
        w = csv.writer(f)
        if new:
            w.writerow(RESULT_COLUMNS)
        w.writerow(row)
    print(f"done: {steps} steps in {time.time() - t0:.1f}s; results appended to {out_csv}")


if __name__ == "__main__":
    main()
