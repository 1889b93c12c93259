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

# the reference's results CSV, column for column and in its order (scripts/train.py:37-57; pinned by tests/golden/api_surface.json
# "harness" -> "results_csv"): a train_results.csv written here concatenates with one the reference wrote
RESULT_COLUMNS = ["timestamp", "batch_size", "model_params", "embedding_dim", "final_embedding_dim", "hidden_dims", "learning_rate",
                  "epochs", "train_loss", "train_acc", "val_loss", "val_acc", "recall_at_5", "recall_at_10", "mrr", "similarity_gap",
                  "train_batches", "test_batches", "gpu_optimization"]
_FROM_HYPERPARAMS = {"batch_size", "model_params", "embedding_dim", "final_embedding_dim", "hidden_dims", "learning_rate", "epochs",
                     "train_batches", "test_batches", "gpu_optimization"}


def save_training_results(hyperparams, metrics, output_file="train_results.csv"):
    """One row per run appended to `output_file` (reference: scripts/train.py:24-75, same signature and default file).  Missing
    values are written as "N/A" like the reference's `.get(key, "N/A")`.  One deliberate difference: the reference's driver fills
    `final_metrics["recall@5"]` / `["recall@10"]` (:479-480) but its writer reads `recall_at_5` / `recall_at_10` (:50-51), so its
    own two columns are always empty (train_results.csv:2-4); here either spelling is accepted and the columns carry the values."""
    row = {"timestamp": time.strftime("%Y-%m-%d %H:%M:%S")}
    for col in RESULT_COLUMNS[1:]:
        if col in _FROM_HYPERPARAMS:
            v = hyperparams.get(col, "N/A")
            row[col] = str(v) if col == "hidden_dims" else v
        else:
            v = metrics.get(col, metrics.get(col.replace("_at_", "@"), "N/A"))
            row[col] = v
    out = Path(output_file)
    new = not out.exists()
    with open(out, "a", newline="", encoding="utf-8") as f:
        w = csv.DictWriter(f, fieldnames=RESULT_COLUMNS)
        if new:
            w.writeheader()
        w.writerow(row)
    print(f"학습 결과 저장 완료: {out}")
    return row


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
    ap.add_argument("--results-csv", default="train_results.csv", help="the results CSV (reference: train_results.csv in the working directory)")
    a = ap.parse_args()
    config = {"batch_size": a.batch_size, "test_split": 0.2, "shuffle_seed": 42, "pair_limit": a.pairs,
              "categorical_embedding_dim": 32, "notice_dense_input_dim": 256, "company_dense_input_dim": 128,
              "tower_hidden_dims": [int(h) for h in a.hidden.split(",")], "final_embedding_dim": a.final_dim, "dropout_rate": 0.1,
              "temperature": 1.0, "loss_type": "cross_entropy", "learning_rate": 1e-3, "weight_decay": 1e-5, "num_epochs": a.epochs,
              "warmup_ratio": 0.05, "log_interval": 20, "output_dir": a.output_dir,
              "gpu_optimization": ("MI355X HIP graph replay + device-resident stores + bf16 MFMA / sparse FusedAdam" if a.fast
                                   else "MI355X HIP kernels, eager, f32 parity mode")}
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
    t0, steps, best_val_loss = time.time(), 0, float("inf")
    if start_epoch >= config["num_epochs"]:
        config["num_epochs"] = start_epoch + 1           # resumed after the last epoch: run one more

    def eager_step(batch):
        optimizer.zero_grad()
        result = train_task(batch, return_metrics=True)
        result["loss"].backward()
        optimizer.step()
        return result

    graphed, ragged = None, None
    if a.fast and len(train_loader) > 1:                 # (a loader of one ragged batch has nothing to capture)
        from jodalrob_twotower_amd.graph import GraphedTrainStep
        train_task.train()
        state = train_loader._gen.get_state()
        example = next(iter(train_loader))               # shapes only: the capture's warm-up steps leave no trace (preserve_state)
        train_loader._gen.set_state(state)
        graphed = GraphedTrainStep(train_task, optimizer, example, warmup=1, accumulate_metrics=True)
        rag = train_loader.ragged_example()              # the epoch's last, smaller batch gets a captured step of its own size:
        if rag is not None and rag["notice"]["dense"].shape[0] >= 2:       # launched eagerly it costs six full steps of host time
            ragged = GraphedTrainStep(train_task, optimizer, rag, warmup=1, accumulate_metrics=True)
    if a.fast:                                           # one-off: capture the evaluation pass too (outside the epoch clock, like the step's capture)
        evaluator._fast_eval(train_task, test_loader)
    log = _AsyncLog(device) if a.fast else None
    torch.cuda.synchronize()
    pairs_seen, t_loop, t_train, t_eval = 0, time.time(), 0.0, 0.0
    avg_train_loss = avg_train_acc = avg_val_loss = avg_val_acc = float("nan")
    val = {}
    stop = False
    for epoch in range(start_epoch, config["num_epochs"]):
        print(f"\nEpoch {epoch + 1}/{config['num_epochs']}")
        t_e0 = time.time()
        train_task.train()
        train_losses, train_accuracies, n_epoch_steps = [], [], 0
        for g in (graphed, ragged):
            if g is not None:
                g.metric_sums.zero_()
        results = train_loader.step_batches(graphed, eager_step, ragged) if graphed is not None else (eager_step(b) for b in train_loader)
        for result in results:
            scheduler.step()
            steps += 1
            n_epoch_steps += 1
            pairs_seen += config["batch_size"]
            if graphed is None:                          # the reference's loop reads loss and accuracy every step (:335-336)
                train_losses.append(result["loss"].item())
                train_accuracies.append(result["accuracy"].item())
                if steps % config["log_interval"] == 0:
                    pos, neg = result["positive_similarity_mean"].item(), result["negative_similarity_mean"].item()
                    gap = result["similarity_gap"].item()
                    print(f"step {steps:5d}  loss {train_losses[-1]:.4f}  acc {train_accuracies[-1]:.4f}  pos {pos:.3f}  neg {neg:.3f}  "
                          f"z-gap {gap / max(abs(neg) + 1e-8, 1e-8):.2f}")
            elif steps % config["log_interval"] == 0:    # fast loop: the figures travel to the host behind the step, nobody waits
                log.post(steps, result.out8)
                log.drain()
            if a.steps and steps >= a.steps:
                stop = True
                break
        if graphed is not None:                          # epoch means from the in-replay sums: ONE host sync per epoch
            sums = graphed.metric_sums.clone()
            if ragged is not None:
                sums += ragged.metric_sums
            sums = sums.cpu()
            log.drain(wait=True)
            # (a ragged batch that went through eager_step is not in the sums: with `ragged` captured there is none)
            avg_train_loss, avg_train_acc = float(sums[0]) / max(n_epoch_steps, 1), float(sums[1]) / max(n_epoch_steps, 1)
        else:
            avg_train_loss = sum(train_losses) / max(len(train_losses), 1)                  # :357-358
            avg_train_acc = sum(train_accuracies) / max(len(train_accuracies), 1)
        torch.cuda.synchronize()
        t_e1 = time.time()
        # the library's sticky device error word (a chained plan launch that gave up, a fused row outside its table): read where the host
        # has synchronised anyway; a raised word invalidates the epoch's steps (include/twotower.h: tt_ctx_check_device_errors)
        from jodalrob_twotower_amd import _lib as _tt_lib
        _tt_lib.check_device_errors(torch.device(device))
        print(f"Train - Loss: {avg_train_loss:.4f}, Accuracy: {avg_train_acc:.3f}")
        # validation (:362-423): mean loss / accuracy over the test loader's batches, model in eval mode
        avg_val_loss = avg_train_loss
        if len(test_loader) > 0:
            if a.fast:                                   # the captured evaluation pass gives the same two means (and the ranks on top)
                val = evaluator.evaluate_comprehensive(train_task, test_loader, verbose=False)
                avg_val_loss, avg_val_acc = val["loss"], val["accuracy"]
            else:
                train_task.eval()
                vl, va = [], []
                with torch.no_grad():
                    for batch in test_loader:
                        result = train_task(batch, return_metrics=True)
                        vl.append(result["loss"].item())
                        va.append(result["accuracy"].item())
                avg_val_loss, avg_val_acc = sum(vl) / len(vl), sum(va) / len(va)
            print(f"Val   - Loss: {avg_val_loss:.4f}, Accuracy: {avg_val_acc:.3f}")
        torch.cuda.synchronize()
        t_train, t_eval = t_train + (t_e1 - t_e0), t_eval + (time.time() - t_e1)
        save_checkpoint(train_task, optimizer, epoch, avg_val_loss, config["output_dir"])                    # :426
        if avg_val_loss < best_val_loss:                                                                      # :429-432
            best_val_loss = avg_val_loss
            save_checkpoint(train_task, optimizer, epoch, avg_val_loss, config["output_dir"], is_best=True)
            print(f"새로운 최고 성능! Loss: {avg_val_loss:.4f}")
        if stop:
            break
    torch.cuda.synchronize()
    t_epochs = time.time() - t_loop
    print(f"throughput: {pairs_seen / t_epochs:,.0f} pairs/s over {steps} steps incl. evaluation "
          f"({'fast: captured step fed from the device stores' if a.fast else 'eager reference loop'}); "
          f"training {t_train * 1e3:.1f} ms = {pairs_seen / max(t_train, 1e-9):,.0f} pairs/s = {t_train * 1e3 / max(steps, 1):.4f} ms/step, "
          f"evaluation {t_eval * 1e3:.1f} ms ({len(test_loader)} batches), checkpoints {max(t_epochs - t_train - t_eval, 0.0) * 1e3:.1f} ms")
    # final evaluation + prediction demo (:441-452)
    print("\n=== 최종 평가 및 추론 테스트 ===")
    if len(test_loader) > 0:
        print("테스트 데이터 종합 평가:")
        test_metrics = evaluator.evaluate_comprehensive(train_task, test_loader, verbose=True)
    else:
        print("훈련 데이터 샘플로 평가:")
        test_metrics = evaluator.evaluate_single_batch(train_task, next(iter(train_loader)), verbose=True)
    evaluator.demonstrate_predictions(train_task, next(iter(train_loader)), top_k=10)
    for g in (graphed, ragged):
        if g is not None:
            g.close()
    evaluator.close()                                    # (captured evaluation graphs)
    torch.cuda.synchronize()
    # results CSV (:455-487): the reference's two dicts, its columns
    print("\n=== 학습 결과 기록 ===")
    hyperparams = {"batch_size": config["batch_size"], "model_params": total_params, "embedding_dim": config["categorical_embedding_dim"],
                   "final_embedding_dim": config["final_embedding_dim"], "hidden_dims": config["tower_hidden_dims"],
                   "learning_rate": config["learning_rate"], "weight_decay": config["weight_decay"], "dropout_rate": config["dropout_rate"],
                   "temperature": config["temperature"], "epochs": config["num_epochs"], "train_batches": len(train_loader),
                   "test_batches": len(test_loader), "gpu_optimization": config["gpu_optimization"]}
    has_test = len(test_loader) > 0
    final_metrics = {"train_loss": avg_train_loss, "train_acc": avg_train_acc,
                     "val_loss": avg_val_loss if has_test else "N/A", "val_acc": avg_val_acc if has_test else "N/A",
                     "recall@5": test_metrics.get("recall@5", "N/A") if has_test else "N/A",
                     "recall@10": test_metrics.get("recall@10", "N/A") if has_test else "N/A",
                     "mrr": test_metrics.get("mrr", "N/A") if has_test else "N/A",
                     "similarity_gap": test_metrics.get("similarity_gap", "N/A") if has_test else "N/A"}
    save_training_results(hyperparams, final_metrics, a.results_csv)
    save_checkpoint(train_task, optimizer, config["num_epochs"] - 1, 0.0, config["output_dir"], is_final=True)      # :490-491
    print(f"done: {steps} steps in {time.time() - t0:.1f}s; results appended to {a.results_csv}")


class _AsyncLog:
    """Progress lines of the fast loop without stalling it: the step's eight figures (out8) are copied into a pinned host slot on
    the step's stream, an event marks the copy, and a line is printed once its event has passed -- a step or two later."""

    def __init__(self, device, slots: int = 8):
        self.host = torch.zeros(slots, 8, dtype=torch.float32).pin_memory()
        self.pending, self.slots, self.n = [], slots, 0

    def post(self, step, out8):
        if len(self.pending) >= self.slots:
            self.drain(wait=True)
        i = self.n % self.slots
        self.n += 1
        self.host[i].copy_(out8, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((step, i, ev))

    def drain(self, wait: bool = False):
        while self.pending and (wait or self.pending[0][2].query()):
            step, i, ev = self.pending.pop(0)
            ev.synchronize()
            loss, acc, pos, neg, gap = (float(x) for x in self.host[i, :5])
            print(f"step {step:5d}  loss {loss:.4f}  acc {acc:.4f}  pos {pos:.3f}  neg {neg:.3f}  z-gap {gap / max(abs(neg) + 1e-8, 1e-8):.2f}")


if __name__ == "__main__":
    main()
