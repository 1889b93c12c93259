#!/usr/bin/env python3
"""Micro-benchmark of tt_embed_lookup_fwd on the bench workload (both towers, 38 keys, B=8192).
Between two timed launches a 512 MiB streaming copy evicts L2 / Infinity Cache, as the other kernels of a
training step do.  Prints mean/min kernel time (HIP events, back-to-back queue) and algorithmic GB/s."""
import argparse, json, sys, tempfile
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import jodalrob_twotower_amd as tt
from jodalrob_twotower_amd import ops, synthetic


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--out-dtype", default="f32")
    ap.add_argument("--no-flush", action="store_true")
    ap.add_argument("--zipf", type=float, default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    schema = synthetic.load_real_schema(ROOT / "jodalrob-twotower_amd" / "schema_real.json")
    kn, kc = schema["notice"]["categorical"], schema["company"]["categorical"]
    vn = synthetic.scale_vocabs(schema["notice"]["vocab_sizes"], a.rows)
    vc = synthetic.scale_vocabs(schema["company"]["vocab_sizes"], a.rows)
    E, B = 32, a.batch
    R = sum(vn) + sum(vc)
    table = torch.randn((R, E), device=dev)
    offs_n = torch.tensor([sum(vn[:i]) for i in range(len(vn))], dtype=torch.int64, device=dev)
    offs_c = torch.tensor([sum(vn) + sum(vc[:i]) for i in range(len(vc))], dtype=torch.int64, device=dev)
    voc_n, voc_c = torch.tensor(vn, dtype=torch.int64, device=dev), torch.tensor(vc, dtype=torch.int64, device=dev)
    odt = torch.float32 if a.out_dtype == "f32" else torch.bfloat16
    xn = torch.empty((B, 128 + len(kn) * E), dtype=odt, device=dev)
    xc = torch.empty((B, 128 + len(kc) * E), dtype=odt, device=dev)
    pool = [synthetic.make_batch(B, vn, vc, kn, kc, 4, 4, dev, seed=100 + i, zipf_alpha=a.zipf) for i in range(8)]
    flush_src = torch.empty(128 << 20, dtype=torch.float32, device=dev)
    flush_dst = torch.empty_like(flush_src)
    evs = []
    for i in range(a.iters + 5):
        b = pool[i % 8]
        sides = [ops.LookupSide(b["notice"]["kjt"].values(), offs_n, voc_n, xn[:, 128:], len(kn)),
                 ops.LookupSide(b["company"]["kjt"].values(), offs_c, voc_c, xc[:, 128:], len(kc))]
        if not a.no_flush:
            flush_dst.copy_(flush_src)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.embed_lookup(table, sides, B, want_rows=True)
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = [s.elapsed_time(e) * 1e3 for s, e in evs[5:]]
    K = len(kn) + len(kc)
    osz = 4 if a.out_dtype == "f32" else 2
    nbytes = B * K * (E * 4 + 8 + E * osz)
    mean, mn = sum(ts) / len(ts), min(ts)
    print(json.dumps({"mean_us": round(mean, 2), "min_us": round(mn, 2), "algo_GBps_mean": round(nbytes / mean / 1e3, 1),
                      "algo_GBps_min": round(nbytes / mn / 1e3, 1), "bytes": nbytes, "flush": not a.no_flush, "out": a.out_dtype}))


if __name__ == "__main__":
    main()
