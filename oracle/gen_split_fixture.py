#!/usr/bin/env python3
"""Generates tests/golden/split_indices.json: the train / test membership AND order that the reference's test-mode loader gets from
`sklearn.model_selection.train_test_split(pairs_df, test_size=test_split, random_state=shuffle_seed)`
(src/towers/pairs/unified_bid_data_loader.py:1222-1226), for a few (n, test_size, seed) cases incl. the driver's own
(test_split 0.2, shuffle_seed 42: scripts/train.py:88-89).  Runs in the build container (scikit-learn is importable there); the
GPU box only sees the JSON.  The package's restatement (data_loader.sklearn_split_indices) is pinned against it in
tests/test_host_logic.py."""
import json
from pathlib import Path

import numpy as np
import sklearn
from sklearn.model_selection import train_test_split

CASES = [(10, 0.2, 42), (11, 0.2, 42), (1000, 0.2, 42), (997, 0.1, 7), (5, 0.5, 0), (3, 0.34, 123), (100_000, 0.2, 42)]


def main():
    out = {"sklearn_version": sklearn.__version__, "cases": []}
    for n, ts, seed in CASES:
        tr, te = train_test_split(np.arange(n), test_size=ts, random_state=seed)
        rec = {"n": n, "test_size": ts, "seed": seed, "n_train": int(len(tr)), "n_test": int(len(te))}
        if n <= 1000:
            rec["train"], rec["test"] = tr.tolist(), te.tolist()
        else:                                   # large case: head, tail and a checksum keep the fixture small
            rec["train_head"], rec["train_tail"] = tr[:16].tolist(), tr[-16:].tolist()
            rec["test_head"], rec["test_tail"] = te[:16].tolist(), te[-16:].tolist()
            w = np.arange(1, n + 1, dtype=np.int64)
            rec["train_checksum"] = int((tr.astype(np.int64) * w[:len(tr)]).sum() % (2 ** 61 - 1))
            rec["test_checksum"] = int((te.astype(np.int64) * w[:len(te)]).sum() % (2 ** 61 - 1))
        out["cases"].append(rec)
    p = Path(__file__).resolve().parents[1] / "tests" / "golden" / "split_indices.json"
    p.write_text(json.dumps(out))
    print("wrote", p, p.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
