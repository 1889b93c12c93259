#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (p_counter_collection.csv + p_kernel_trace.csv) per kernel.

usage: pmc_summary.py OUT.json DIR [DIR ...]   (each DIR one --pmc pass of the same command)

Per kernel: mean of every counter over the launches, mean duration from the pass's kernel trace,
register / LDS / scratch footprint, and derived figures for MFMA kernels:
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz).
"""
import collections
import csv
import json
import re
import sys
from pathlib import Path

SIMDS, CLK = 1024, 2.4e9


def short(name):
    if "at::native" in name or "rocclr" in name:
        return None
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = collections.defaultdict(dict)
    for d in dirs:
        d = Path(d)
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(next(d.rglob("*kernel_trace.csv")))):
            k = short(r["Kernel_Name"])
            if k:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(next(d.rglob("*counter_collection.csv")))):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            res[k].update(vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), lds_bytes=int(r["LDS_Block_Size"]),
                          scratch_bytes=int(r["Scratch_Size"]), grid_threads=int(r["Grid_Size"]), workgroup=int(r["Workgroup_Size"]))
        for k, cs in agg.items():
            res[k].setdefault("launches", len(next(iter(cs.values()))))
            res[k].setdefault("us_under_pmc", round(sum(dur[k]) / len(dur[k]), 2))
            for c, v in cs.items():
                res[k][c] = round(sum(v) / len(v))
    for k, r in res.items():
        cyc = SIMDS * r["us_under_pmc"] * 1e-6 * CLK
        if r.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            r["mfma_busy_frac"] = round(r["SQ_VALU_MFMA_BUSY_CYCLES"] / cyc, 3)
        if r.get("SQ_WAVE_CYCLES"):
            # SQ_*_CYCLES per-wave counters tick once per 4 clocks
            r["wave_time_waiting_frac"] = round(r.get("SQ_WAIT_ANY", 0) / r["SQ_WAVE_CYCLES"], 3)
            r["valu_issue_frac_of_simd_time"] = round(4 * r.get("SQ_ACTIVE_INST_VALU", 0) / cyc, 3)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: {x: r[x] for x in ("us_under_pmc", "mfma_busy_frac") if x in r} for k, r in res.items()}, indent=1))


if __name__ == "__main__":
    main()
