#!/usr/bin/env python3
"""Print per-kernel average durations (us) from a rocprofv3 *_kernel_stats.csv (short names)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Name"])
    name = m.group(1) if m else r["Name"][:50]
    if "at::native" in r["Name"] or (pat and not re.search(pat, name)):
        continue
    print(f"{name:60s} {int(r['Calls']):5d} {float(r['AverageNs']) / 1e3:8.2f}")
