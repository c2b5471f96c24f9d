#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one row per dispatch and counter) as per-kernel averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0][:60]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = []
for name, ctrs in sorted(acc.items()):
    if not any(k in name for k in ("pow2", "small_lattice", "generic", "finalize")):
        continue
    lines.append(name)
    for c, vals in sorted(ctrs.items()):
        lines.append(f"    {c:28s} calls={len(vals):4d} avg={sum(vals) / len(vals):.6g}")
text = "\n".join(lines)
print(text)
open(os.path.join(out, "summary.txt"), "w").write(text + "\n")
