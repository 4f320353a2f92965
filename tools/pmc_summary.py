#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per launch for every sa:: kernel (dev tool)."""
import csv
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if "sa::" not in n:
            continue
        n = re.sub(r"\(.*$", "", n.replace("void ", ""))[:70]
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n)
    wc = sum(cs.get("SQ_WAVE_CYCLES", [0])) or 1.0
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}  /WAVE_CYCLES={sum(v)/wc:7.3f}")
