#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per dispatch for each nsa:: (and nsa_bf16::) kernel.
usage: pmc_summary.py <dir-with-*counter_collection.csv> [...]  -> prints CSV (kernel, counter, mean, n)"""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "nsa::" not in k and "nsa_bf16::" not in k:
                    continue
                k = k.split("(")[0].replace("void ", "")
                a = acc[(k, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "mean_per_dispatch", "dispatches"])
for (k, c), (s, n) in sorted(acc.items()):
    w.writerow([k, c, f"{s / n:.1f}", n])
