#!/usr/bin/env python3
"""Per-kernel means of a `rocprofv3 --pmc ... --output-format csv` run: reads every
*counter_collection.csv under a directory and prints kernel,counter,dispatches,mean_per_dispatch."""
import csv, glob, os, sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            key = (row["Kernel_Name"], row["Counter_Name"])
            acc[key][0] += 1
            acc[key][1] += float(row["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
for (k, c), (n, tot) in sorted(acc.items()):
    w.writerow([k, c, n, tot / n])
