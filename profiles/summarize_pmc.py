#!/usr/bin/env python3
"""Per-kernel means of a `rocprofv3 --pmc ... --output-format csv` run: reads every
*counter_collection.csv under a directory and prints kernel,kernel_key,counter,dispatches,mean_per_dispatch
(kernel_key = function<template,arguments>, the form bench.py looks kernels up by: kernel_names.py)."""
import csv, glob, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import kernel_key

acc = defaultdict(lambda: [0, 0.0])
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            key = (row["Kernel_Name"], row["Counter_Name"])
            acc[key][0] += 1
            acc[key][1] += float(row["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "kernel_key", "counter", "dispatches", "mean_per_dispatch"])
for (k, c), (n, tot) in sorted(acc.items()):
    w.writerow([k, kernel_key(k), c, n, tot / n])
