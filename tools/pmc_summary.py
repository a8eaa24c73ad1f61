#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, per counter: mean per dispatch."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "k_encode_dense" in k: k = "k_encode_dense"
        elif "k_encode_tiles" in k: k = "k_encode_tiles"
        elif "k_gather" in k: k = "k_gather"
        else: continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
