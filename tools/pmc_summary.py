#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, per counter: mean per dispatch; plus the mean dispatch duration of each
kernel from the kernel-trace CSVs of the same passes (pseudo counter DURATION_NS)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]


def short(k):
    if "k_encode_dense" in k:
        return "k_encode_dense"
    if "k_encode_tiles" in k:
        return "k_encode_tiles"
    if "k_gather" in k:
        return "k_gather"
    if "k_assemble" in k:
        return "k_assemble"
    return None


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = short(row.get("Kernel_Name", ""))
        if k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*kernel_trace.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = short(row.get("Kernel_Name", ""))
        if k:
            acc[k]["DURATION_NS"].append(float(int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"  {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
