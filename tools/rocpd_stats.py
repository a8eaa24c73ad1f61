#!/usr/bin/env python3
"""Kernel statistics (the columns of rocprofv3's kernel_stats.csv) from a rocprofv3 rocpd database.

    tools/rocpd_stats.py gpurun_out/final/prof1080/r_results.db [--last K] > profiles/<name>_kernel_stats.csv

--last K keeps only the last K dispatches of every kernel (the timed region of `bench.py --steps K`; the warm-up
launches before it run slower while the GPU's clocks ramp up).

rocprofv3 --kernel-trace --stats writes <prefix>_results.db by default on this image (csv only with
--output-format csv); the `kernels` view holds one row per dispatch with start/end in ns.
"""
import math
import sqlite3
import sys


def main(path, last=0):
    db = sqlite3.connect(path)
    rows = {}
    for name, dur, vgpr, lds, scratch in db.execute("select name, end - start, vgpr_count, lds_size, scratch_size from kernels order by start"):
        rows.setdefault(name, []).append((dur, vgpr, lds, scratch))
    if last > 0:
        rows = {k: v[-last:] for k, v in rows.items()}
    total = sum(d for v in rows.values() for d, *_ in v)
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev","LDSBytes","ScratchBytes"')
    for name, v in sorted(rows.items(), key=lambda kv: -sum(d for d, *_ in kv[1])):
        d = [x[0] for x in v]
        mean = sum(d) / len(d)
        sd = math.sqrt(sum((x - mean) ** 2 for x in d) / (len(d) - 1)) if len(d) > 1 else 0.0
        print(f'"{name}",{len(d)},{sum(d)},{mean:.3f},{100.0 * sum(d) / total:.4f},{min(d)},{max(d)},{sd:.3f},{v[0][2]},{v[0][3]}')


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0)
