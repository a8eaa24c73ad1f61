#!/usr/bin/env python3
"""Writes profiles/r02_pmc.json, the committed PMC record bench.py reads for `roofline.traffic` and `roofline.valu`.

    python tools/pmc_record.py gpurun_out/pmc_<tag1080p> [gpurun_out/pmc_<tag4k> ...]

Each directory holds the per-group CSVs of tools/pmc.sh for ONE workload (its bench.log names width/height/frames).
From the counters of k_encode_dense:
  fetch_size_kib, write_size_kib   per launch (FETCH_SIZE / WRITE_SIZE are in KiB)
  valu.insts_per_launch            SQ_INSTS_VALU
  valu.clock_ghz                   GRBM_GUI_ACTIVE / 8 XCDs / kernel duration of the same pass (kernel-trace CSV)
and from a measurement of the kernel's own instruction stream (tools/ubench/gen_real_stream.py -> profiles/r02_real_stream.txt:
the vector instructions of the compiled pixel stage, same order / registers / dependencies, looped without memory, 5 waves/SIMD):
  valu.ns_per_inst_per_simd        what ONE vector instruction of this stream costs a SIMD when nothing else is in the way
                                   (half-rate classes run beside float ops of other instructions: the cost is per
                                   instruction, ~2.15 cycles at 2 GHz, not the sum of class costs round 2 first assumed)
  valu.static_mix_common_path      for the record: full-rate / half-rate instruction counts of the common path"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL_RATE = re.compile(r"^v_(add|sub|subrev)_(u32|f32|co_u32)|^v_(and|or|xor|not)_b32|^v_(lshlrev|lshrrev|ashrrev)_(b32|i32)"
                       r"|^v_mov_b32|^v_(mul|fma|fmac|fmamk|fmaak|mac)_f32|^v_accvgpr")


def issue_cost_of_common_path(kernel="k_encode_denseILi1ELb1"):
    src = os.path.join(ROOT, "ec504_imageencoder_amd", "csrc", "m1v_kernels.hip")
    out = "/tmp/m1v_pmc_record.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize",
                    "-std=c++17", "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(kernel) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], []
    for l in lines[start + 1:end]:
        if re.match(r"^\.LBB\S+:", l):
            blocks.append(cur)
            cur = []
        elif l.startswith("\t") and l.strip() and l.strip()[0] not in ".;":
            cur.append(l.strip().split()[0])
    blocks.append(cur)
    n = cyc = 0
    mix = collections.Counter()
    for b in blocks:
        if any("f64" in op for op in b):
            continue
        for op in b:
            if op.startswith("v_"):
                c = 2 if FULL_RATE.match(op) else 4
                n += 1
                cyc += c
                mix[c] += 1
    return cyc / n, {"full_rate_2_cycles": mix[2], "half_rate_4_cycles": mix[4]}


def stream_cost_ns():
    """profiles/r02_real_stream.txt, line of 5 waves/SIMD: cycles at 2.0 GHz per instruction -> ns"""
    for l in open(os.path.join(ROOT, "profiles", "r02_real_stream.txt")):
        m = re.search(r"as compiled\s+5 waves/SIMD\s+\d+ instr\s+[\d.]+ cycles@2GHz per pass\s+([\d.]+) per instr", l)
        if m:
            return round(float(m.group(1)) / 2.0, 4)
    raise SystemExit("profiles/r02_real_stream.txt: no 5-wave line")


def read_dir(d):
    acc = collections.defaultdict(list)
    dur = []
    for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if "k_encode_dense" in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*kernel_trace.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if "k_encode_dense" in row.get("Kernel_Name", ""):
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9)
    mean = {k: sum(v) / len(v) for k, v in acc.items()}
    geo = None
    for f in glob.glob(os.path.join(d, "*.log")):
        m = re.search(r"(\d+) x (\d+)x(\d+) synthetic", open(f, errors="replace").read())
        if m:
            geo = (int(m.group(2)), int(m.group(3)), int(m.group(1)))
            break
    return mean, (sum(dur) / len(dur) if dur else None), geo


def main():
    cost, mix = issue_cost_of_common_path()
    recs = []
    for d in sys.argv[1:]:
        mean, dur, geo = read_dir(d)
        if geo is None:
            raise SystemExit(f"{d}: no bench log with the workload geometry")
        W, H, n = geo
        rec = {"width": W, "height": H, "frames": n, "fetch_size_kib": mean["FETCH_SIZE"], "write_size_kib": mean["WRITE_SIZE"],
               "kernel_us_under_profiler": round(dur * 1e6, 1), "source_dir": os.path.basename(d.rstrip("/"))}
        if "SQ_INSTS_VALU" in mean and "GRBM_GUI_ACTIVE" in mean:
            rec["valu"] = {"insts_per_launch": int(mean["SQ_INSTS_VALU"]), "ns_per_inst_per_simd": stream_cost_ns(), "simds": 1024,
                           "clock_ghz": round(mean["GRBM_GUI_ACTIVE"] / 8 / dur / 1e9, 3), "static_mix_common_path": mix,
                           "source": "rocprofv3 --pmc SQ_INSTS_VALU on the shipped kernel; cost per instruction from the kernel's own "
                                     "instruction stream looped without memory at 5 waves per SIMD (profiles/r02_real_stream.txt)"}
        recs.append(rec)
    # GRBM_GUI_ACTIVE / 8 / duration reads high on dispatches well below a millisecond (MI355X_MICROARCH.md, DVFS give-back):
    # take the shader clock of this kernel from the longest dispatch measured and use it for every workload
    with_clock = [r for r in recs if "valu" in r]
    if with_clock:
        longest = max(with_clock, key=lambda r: r["kernel_us_under_profiler"])
        for r in with_clock:
            r["valu"]["clock_ghz_of_this_dispatch"] = r["valu"]["clock_ghz"]
            r["valu"]["clock_ghz"] = longest["valu"]["clock_ghz"]
    out = os.path.join(ROOT, "profiles", "r02_pmc.json")
    json.dump({"kernel": "k_encode_dense<1, true>", "workloads": recs}, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
