#!/usr/bin/env python3
"""Static instruction table of the two encode kernels (common path vs the basic blocks of the fp64 tie path), by opcode
class, next to the dynamic totals of the committed PMC record: where the vector-ALU budget of a block goes.
    python tools/valu_table.py > profiles/r03_valu_table.txt"""
import collections
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "ec504_imageencoder_amd", "csrc", "m1v_kernels.hip")
out = "/tmp/m1v_valu_table.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17",
                "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
text = open(out).read()

CLASSES = [
    ("byte -> float (v_cvt_f32_ubyte*)", r"^v_cvt_f32_ubyte"),
    ("float multiply-add family (fma, fmac, fmamk, fmaak, mul_f32)", r"^v_(fma|fmac|fmamk|fmaak|mul)_f32"),
    ("float add / sub", r"^v_(add|sub|subrev)_f32"),
    ("v_floor_f32", r"^v_floor_f32"),
    ("float <-> int conversions (cvt_i32_f32, cvt_f32_i32/u32)", r"^v_cvt_(i32_f32|f32_i32|f32_u32|u32_f32)"),
    ("f16 pack / unpack (cvt_pkrtz, cvt_f32_f16)", r"^v_cvt_(pkrtz_f16_f32|f32_f16)"),
    ("v_min3_f32 / v_max3", r"^v_(min3|max3)_"),
    ("32-bit integer multiply (v_mul_lo_u32, mul_hi, mad_u64)", r"^v_(mul_lo_u32|mul_hi_|mad_u64_u32|mad_i64)"),
    ("compares", r"^v_cmp"),
    ("bit logic (and, or, xor, not, bfe, bfi, perm, bitop3, and_or, or3, alignbit)", r"^v_(and|or|xor|not|bfe|bfi|perm|bitop3|and_or|or3|alignbit|alignbyte)"),
    ("shifts", r"^v_(lshl|lshr|ashr)"),
    ("integer add / sub / min / max / ffbl / mbcnt", r"^v_(add|sub|subrev|add3|min|max|ffbl|mbcnt|addc|subb|lshl_add|mad_u32|mul_u32_u24|mad_u32_u24)"),
    ("moves, selects, lane ops (mov, cndmask, readlane, readfirstlane, dpp, swap)", r"^v_(mov|cndmask|readlane|readfirstlane|writelane|swap|permlane|accvgpr)"),
    ("fp64 (tie path)", r"f64"),
]


def kernel_lines(mangled):
    a = text.index(mangled + ":")
    b = text.index(".Lfunc_end", a)
    return text[a:b].splitlines()[1:]


def table(mangled, title):
    blocks, cur = [], []
    for l in kernel_lines(mangled):
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur)
            cur = []
        elif l.startswith("\t") and t and t[0] not in ".;":
            cur.append(t.split()[0])
    blocks.append(cur)
    fast, slow, other = collections.Counter(), collections.Counter(), collections.Counter()
    for bl in blocks:
        tie = any("f64" in op for op in bl)
        for op in bl:
            if op.startswith("v_"):
                for name, pat in reversed(CLASSES):
                    if re.search(pat, op):
                        break
                else:
                    name = "other vector"
                for nm, pat in CLASSES:     # first match wins, fp64 checked first
                    if re.search(pat, op) and (nm != "integer add / sub / min / max / ffbl / mbcnt" or not re.search(r"_f32", op)):
                        name = nm
                        break
                if "f64" in op:
                    name = "fp64 (tie path)"
                (slow if tie else fast)[name] += 1
            else:
                kind = "scalar" if op.startswith("s_") else "LDS" if op.startswith("ds_") else "vector memory" if op.startswith(("global_", "scratch_", "buffer_", "flat_")) else "other"
                other[kind + (" (tie-path blocks)" if tie else "")] += 1
    print(f"== {title}")
    print(f"   {'class':78s} {'common path':>12s} {'tie-path blocks':>16s}")
    for name in [c[0] for c in CLASSES] + ["other vector"]:
        if fast[name] or slow[name]:
            print(f"   {name:78s} {fast[name]:12d} {slow[name]:16d}")
    print(f"   {'vector instructions, static':78s} {sum(fast.values()):12d} {sum(slow.values()):16d}")
    print("   other instruction kinds (static): " + ", ".join(f"{k} {v}" for k, v in sorted(other.items())))
    print()


print("# Static instruction mix of the encode kernels as compiled for gfx950 (tools/valu_table.py), and the dynamic totals of the\n"
      "# committed PMC passes.  A wave executes the common path once (64 blocks), pass 1's loop body once per coded coefficient of its\n"
      "# longest lane, and the basic blocks of the tie path only where a lane's colour sum is within 3e-4 of an integer.\n")
table("_ZN12_GLOBAL__N_114k_encode_denseILi1ELb1EEEvNS_9DenseArgsE", "k_encode_dense<1, true> (run kernel, aligned 3-channel input)")
table("_ZN12_GLOBAL__N_114k_encode_tilesILb1ELi2EEEvNS_8TileArgsE", "k_encode_tiles<true, 2> (tile kernel)")
try:
    doc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc.json")))
    print("== dynamic (rocprofv3 --pmc, 300 x 1920x1080 and 300 x 3840x2160, per launch)")
    for r in doc["workloads"]:
        waves = r["waves"]
        print(f"   {r['kernel']:15s} {r['width']}x{r['height']}: {r['valu']['insts_per_launch']:>11d} vector instructions = "
              f"{r['valu']['insts_per_launch'] / waves:7.0f} per wave (64 blocks), {r['kernel_us_under_profiler']:8.1f} us under the profiler, "
              f"L1->L2 read requests {r['l1_to_l2_read_requests'] / r['pixel_lines_128B']:.2f} per 128-byte line of pixels")
except (OSError, KeyError):
    pass
