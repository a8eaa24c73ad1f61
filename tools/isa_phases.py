#!/usr/bin/env python3
"""Static VALU/SALU/LDS/VMEM instruction counts between the PHASE_MARK comments of k_encode_strips
(build: hipcc -DM1V_MARKS -S).  Slow-path (fp64) instructions are listed separately."""
import collections
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "ec504_imageencoder_amd", "csrc", "m1v_kernels.hip")
out = "/tmp/m1v_marks.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17",
                "-DM1V_MARKS", "-S", "--cuda-device-only", src, "-o", out] + os.environ.get("M1V_DEFS", "").split(),
               check=True, stderr=subprocess.DEVNULL)
pat = sys.argv[1] if len(sys.argv) > 1 else "k_encode_stripsILb1ELb0"
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
phase = "pre"
counts = collections.OrderedDict()
detail = collections.defaultdict(collections.Counter)
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.search(r"PHASE_MARK (\S+)", t)
    if m:
        phase = "after_" + m.group(1)
        continue
    if not l.startswith("\t") or not t or t[0] in ".;":
        continue
    op = t.split()[0]
    kind = ("dp" if re.search(r"_f64|f64_", op) else "valu" if op.startswith("v_") else "salu" if op.startswith("s_")
            else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
    counts.setdefault(phase, collections.Counter())[kind] += 1
    detail[phase][op] += 1
for ph, c in counts.items():
    print(f"{ph:14s} valu={c['valu']:5d} dp(slow path)={c['dp']:4d} salu={c['salu']:4d} lds={c['lds']:3d} vmem={c['vmem']:3d}")
if len(sys.argv) > 2:
    for op, n in detail[sys.argv[2]].most_common(40):
        print(f"   {op:28s}{n}")
