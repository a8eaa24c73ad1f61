#!/usr/bin/env python3
"""Static instruction mix of one kernel from `hipcc -S --cuda-device-only` output.
usage: isa_mix.py file.s substring-of-kernel-symbol"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ins = []
for l in lines[start + 1:end]:
    t = l.strip()
    if not l.startswith("\t") or not t or t[0] in ".;":
        continue
    ins.append(t.split()[0])
c = collections.Counter(ins)
groups = collections.Counter()
for k, v in c.items():
    g = ("valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_")
         else "vmem" if k.startswith(("global_", "buffer_", "scratch_", "flat_")) else "other")
    groups[g] += v
print(len(ins), dict(groups))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"  {k:30s}{v}")
