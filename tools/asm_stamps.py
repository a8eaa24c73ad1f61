#!/usr/bin/env python3
"""Timeline of the workgroups of one k_assemble launch from a diagnostic build (tools/mkvariant.sh NAME -DM1V_ASM_STAMPS):
s_memrealtime (100 MHz) at the start of every workgroup and behind each of its phases.  usage: asm_stamps.py NAME [W H N]."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ec504_imageencoder_amd import _ffi
_ffi.LIB_PATH = os.path.join(ROOT, "build", f"libencoder_{sys.argv[1]}.so")
import torch
from ec504_imageencoder_amd import Mpeg1Encoder

W, H, n = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (1920, 1080, 300)
enc = Mpeg1Encoder(W, H, 12, "full", max_frames=n)
rgb = enc.synth(n)
for _ in range(5):
    enc.encode(rgb)
torch.cuda.synchronize()
L = C.CDLL(_ffi.LIB_PATH)
groups = 8 if W <= 1920 else 4
wgs = n * ((W // 16 + groups - 1) // groups)
buf = np.zeros((wgs, 8), dtype=np.uint64)
L.m1v_debug_read_timeline(C.c_void_p(enc._h.value if hasattr(enc._h, "value") else enc._h), buf.ctypes.data_as(C.c_void_p), C.c_int(wgs))
t = buf.astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
names = ["start", "0 counter/segment loads + reductions", "1 housekeeping", "2 image clear + segment scan", "3 source loads + OR", "4 barrier",
         "5 stores issued", "6 stores drained"]
print(f"{len(t)} workgroups; launch spans {us.max():.2f} us from the first start to the last end")
print(f"workgroup starts: median {np.median(us[:, 0]):.2f} us, 90 % {np.percentile(us[:, 0], 90):.2f}, last {us[:, 0].max():.2f}")
life = us[:, 7] - us[:, 0]
print(f"workgroup lifetime: median {np.median(life):.2f} us, 90 % {np.percentile(life, 90):.2f}, max {life.max():.2f}")
for i in range(1, 8):
    d = us[:, i] - us[:, i - 1]
    print(f"   {names[i]:44s} median {np.median(d):6.2f} us   mean {d.mean():6.2f}   90 % {np.percentile(d, 90):6.2f}")
# how many workgroups are alive over time
ev = np.concatenate([np.stack([us[:, 0], np.ones(len(t))], 1), np.stack([us[:, 7], -np.ones(len(t))], 1)])
ev = ev[np.argsort(ev[:, 0])]
alive = np.cumsum(ev[:, 1])
for q in (0.1, 0.25, 0.5, 0.75, 0.9):
    k = int(q * (len(ev) - 1))
    print(f"   alive at {ev[k, 0]:6.2f} us: {int(alive[k])}")
