#!/usr/bin/env python3
"""Per-phase cycle shares of k_encode_strips from the diagnostic build (make -C ec504_imageencoder_amd/csrc stamps).
Reads SHARES, never the run time (the stamps serialise the phases)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ec504_imageencoder_amd import _ffi
_ffi.LIB_PATH = os.path.join(ROOT, "build", "libencoder_stamps.so")
import torch
from ec504_imageencoder_amd import Mpeg1Encoder

W, H, n = 1920, 1080, 300
enc = Mpeg1Encoder(W, H, 12, "full", max_frames=n)
rgb = enc.synth(n)
for _ in range(2):
    enc.encode(rgb)
torch.cuda.synchronize()
L = _ffi.lib()
buf = (C.c_ulonglong * 16)()
L.m1v_debug_read_stamps(enc._h, buf)
for _ in range(3):
    enc.encode(rgb)
L.m1v_debug_read_stamps(enc._h, buf)
names = ["0 prologue (tables, zero image, wait for pixels, barrier)", "1 -", "2 convert + FDCT + quantise + stage", "3 -", "4 DC hdr + emit set + pass 1",
         "5 workgroup scan (1 barrier)", "6 pass 2 (OR into image)", "7 barrier before the store", "8 store run image"]
tot = sum(buf[i] for i in range(9))
for i, nm in enumerate(names):
    print(f"{nm:40s} {buf[i] / tot * 100:6.2f} %   {buf[i] / (3 * (300 * 189 * 4)):10.0f} cycles/wave")
