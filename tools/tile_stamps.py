#!/usr/bin/env python3
"""Per-phase cycles of a wave of k_encode_tiles, from a diagnostic build (tools/mkvariant.sh NAME -DM1V_TILE_STAMPS ...):
usage: tile_stamps.py NAME.  Reads SHARES and cycles per wave; the stamps themselves perturb the schedule a little."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ec504_imageencoder_amd import _ffi
_ffi.LIB_PATH = os.path.join(ROOT, "build", f"libencoder_{sys.argv[1]}.so")
import torch
from ec504_imageencoder_amd import Mpeg1Encoder

W, H, n = 1920, 1080, 300
enc = Mpeg1Encoder(W, H, 12, "full", max_frames=n)
rgb = enc.synth(n)
for _ in range(3):
    enc.encode(rgb)
torch.cuda.synchronize()
L = _ffi.lib()
buf = (C.c_ulonglong * 32)()
L.m1v_debug_read_stamps(enc._h, buf)
reps = 5
for _ in range(reps):
    enc.encode(rgb)
L.m1v_debug_read_stamps(enc._h, buf)
names = ["0 geometry + DMA issue + image clear", "1 wait: VLC table landed + barrier", "2 rows: ring waits/reads, convert, row pass",
         "3 column pass + quantise + stage + mask", "4 DC header + pass 1", "5 barrier (bit counts)", "6 scan + segment table",
         "7 pass 2 (OR into image)", "8 barrier (image complete)", "9 store + exit"]
tiles = n * 15 * 17
for kind, base, waves in (("luma waves", 0, 2 * tiles * reps), ("chroma wave", 12, tiles * reps)):
    tot = sum(buf[base + i] for i in range(10))
    print(f"{kind}: {tot / waves:.0f} cycles per wave between the first and the last stamp")
    for i, nm in enumerate(names):
        print(f"   {nm:46s} {buf[base + i] / tot * 100:6.2f} %   {buf[base + i] / waves:8.0f} cycles")
