#!/usr/bin/env python3
"""How does k_encode_dense react to occupancy?  Same kernel, LDS bit image enlarged so that fewer workgroups fit per CU
(96 VGPRs allow 5 four-wave workgroups; LDS per workgroup = 22,656 B at quality 12's default of 512 image words)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ec504_imageencoder_amd import Mpeg1Encoder
W, H, n = 1920, 1080, 300
enc = Mpeg1Encoder(W, H, 12, "full", max_frames=n)
rgb = enc.synth(n)
for _ in range(60):                      # past the clock ramp
    enc.encode(rgb)
for words in (512, 3060, 5100, 8530, 512):
    enc.debug_set_lds_words(words)
    lds = 22656 + (words - 512) * 4
    for _ in range(3):
        enc.encode(rgb)
    torch.cuda.synchronize()
    enc.profile(True)
    for _ in range(10):
        enc.encode(rgb)
    torch.cuda.synchronize()
    k, ms = enc.profile_read()
    enc.profile(False)
    print(f"image words {words:6d}  LDS/WG {lds/1024:6.1f} KiB  WGs/CU {min(5, int(160*1024//lds))}  kernel {ms/k*1e3:8.1f} us", flush=True)
