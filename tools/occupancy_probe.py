#!/usr/bin/env python3
"""How does k_encode_strips react to occupancy?  Same kernel, LDS image enlarged so that fewer workgroups fit per CU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ec504_imageencoder_amd import Mpeg1Encoder
W, H, n = 1920, 1080, 300
enc = Mpeg1Encoder(W, H, 12, "full", max_frames=n)
rgb = enc.synth(n)
for words in (1024, 4096, 6000, 24000):
    enc.debug_set_lds_words(words)
    lds = (1280 + 32 + 16 + 32 + 32 * 448 + words) * 4
    for _ in range(2):
        enc.encode(rgb)
    torch.cuda.synchronize()
    enc.profile(True)
    for _ in range(6):
        enc.encode(rgb)
    torch.cuda.synchronize()
    k, ms = enc.profile_read()
    enc.profile(False)
    print(f"image words {words:6d}  LDS/WG {lds/1024:6.1f} KiB  WGs/CU {int(160*1024//lds)}  kernel {ms/k*1e3:8.1f} us")
