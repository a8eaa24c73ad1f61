#!/usr/bin/env python3
"""Kernel time of the aligned 3-channel fast path vs the general path (4 channels, width not a multiple of 8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ec504_imageencoder_amd import Mpeg1Encoder
n = 100
for W, H, C in ((1920, 1080, 3), (1920, 1080, 4), (1916, 1080, 3)):
    enc = Mpeg1Encoder(W, H, 12, "full", channels=C, max_frames=n)
    rgb = torch.randint(0, 256, (n, H, W, C), dtype=torch.uint8, device="cuda")
    for _ in range(30):
        enc.encode(rgb)
    torch.cuda.synchronize()
    enc.profile(True)
    for _ in range(20):
        enc.encode(rgb)
    torch.cuda.synchronize()
    k, ms = enc.profile_read()
    print(f"{W}x{H}x{C}: {ms / k * 1e3:8.1f} us per {n} frames  ->  {n / (ms / k) * 1e3:9.0f} frames/s (kernel only)", flush=True)
    enc.close()
