#!/usr/bin/env python3
"""Per-step wall time of the first steps after start-up (clock ramp / cold effects): tools/step_ramp.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ec504_imageencoder_amd import Mpeg1Encoder

n, W, H = 300, 1920, 1080
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
enc = Mpeg1Encoder(W, H, 12, "full", max_frames=n)
dev = torch.device("cuda", 0)
rgb = enc.synth(n, seed=504, device=dev)
out = torch.empty(enc.default_out_capacity(n), dtype=torch.uint8, device=dev)
sizes = torch.empty(n, dtype=torch.int64, device=dev)
meta = torch.zeros(2, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ts = []
for i in range(steps):
    t0 = time.perf_counter()
    enc.encode(rgb, 0, out=out, sizes=sizes, meta=meta)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms (synchronised each step):", " ".join(f"{t:.3f}" for t in ts))
