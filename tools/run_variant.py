#!/usr/bin/env python3
"""Run a few encode steps of a library VARIANT (build/libencoder_<name>.so, tools/mkvariant.sh; "base" = the in-tree
library) so that rocprofv3 can be wrapped around it:  rocprofv3 --pmc ... -- python3 tools/run_variant.py v2 --steps 3"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("name")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--n", type=int, default=300)
ap.add_argument("--path", default="", help="runs | tiles (default: what the library picks)")
a = ap.parse_args()
from ec504_imageencoder_amd import _ffi
if a.name != "base":
    _ffi.LIB_PATH = os.path.join(ROOT, "build", f"libencoder_{a.name}.so")
import torch
from ec504_imageencoder_amd import Mpeg1Encoder

enc = Mpeg1Encoder(a.w, a.h, 12, "full", max_frames=a.n)
if a.path:
    enc.debug_set_path(a.path)
rgb = enc.synth(a.n)
for _ in range(a.steps):
    enc.encode(rgb)
torch.cuda.synchronize()
print("ok", a.name)
