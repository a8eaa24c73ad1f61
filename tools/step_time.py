#!/usr/bin/env python3
"""Whole-step time (encode + layout + gather, stream-ordered) of library variants, one after the other in this
process, interleaved rounds:  tools/step_time.py base gs2 gs4     (base = the in-tree libencoder.so)"""
import ctypes as C
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch

vp = C.c_void_p
W, H, n = 1920, 1080, 300
libs = {}
for nm in sys.argv[1:]:                     # name[:path]  (path = runs | tiles, default: the library's choice)
    lib_nm, _, which = nm.partition(":")
    path = os.path.join(ROOT, "ec504_imageencoder_amd", "libencoder.so") if lib_nm == "base" else os.path.join(ROOT, "build", f"libencoder_{lib_nm}.so")
    L = C.CDLL(path)
    L.m1v_create.argtypes = [C.POINTER(vp)] + [C.c_int] * 7
    L.m1v_encode_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_size_t, vp, vp, vp, vp]
    L.m1v_synth_device.argtypes = [vp, C.c_size_t, C.c_int, C.c_uint64, C.c_uint64, vp]
    L.m1v_frame_bound.argtypes = [vp]
    L.m1v_frame_bound.restype = C.c_size_t
    h = vp()
    assert L.m1v_create(C.byref(h), 0, W, H, 3, 12, 1, n) == 0
    if which:
        L.m1v_debug_set_path.argtypes = [vp, C.c_int]
        assert L.m1v_debug_set_path(h, {"runs": 0, "tiles": 1}[which]) == 0
    libs[nm] = (L, h)
dev = torch.device("cuda", 0)
rgb = torch.empty(n * H * W * 3, dtype=torch.uint8, device=dev)
L0, _ = next(iter(libs.values()))
L0.m1v_synth_device(rgb.data_ptr(), H * W * 3, n, 504, 0, None)
cap = 200 * 1024 * n
out = torch.empty(cap, dtype=torch.uint8, device=dev)
sizes = torch.empty(n, dtype=torch.int64, device=dev)
meta = torch.zeros(2, dtype=torch.int64, device=dev)
ref = None
res = {k: [] for k in libs}


def run(L, h, steps):
    for _ in range(steps):
        L.m1v_encode_device(h, rgb.data_ptr(), n, 0, out.data_ptr(), cap, sizes.data_ptr(), meta.data_ptr(), meta.data_ptr() + 8, None)


for nm, (L, h) in libs.items():          # correctness of every variant against the first one + clock ramp
    run(L, h, 40)
    torch.cuda.synchronize()
    total = int(meta[0].item())
    blob = out[:total].cpu()
    if ref is None:
        ref = blob
    assert torch.equal(blob, ref), nm
for rnd in range(7):
    for nm, (L, h) in libs.items():
        run(L, h, 5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(L, h, 50)
        torch.cuda.synchronize()
        res[nm].append((time.perf_counter() - t0) / 50 * 1e6)
for nm, v in res.items():
    print(f"{nm:10s} median {statistics.median(v):8.1f} us/step   min {min(v):8.1f}")
