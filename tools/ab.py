#!/usr/bin/env python3
"""In-process A/B of library variants (build/libencoder_<name>.so): interleaved rounds on ONE device,
kernel time of the encode kernel from the library's HIP events.  usage: ab.py name1 name2 ... [--w W --h H --n N]
name = variant[:T[:W[:path]]], e.g. base:::runs base:::tiles"""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--n", type=int, default=300)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--nocheck", action="store_true")
a = ap.parse_args()
import torch

vp = C.c_void_p
libs = {}
for nm in a.names:
    parts = nm.split(":")                      # "name[:T[:W[:path]]]" = library variant, run length T, LDS image words W, path (runs | tiles)
    lib_nm, dense_t, lds_w = parts[0], (parts[1] if len(parts) > 1 else ""), (parts[2] if len(parts) > 2 else "")
    which = parts[3] if len(parts) > 3 else ""
    path = os.path.join(ROOT, "ec504_imageencoder_amd", "libencoder.so") if lib_nm == "base" else os.path.join(ROOT, "build", f"libencoder_{lib_nm}.so")
    L = C.CDLL(path)
    L.m1v_create.argtypes = [C.POINTER(vp)] + [C.c_int] * 7
    L.m1v_encode_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_size_t, vp, vp, vp, vp]
    L.m1v_synth_device.argtypes = [vp, C.c_size_t, C.c_int, C.c_uint64, C.c_uint64, vp]
    L.m1v_profile_enable.argtypes = [vp, C.c_int]
    L.m1v_profile_read.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.m1v_last_error.restype = C.c_char_p
    h = vp()
    rc = L.m1v_create(C.byref(h), 0, a.w, a.h, 3, 12, 1, a.n)
    assert rc == 0, L.m1v_last_error()
    if dense_t:
        L.m1v_debug_set_dense_threads.argtypes = [vp, C.c_int]
        assert L.m1v_debug_set_dense_threads(h, int(dense_t)) == 0, L.m1v_last_error()
    if lds_w:
        L.m1v_debug_set_lds_words.argtypes = [vp, C.c_int]
        L.m1v_debug_set_lds_words(h, int(lds_w))
    if which:
        L.m1v_debug_set_path.argtypes = [vp, C.c_int]
        assert L.m1v_debug_set_path(h, {"runs": 0, "tiles": 1}[which]) == 0, L.m1v_last_error()
    libs[nm] = (L, h)
rgb = torch.empty((a.n, a.h, a.w, 3), dtype=torch.uint8, device="cuda")
L0, h0 = libs[a.names[0]]
L0.m1v_synth_device(rgb.data_ptr(), a.w * a.h * 3, a.n, 504, 0, None)
out = torch.empty(a.n * (a.w * a.h // 2 + 4096), dtype=torch.uint8, device="cuda")
outs = {}
sizes = torch.empty(a.n, dtype=torch.int64, device="cuda")
meta = torch.zeros(2, dtype=torch.int64, device="cuda")
times = {nm: [] for nm in a.names}
ref = None
for r in range(a.rounds + 1):
    # alternate the order from round to round: within a round the later variants run on warmer clocks
    for nm in (a.names if r % 2 == 0 else list(reversed(a.names))):
        L, h = libs[nm]
        L.m1v_profile_enable(h, 1)
        for _ in range(a.reps):
            rc = L.m1v_encode_device(h, rgb.data_ptr(), a.n, 0, out.data_ptr(), out.numel(), sizes.data_ptr(), meta.data_ptr(), meta.data_ptr() + 8, None)
            assert rc == 0
        torch.cuda.synchronize()
        n, ms = C.c_int(), C.c_double()
        L.m1v_profile_read(h, C.byref(n), C.byref(ms))
        if r > 0:
            times[nm].append(ms.value / n.value)
        if r == 0:
            tot = int(meta[0].item())
            digest = hash(out[:tot].cpu().numpy().tobytes())
            if ref is None:
                ref = (tot, digest)
            assert a.nocheck or (tot, digest) == ref, f"{nm}: output differs from {a.names[0]}"
base = statistics.median(times[a.names[0]])
for nm in a.names:
    t = times[nm]
    med = statistics.median(t)
    print(f"{nm:24s} median {med*1e3:8.1f} us  min {min(t)*1e3:8.1f} us  x{base/med:5.3f}  fps(kernel only) {a.n/med*1e3:9.0f}")
