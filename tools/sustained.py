#!/usr/bin/env python3
"""Sustained A/B of library variants and paths on ONE device: every variant runs `--settle` back-to-back launches untimed (the
package power controller needs ~0.3 s to pull the shader clock down to what 1400 W allow) and then `--launches` timed ones
(wall time per step = encode kernel + layout + gather, one synchronisation at the end; no events), variants alternating for `--rounds` rounds.  tools/ab.py times bursts of five
launches between variant switches and therefore sees higher clocks: kernels that differ in power draw rank differently there.
Socket power and shader clock are sampled from the amdgpu hwmon files during the timed part when they are readable.
    usage: sustained.py name[:path] ...      name = base | build/libencoder_<name>.so variant, path = runs | tiles"""
import argparse
import ctypes as C
import glob
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="+")
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--n", type=int, default=300)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--settle", type=int, default=500)
ap.add_argument("--launches", type=int, default=500)
ap.add_argument("--nocheck", action="store_true", help="timing builds: do not compare the outputs")
a = ap.parse_args()
import torch

vp = C.c_void_p


def hwmon():
    for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        p = [os.path.join(d, f) for f in ("power1_average", "power1_input")]
        p = [x for x in p if os.path.exists(x)]
        f = os.path.join(d, "freq1_input")
        if p and os.path.exists(f):
            return p[0], f
    return None, None


POWER, FREQ = hwmon() if os.environ.get("SUSTAINED_HWMON") else (None, None)   # reading the SMU files stalls submissions: off by default


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop, self.w, self.mhz = False, [], []

    def run(self):
        while not self.stop and POWER:
            try:
                self.w.append(int(open(POWER).read()) / 1e6)
                self.mhz.append(int(open(FREQ).read()) / 1e6)
            except (OSError, ValueError):
                pass
            time.sleep(0.02)


libs = {}
for nm in a.names:
    lib_nm, _, which = nm.partition(":")
    path = os.path.join(ROOT, "ec504_imageencoder_amd", "libencoder.so") if lib_nm == "base" else os.path.join(ROOT, "build", f"libencoder_{lib_nm}.so")
    L = C.CDLL(path)
    L.m1v_create.argtypes = [C.POINTER(vp)] + [C.c_int] * 7
    L.m1v_encode_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_size_t, vp, vp, vp, vp]
    L.m1v_synth_device.argtypes = [vp, C.c_size_t, C.c_int, C.c_uint64, C.c_uint64, vp]
    L.m1v_profile_enable.argtypes = [vp, C.c_int]
    L.m1v_profile_read.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.m1v_last_error.restype = C.c_char_p
    h = vp()
    assert L.m1v_create(C.byref(h), 0, a.w, a.h, 3, 12, 1, a.n) == 0, L.m1v_last_error()
    if which:
        L.m1v_debug_set_path.argtypes = [vp, C.c_int]
        assert L.m1v_debug_set_path(h, {"runs": 0, "tiles": 1}[which]) == 0, L.m1v_last_error()
    libs[nm] = (L, h)
rgb = torch.empty((a.n, a.h, a.w, 3), dtype=torch.uint8, device="cuda")
L0, h0 = libs[a.names[0]]
L0.m1v_synth_device(rgb.data_ptr(), a.w * a.h * 3, a.n, 504, 0, None)
out = torch.empty(a.n * (a.w * a.h // 2 + 4096), dtype=torch.uint8, device="cuda")
sizes = torch.empty(a.n, dtype=torch.int64, device="cuda")
meta = torch.zeros(2, dtype=torch.int64, device="cuda")
res = {nm: [] for nm in a.names}
ref = None
for r in range(a.rounds):
    for nm in (a.names if r % 2 == 0 else list(reversed(a.names))):
        L, h = libs[nm]

        def go(k):
            for _ in range(k):
                assert L.m1v_encode_device(h, rgb.data_ptr(), a.n, 0, out.data_ptr(), out.numel(), sizes.data_ptr(), meta.data_ptr(), meta.data_ptr() + 8, None) == 0
        # no events: recording two per launch halves the duty cycle (the GPU then idles between launches and boosts)
        L.m1v_profile_enable(h, 0)
        go(a.settle)
        torch.cuda.synchronize()
        s = Sampler()
        s.start()
        t0 = time.perf_counter()
        go(a.launches)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.launches
        s.stop = True
        s.join()
        res[nm].append((wall, wall, statistics.median(s.w) if s.w else 0.0, statistics.median(s.mhz) if s.mhz else 0.0))
        if r == 0:
            tot = int(meta[0].item())
            digest = hash(out[:tot].cpu().numpy().tobytes())
            if ref is None:
                ref = (tot, digest)
            assert a.nocheck or (tot, digest) == ref, f"{nm}: output differs from {a.names[0]}"
base = statistics.median(x[0] for x in res[a.names[0]])
for nm in a.names:
    k = statistics.median(x[0] for x in res[nm])
    print(f"{nm:22s} step {k*1e6:7.1f} us  x{base/k:5.3f}  "
          f"{statistics.median(x[2] for x in res[nm]):6.0f} W  {statistics.median(x[3] for x in res[nm]):5.0f} MHz   rounds: "
          + " ".join(f"{x[0]*1e6:.1f}" for x in res[nm]))
