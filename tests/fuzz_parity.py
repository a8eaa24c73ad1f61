#!/usr/bin/env python3
"""(The soak is run by hand on a GPU box: python tests/fuzz_parity.py 300; tests/test_gpu_parity.py runs a seeded slice of it.)
Fuzz soak on the GPU: random geometry / quality / channels / mode / dense run length / LDS image size / content
class, HIP stream vs oracle stream, byte for byte.  usage: fuzz_parity.py [seconds] [seed] [big]
"big" draws large pictures (up to 4128 x 2304, up to 6 frames): fewer cases, long strips, many strips, offsets > 2^24."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # this file lives in tests/: the oracle is test infrastructure
import numpy as np
import torch
import oracle_ffi as orc
from ec504_imageencoder_amd import EncoderError, Mpeg1Encoder


def content(rng, kind, n, H, W, C):
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == "noise":
        a = rng.integers(0, 256, (n, H, W, C), dtype=np.uint8)
    elif kind == "extremes":
        a = (rng.integers(0, 2, (n, H, W, C)) * 255).astype(np.uint8)
    elif kind == "smooth":
        base = (xx * 3 + yy * 2) % 256
        a = np.clip(base[None, :, :, None] + rng.integers(-6, 7, (n, H, W, C)), 0, 255).astype(np.uint8)
    elif kind == "grey":    # r == g == b: every chroma sample is an exact tie of the colour formulas (fp64 path)
        v = rng.integers(0, 256, (n, H, W, 1), dtype=np.uint8)
        a = np.repeat(v, C, 3)
    elif kind == "blocks":
        cell = int(rng.choice([2, 3, 4, 5, 8]))
        v = rng.integers(0, 256, (n, (H + cell - 1) // cell, (W + cell - 1) // cell, C), dtype=np.uint8)
        a = np.repeat(np.repeat(v, cell, 1), cell, 2)[:, :H, :W]
    else:  # stripes: strong isolated coefficients
        period = int(rng.choice([2, 3, 4, 6, 8, 16]))
        s = ((xx // period + (yy // period) * int(rng.integers(0, 2))) % 2 * 255).astype(np.uint8)
        a = np.broadcast_to(s[None, :, :, None], (n, H, W, C)).copy()
        a ^= rng.integers(0, 8, a.shape, dtype=np.uint8)
    return np.ascontiguousarray(a)


def run(budget=120.0, seed=2026, big=False, max_cases=None, max_pixels=1920 * 1200, verbose=True):
    """Returns (cases, expected-unencodable, failures as a list of descriptions)."""
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases = skipped = 0
    fails = []
    t_report = time.time() + 30
    while time.time() < t_end and (max_cases is None or cases < max_cases):
        if verbose and time.time() > t_report:  # a line every 30 s: long silent runs look hung to the job runner
            print(f"... {cases} cases, {len(fails)} failures so far", flush=True)
            t_report = time.time() + 30
        mode = "full" if rng.random() < 0.8 else "strict"
        if big:
            W = int(rng.choice([1280, 1920, 2048, 2560, 3840, 4096, 4112, 4128]))
            H = int(rng.choice([144, 720, 1080, 1088, 1440, 2160, 2304]))
        else:
            W = int(rng.choice([16, 24, 96, 100, 101, 176, 200, 333, 352, 366, 640, 720, 1024, 1366, 1920]))
            H = int(rng.choice([16, 40, 49, 144, 150, 288, 300, 301, 480, 576, 768, 1088, 1504, 2304]))
        if mode == "strict" and (W < 96 or H < 144):
            continue
        if not big and W * H > max_pixels:
            continue
        C = 3 if rng.random() < 0.85 else 4
        qf = int(rng.choice([1, 5, 12, 12, 12, 25, 40, 50, 60, 75, 76, 77, 85, 92, 100]))
        n = int(rng.integers(1, 7 if big else 5))
        kind = str(rng.choice(["noise", "extremes", "smooth", "blocks", "stripes", "grey"]))
        rgb = content(rng, kind, n, H, W, C)
        first = int(rng.integers(0, 600))
        m = orc.MODE_FULL if mode == "full" else orc.MODE_STRICT
        try:
            want, wsizes = orc.encode_frames(rgb, n, W, H, first, qf, m, channels=C, threads=8)
            expect_error = False
        except ValueError:
            expect_error = True
        enc = Mpeg1Encoder(W, H, qf, mode, channels=C, max_frames=n)
        bps = enc.mb_rows * 6
        if bps >= 64 and rng.random() < 0.6:
            choices = [t for t in (64, 128, 192, 256, 320, 384) if t <= bps]
            enc.debug_set_dense_threads(int(rng.choice(choices)))
        if rng.random() < 0.4:
            enc.debug_set_lds_words(int(rng.choice([4, 16, 64, 256, 1024, 4096])))
        if rng.random() < 0.3:
            enc.set_pipelined(True)
        if rng.random() < 0.25:
            enc.debug_set_input_mode(int(rng.choice([0, 2])))
        if C == 3 and rng.random() < 0.5:
            enc.debug_set_path("tiles")     # wins over a forced run length; a forced input mode still selects the run kernel
        shift = int(rng.choice([0, 0, 1, 2, 3, 4, 8]))          # where the frames start inside their allocation
        flat = torch.empty(rgb.size + 16, dtype=torch.uint8, device="cuda")
        dev = flat[shift:shift + rgb.size].view(rgb.shape)
        dev.copy_(torch.from_numpy(rgb))
        desc = f"{W}x{H}x{C} {mode} qf{qf} n{n} {kind} first{first} path={enc.path} shift{shift}"
        try:
            got, sizes = enc.encode_to_bytes(dev, first)
            if expect_error:
                fails.append("MISSED ERROR " + desc)
            elif got != want or sizes != [int(x) for x in wsizes]:
                fails.append(f"MISMATCH {desc} {len(got)} {len(want)}")
        except EncoderError as e:
            if not expect_error or e.code != -2:
                fails.append(f"UNEXPECTED ERROR {desc} {e}")
            else:
                skipped += 1
        enc.close()
        cases += 1
    return cases, skipped, fails


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    cases, skipped, fails = run(budget, int(sys.argv[2]) if len(sys.argv) > 2 else 2026, len(sys.argv) > 3 and sys.argv[3] == "big")
    for f in fails:
        print(f)
    print(f"fuzz: {cases} cases, {skipped} expected-unencodable, {len(fails)} failures")
    sys.exit(1 if fails else 0)
