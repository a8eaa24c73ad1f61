#!/usr/bin/env python3
"""Phase times of the folder-level CLI path on the GPU box: makes N JPEG files in /dev/shm and runs ./encoder with
EC504_TIMING=1 for a few host-thread counts.   tools/cli_timing.py [N] [W] [H]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, W, H = (int(x) for x in (sys.argv[1:4] + ["128", "1920", "1080"][len(sys.argv) - 1:]))
d = tempfile.mkdtemp(prefix="ec504_t_", dir="/dev/shm")
try:
    rng = np.random.default_rng(1)
    os.makedirs(d + "/images")
    for i in range(n):
        coarse = rng.integers(0, 256, (H // 40 + 1, W // 40 + 1, 3), dtype=np.uint8).repeat(40, 0).repeat(40, 1)[:H, :W]
        img = np.clip(coarse.astype(np.int16) + rng.integers(-12, 13, (H, W, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(f"{d}/images/f{i:04d}.jpg", quality=90)
    # (host threads, .bit files, frames per chunk, EC504_DEVICES, calls in one process)
    for threads, bit, batch, devices, repeat in ((0, 1, 16, "0", 1), (0, 1, 16, "0,0", 1), (0, 1, 16, "0,0,0", 1), (0, 0, 16, "0", 1), (0, 0, 16, "0,0", 1),
                                                 (16, 1, 16, "0,0", 1), (64, 1, 16, "0,0", 1), (128, 1, 16, "0,0", 1), (256, 1, 16, "0,0", 1),
                                                 (0, 1, 64, "0,0", 1), (0, 1, 16, "0,0", 3), (0, 0, 16, "0,0", 3), (64, 1, 16, "0,0", 3), (256, 1, 16, "0,0", 3), (1, 1, 16, "0", 1)):
        shutil.rmtree(d + "/out", ignore_errors=True)
        os.makedirs(d + "/out")
        env = dict(os.environ, EC504_TIMING="1", EC504_WRITE_BIT=str(bit), EC504_BATCH=str(batch), EC504_DEVICES=devices,
                   EC504_CLI_REPEAT=str(repeat))
        if threads:
            env["EC504_HOST_THREADS"] = str(threads)
        t0 = time.perf_counter()
        p = subprocess.run([ROOT + "/encoder", "images/", "out", "out/v.mpeg", "12", "full"], cwd=d, env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        lines = p.stderr.decode().strip().splitlines()
        print(f"threads={threads or 'auto'} write_bit={bit} batch={batch} devices={devices} calls={repeat}: process {time.perf_counter() - t0:.3f} s", flush=True)
        for ln in lines:
            print("     | " + ln, flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
