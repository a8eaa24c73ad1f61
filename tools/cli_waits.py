"""What the caller of the chunk loop waits for and how long each kind of host task takes while the others run (EC504_TIMING=1),
for a few thread counts / device lists / slot counts: python tools/cli_waits.py (GPU box; profiles/r04_cli_phase_times.txt)."""
import os, shutil, subprocess, sys, tempfile, time
import numpy as np
from PIL import Image
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
n, W, H = 512, 1920, 1080
d = tempfile.mkdtemp(prefix="ec504_t_", dir="/dev/shm")
try:
    rng = np.random.default_rng(1)
    os.makedirs(d + "/images")
    for i in range(n):
        coarse = rng.integers(0, 256, (H // 40 + 1, W // 40 + 1, 3), dtype=np.uint8).repeat(40, 0).repeat(40, 1)[:H, :W]
        img = np.clip(coarse.astype(np.int16) + rng.integers(-12, 13, (H, W, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(f"{d}/images/f{i:04d}.jpg", quality=90)
    for threads, devices, bit, extra in ((1, "0", 1, 1), (8, "0,0", 1, 1), (16, "0,0", 1, 1), (32, "0,0", 1, 1), (64, "0,0", 1, 1), (64, "0,0", 1, 3), (128, "0,0", 1, 3)):
        shutil.rmtree(d + "/out", ignore_errors=True); os.makedirs(d + "/out")
        env = dict(os.environ, EC504_TIMING="1", EC504_WRITE_BIT=str(bit), EC504_BATCH="16", EC504_DEVICES=devices, EC504_CLI_REPEAT="2", EC504_HOST_THREADS=str(threads), EC504_EXTRA_SLOTS=str(extra))
        p = subprocess.run([ROOT + "/encoder", "images/", "out", "out/v.mpeg", "12", "full"], cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        print(f"threads={threads} devices={devices} bit={bit} extra_slots={extra}")
        for ln in p.stderr.decode().strip().splitlines():
            print("   | " + (ln.split("pinned buffers")[1] if "pinned buffers" in ln else ln), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
