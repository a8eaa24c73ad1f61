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
    for threads in (16, 32, 64, 128):
        for tun in ("", "glibc.malloc.mmap_threshold=1073741824:glibc.malloc.trim_threshold=17179869184:glibc.malloc.top_pad=268435456"):
            for bit in (1, 0):
                shutil.rmtree(d + "/out", ignore_errors=True); os.makedirs(d + "/out")
                env = dict(os.environ, EC504_TIMING="1", EC504_WRITE_BIT=str(bit), EC504_BATCH="16", EC504_DEVICES="0,0", EC504_CLI_REPEAT="3", EC504_HOST_THREADS=str(threads))
                if tun: env["GLIBC_TUNABLES"] = tun
                t0 = time.perf_counter()
                p = subprocess.run([ROOT + "/encoder", "images/", "out", "out/v.mpeg", "12", "full"], cwd=d, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                lines = p.stderr.decode().strip().splitlines()
                ch = [ln.split("chunks ")[1].split(" s")[0] for ln in lines if "chunks" in ln]
                print(f"threads={threads} tunables={'on' if tun else 'off'} bit={bit}: process {time.perf_counter()-t0:.3f} s chunks {ch}", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
