"""The host driver (csrc/encoder_host.c: directory scan, parallel JPEG decode, pinned-batch staging, .bit write-behind)
run in THIS container, where there is no GPU: it is linked with tests/host_driver_standin.c (the m1v_* calls it makes,
answered by the oracle) and tools/encoder_cli.c + the reference's stb_image.h, under ThreadSanitizer and
AddressSanitizer, and its files are compared with the real reference binary's on the same folder of JPEGs.
Needs /root/reference (stb_image.h, oracle/_ref binaries), so it runs here and is skipped on the GPU box, where
tests/test_dropin.py does the same comparison through the real library."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STB_DIR = "/root/reference/include"
REF = {"strict": os.path.join(ROOT, "oracle", "_ref", "ref_encoder_strict"),
       "full": os.path.join(ROOT, "oracle", "_ref", "ref_encoder_full")}
SOURCES = ["tools/encoder_cli.c", "ec504_imageencoder_amd/csrc/encoder_host.c", "tests/host_driver_standin.c",
           "oracle/mpeg1_oracle.c"]

pytestmark = pytest.mark.reference


@pytest.fixture(scope="module")
def builds(tmp_path_factory):
    if not os.path.exists(os.path.join(STB_DIR, "stb_image.h")):
        pytest.skip("reference stb_image.h not present")
    if not all(os.path.exists(p) for p in REF.values()):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    out = tmp_path_factory.mktemp("hostdrv")
    exes = {}
    for san in ("thread", "address"):
        exe = str(out / f"host_{san}")
        cmd = ["gcc", "-O1", "-g", "-w", f"-fsanitize={san}", "-fno-omit-frame-pointer", "-ffp-contract=off",
               "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"), "-idirafter", STB_DIR,
               *[os.path.join(ROOT, s) for s in SOURCES], "-o", exe, "-lm", "-lpthread"]
        subprocess.run(cmd, check=True)
        exes[san] = exe
    return exes


def _folder(d, n, W=352, H=288, seed=7, quality=90):
    from PIL import Image
    rng = np.random.default_rng(seed)
    (d / "images").mkdir()
    for i in range(n):
        base = rng.integers(0, 256, (H // 8, W // 8, 3), dtype=np.uint8).repeat(8, 0).repeat(8, 1)
        noise = rng.integers(-20, 21, (H, W, 3))
        Image.fromarray(np.clip(base + noise, 0, 255).astype(np.uint8)).save(str(d / "images" / f"im{i:02d}.jpg"), quality=quality)
    (d / "images" / "readme.txt").write_text("x")
    (d / "images" / "zz_broken.jpeg").write_bytes(b"\xff\xd8\xff\xe0 not a jpeg")     # reported and skipped (encoder.h:163-167)
    return n


def _run(exe, cwd, *args, env=None):
    e = dict(os.environ, TSAN_OPTIONS="exitcode=66 halt_on_error=1", ASAN_OPTIONS="exitcode=67 detect_leaks=0")
    e.update(env or {})
    p = subprocess.run([exe, *args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    return p.returncode, p.stderr.decode(errors="replace")


def _reference_files(d, region, n):
    (d / "ref").mkdir()
    rc, _ = _run(REF[region], str(d), "images/", "ref", "ref/v.mpeg", "12")
    assert rc == 0
    return (d / "ref" / "v.mpeg").read_bytes(), [(d / "ref" / f"image_{k}.bit").read_bytes() for k in range(1, n + 1)]


@pytest.mark.parametrize("san,region,threads,batch", [("thread", "full", 6, 3), ("thread", "strict", 3, 64),
                                                      ("address", "full", 4, 4), ("address", "strict", 1, 1)])
def test_threaded_host_driver_matches_reference_binary(builds, san, region, threads, batch, tmp_path):
    n = _folder(tmp_path, 10)
    video, bits = _reference_files(tmp_path, region, n)
    (tmp_path / "out").mkdir()
    rc, err = _run(builds[san], str(tmp_path), "images/", "out", "out/v.mpeg", "12", region,
                   env={"EC504_HOST_THREADS": str(threads), "EC504_BATCH": str(batch)})
    assert rc == 0, err[-3000:]
    assert (tmp_path / "out" / "v.mpeg").read_bytes() == video
    for k in range(1, n + 1):
        assert (tmp_path / "out" / f"image_{k}.bit").read_bytes() == bits[k - 1], k
    assert not (tmp_path / "out" / f"image_{n + 1}.bit").exists()


@pytest.mark.parametrize("san,devices,threads,batch", [("thread", "0", 4, 2), ("thread", "0,0,0", 6, 1), ("address", "0,1", 3, 3),
                                                       ("thread", "", 1, 2), ("address", "0,0,0,0", 5, 64)])
def test_several_encoders_in_flight_and_kept_between_calls(builds, san, devices, threads, batch, tmp_path):
    """EC504_DEVICES: one encoder per list entry, chunk c on entry c mod N, N chunks on GPUs at once, retired in order (the
    stand-in ignores the device index, so "0,1" runs here too).  Same bytes as the reference binary for every lane count —
    one lane, the default two on one device, more lanes than chunks — and for a SECOND call in the same process, which takes
    over the encoders and pinned buffers of the first (EC504_CLI_REPEAT)."""
    n = _folder(tmp_path, 11)
    video, bits = _reference_files(tmp_path, "full", n)
    (tmp_path / "out").mkdir()
    env = {"EC504_HOST_THREADS": str(threads), "EC504_BATCH": str(batch), "EC504_CLI_REPEAT": "2"}
    if devices:
        env["EC504_DEVICES"] = devices
    rc, err = _run(builds[san], str(tmp_path), "images/", "out", "out/v.mpeg", "12", "full", env=env)
    assert rc == 0, err[-3000:]
    assert (tmp_path / "out" / "v.mpeg").read_bytes() == video
    assert (tmp_path / "out" / "v.mpeg.2").read_bytes() == video
    for k in range(1, n + 1):
        assert (tmp_path / "out" / f"image_{k}.bit").read_bytes() == bits[k - 1], k


def test_threaded_error_paths_and_opt_out(builds, tmp_path):
    """Dimension mismatch found after the parallel decode: -1, only the 27-byte prolog written, no .bit files (what the
    reference leaves behind, encoder.h:175-183); EC504_WRITE_BIT=0 skips the side files; an empty folder gives -1."""
    from PIL import Image
    n = _folder(tmp_path, 5)
    video, _ = _reference_files(tmp_path, "strict", n)
    (tmp_path / "o1").mkdir()
    rc, err = _run(builds["thread"], str(tmp_path), "images/", "o1", "o1/v.mpeg", "12", "strict", env={"EC504_WRITE_BIT": "0"})
    assert rc == 0, err[-3000:]
    assert (tmp_path / "o1" / "v.mpeg").read_bytes() == video and not list((tmp_path / "o1").glob("*.bit"))
    Image.fromarray(np.zeros((160, 208, 3), np.uint8)).save(str(tmp_path / "images" / "odd.jpg"))
    rc, err = _run(builds["address"], str(tmp_path), "images/", "o2", "o2/v.mpeg", "12", "strict")
    assert rc == 1                                             # o2/ does not exist yet: fopen fails first (encoder.h:75-80)
    (tmp_path / "o2").mkdir()
    # the odd file is met somewhere along the folder; with one-file chunks several frames have been encoded and
    # written by then and must be taken back: same end state as the reference, which checks before it encodes
    order = [e.name for e in os.scandir(tmp_path / "images") if ".jpg" in e.name]
    assert "odd.jpg" in order
    for exe, batch in ((builds["address"], "1"), (builds["thread"], "2"), (builds["address"], "64")):
        rc, err = _run(exe, str(tmp_path), "images/", "o2", "o2/v.mpeg", "12", "strict", env={"EC504_BATCH": batch})
        assert rc == 255, err[-3000:]                          # -1 as a process exit status
        assert (tmp_path / "o2" / "v.mpeg").stat().st_size == 27 and not list((tmp_path / "o2").glob("*.bit"))
    rc, _ = _run(REF["strict"], str(tmp_path), "images/", "o2", "o2/r.mpeg", "12")
    assert rc == 255 and (tmp_path / "o2" / "r.mpeg").stat().st_size == 27 and not list((tmp_path / "o2").glob("*.bit"))
    (tmp_path / "empty").mkdir()
    rc, err = _run(builds["thread"], str(tmp_path), "empty", "o2", "o2/w.mpeg", "12", "strict")
    assert rc == 255, err[-3000:]


def test_output_buffer_grows_when_a_batch_exceeds_the_first_guess(builds, tmp_path):
    """The driver's pinned output buffer starts at 1/16 of the worst case.  Pictures built from isolated DCT
    coefficients (long code sequences; ordinary pictures stop at the first run-0 pair and stay far below) need more, so
    the batch is redone with the worst-case buffer (encoder_host.c batch loop).  Raw-pixel loader, compared with the oracle."""
    import struct
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import oracle_ffi as orc
    from test_gpu_parity import _heavy_picture
    exe = str(tmp_path / "raw_main")
    subprocess.run(["gcc", "-O1", "-g", "-w", "-fsanitize=address", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "oracle"), *[os.path.join(ROOT, f) for f in ["tests/raw_loader_main.c"] + SOURCES[1:]],
                    "-o", exe, "-lm", "-lpthread"], check=True)
    rng = np.random.default_rng(5)
    W, H, n, qf = 176, 144, 5, 90
    frames = np.stack([_heavy_picture(rng, W, H, 16, 120) for _ in range(n)])
    (tmp_path / "images").mkdir()
    for i in range(n):
        (tmp_path / "images" / f"p{i}.jpg").write_bytes(struct.pack("<iii", W, H, 3) + frames[i].tobytes())
    order = [int(e.name[1]) for e in os.scandir(tmp_path / "images")]
    want = orc.encode_sequence(frames[order], n, W, H, qf, orc.MODE_FULL)
    bound = orc.frame_bound(W, H, orc.MODE_FULL)
    batch = 3
    sizes = [len(orc.encode_sequence(frames[i:i + 1], 1, W, H, qf, orc.MODE_FULL)) - 27 for i in order]
    assert sum(sizes[:batch]) > bound * (batch + 1) // 16, "pictures too easy to exercise the grow path"
    (tmp_path / "out").mkdir()
    rc, err = _run(exe, str(tmp_path), "images", "out", "out/v.mpeg", str(qf), "full", env={"EC504_BATCH": str(batch), "EC504_HOST_THREADS": "3"})
    assert rc == 0, err[-3000:]
    assert (tmp_path / "out" / "v.mpeg").read_bytes() == want
