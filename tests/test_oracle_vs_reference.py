"""Pins oracle/mpeg1_oracle.c to the REAL reference (compiled from /root/reference by
`make -C oracle _ref`).  Runs in the authoring container; skipped where the reference is absent.
Covers every row of SURVEY §8(a): colour conversion (exhaustive 2^24), subsampling, FDCT, matrix
scaling, quantise+zigzag, run-length, DC/AC/EOB coding incl. the KATs listed in SURVEY §8(a) rows
10-11, slice/macroblock headers, and the whole driver end to end (STRICT = unmodified reference,
FULL = reference with the two loop-bound literals restored).
"""
import hashlib
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.reference


def test_colour_conversion_exhaustive(ref, orc):
    # all 2^24 RGB triples, in 16 slabs of 2^20
    g, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for r0 in range(0, 256, 16):
        rgb = np.empty((16, 256, 256, 3), np.uint8)
        rgb[..., 0] = np.arange(r0, r0 + 16, dtype=np.uint8)[:, None, None]
        rgb[..., 1] = g
        rgb[..., 2] = b
        mine = orc.convert(rgb)
        theirs = ref.convert(rgb)
        for m, t in zip(mine, theirs):
            assert np.array_equal(m, t)


def test_colour_conversion_rgba(ref, orc):
    rng = np.random.default_rng(7)
    rgba = rng.integers(0, 256, (5000, 4), dtype=np.uint8)
    for m, t in zip(orc.convert(rgba, 4), ref.convert(rgba, 4)):
        assert np.array_equal(m, t)


def test_subsample(ref, orc):
    rng = np.random.default_rng(8)
    for W, H in ((16, 16), (352, 288), (50, 34)):
        cb = rng.integers(0, 256, W * H, dtype=np.uint8)
        cr = rng.integers(0, 256, W * H, dtype=np.uint8)
        for m, t in zip(orc.subsample(cb, cr, W, H), ref.subsample(cb, cr, W, H)):
            assert np.array_equal(m, t)


def _blocks(rng, n):
    blocks = [np.full(64, v, np.uint8) for v in (0, 1, 127, 128, 254, 255)]
    checker = ((np.arange(64) // 8 + np.arange(64) % 8) % 2 * 255).astype(np.uint8)
    blocks += [checker, 255 - checker, np.arange(64, dtype=np.uint8) * 4,
               np.repeat(np.arange(8, dtype=np.uint8) * 36, 8)]
    for _ in range(n):
        kind = rng.integers(0, 4)
        if kind == 0:
            blocks.append(rng.integers(0, 256, 64, dtype=np.uint8))
        elif kind == 1:  # smooth
            base = rng.integers(0, 200)
            blocks.append((base + rng.integers(0, 8, 64)).astype(np.uint8))
        elif kind == 2:  # extremes only
            blocks.append((rng.integers(0, 2, 64) * 255).astype(np.uint8))
        else:  # gradient + noise
            gx, gy = rng.integers(-8, 9, 2)
            i, j = np.divmod(np.arange(64), 8)
            blocks.append(np.clip(128 + gx * j + gy * i + rng.integers(-3, 4, 64), 0, 255).astype(np.uint8))
    return blocks


def test_fdct_bit_exact(ref, orc):
    rng = np.random.default_rng(1)
    for blk in _blocks(rng, 3000):
        t = ref.fast_dct(blk)
        assert np.array_equal(t, np.round(t))  # the doubles hold integers
        assert np.array_equal(orc.fdct(blk), t.astype(np.int32))
    # flat 255 -> DC 2042, every AC = 2
    d = orc.fdct(np.full(64, 255, np.uint8))
    assert d[0] == 2042 and np.all(d[1:] == 2)


def test_scaled_matrix_all_quality_factors(ref, orc):
    for qf in range(-1, 103):
        assert np.array_equal(orc.scale_qmatrix(qf), ref.scale_qmatrix(qf)), qf
    q12 = orc.scale_qmatrix(12)
    assert list(q12[:8]) == [33, 67, 79, 92, 108, 113, 121, 142] and q12[63] == 346
    assert np.all(orc.scale_qmatrix(100) == 1)


def test_quant_zigzag(ref, orc):
    rng = np.random.default_rng(2)
    for qf in (1, 5, 12, 49, 50, 75, 100):
        q = orc.scale_qmatrix(qf)
        for blk in _blocks(rng, 300):
            d = ref.fast_dct(blk)
            assert np.array_equal(orc.quant_zigzag(d.astype(np.int32), q), ref.quant_zigzag(d, qf))
    # negative / large magnitudes the FDCT can produce only rarely
    for _ in range(200):
        d = rng.integers(-2100, 2100, 64).astype(np.float64)
        qf = int(rng.integers(1, 101))
        assert np.array_equal(orc.quant_zigzag(d.astype(np.int32), orc.scale_qmatrix(qf)), ref.quant_zigzag(d, qf))


def _zigzag_cases(rng, n):
    cases = [np.zeros(64, np.int32)]
    for dc in (1, -1, 31, -31, 32, 255, 256, 300, -300, 2042):
        z = np.zeros(64, np.int32); z[0] = dc; cases.append(z)
    z = np.zeros(64, np.int32); z[[0, 2, 5, 6]] = [30, -2, 1, 9]; cases.append(z)  # SURVEY block KAT
    z = np.zeros(64, np.int32); z[63] = 1; cases.append(z)           # run 63 -> escape run field 62
    z = np.zeros(64, np.int32); z[33] = 1; cases.append(z)           # run 33 -> escape
    z = np.zeros(64, np.int32); z[32] = 1; cases.append(z)           # run 32 -> table row 31
    z = np.ones(64, np.int32); cases.append(z)                      # 64 non-zeros
    z = np.zeros(64, np.int32); z[1::2] = 255; cases.append(z)      # longest emission, 2-byte escapes
    z = np.zeros(64, np.int32); z[0] = -5; z[2::2] = -200; cases.append(z)
    for _ in range(n):
        z = np.zeros(64, np.int32)
        density = rng.choice([0.02, 0.1, 0.3, 0.6])
        mask = rng.random(64) < density
        mag = rng.choice([2, 5, 41, 130, 256])
        z[mask] = rng.integers(-mag + 1, mag, mask.sum())
        if rng.random() < 0.3:
            z[0] = 0
        cases.append(z)
    return cases


def test_run_length(ref, orc):
    rng = np.random.default_rng(3)
    for z in _zigzag_cases(rng, 500):
        mine, n = orc.run_length(z)
        theirs = ref.run_length(z)
        assert np.array_equal(mine[:2 * n + 2], theirs[:2 * n + 2])


def test_block_bits_vs_reference(ref, orc):
    rng = np.random.default_rng(4)
    for z in _zigzag_cases(rng, 1500):
        for is_luma in (1, 0):
            rc, mine = orc.encode_block_bits(is_luma, z)
            assert rc == 0
            assert mine == ref.block_bits(is_luma, z), (is_luma, z)


def test_block_bits_known_answers(orc):
    # SURVEY §8(a) rows 10-11 (captured from the reference): bits before the EOB "10"
    def dc(v, luma=1):
        z = np.zeros(64, np.int32); z[0] = v
        rc, s = orc.encode_block_bits(luma, z)
        assert rc == 0 and s.endswith("10")
        return s[:-2]
    assert dc(32) == "11110" + "100000"
    assert dc(31) == "1110" + "11111"
    assert dc(-31) == "1110" + "01111"
    assert dc(1) == "00" + "1" and dc(-1) == "00" + "0" and dc(256) == "00" + "0"
    assert dc(300) == "11110" + "101100"
    assert dc(0) == "100" and dc(32, 0) == "111110" + "100000" and dc(0, 0) == "00"

    def ac(run, level):
        z = np.zeros(64, np.int32); z[run] = level  # DC == 0: first pair has run == position
        rc, s = orc.encode_block_bits(1, z)
        return rc, s[3:-2] if rc == 0 else None
    kat = {(1, 1): "11", (1, -1): "11", (1, 2): "00101", (1, 3): "0000110", (2, 1): "011", (2, -1): "011",
           (2, 18): "0000000000010000", (3, 1): "0101", (17, 1): "0000001000", (32, 1): "0000000000011011",
           (33, 1): "000001" "100000" "00000001", (1, 40): "000001" "000000" "00101000",
           (1, -127): "000001" "000000" "10000001", (5, -200): "000001" "000100" "10000000" "00111000",
           (1, 255): "000001" "000000" "00000000" "11111111"}
    for (run, level), bits in kat.items():
        assert ac(run, level) == (0, bits), (run, level)
    assert ac(1, 256)[0] == orc.E_UNENCODABLE  # the reference returns NULL and segfaults (vlc.c:349)
    z = np.zeros(64, np.int32); z[[0, 2, 5, 6]] = [30, -2, 1, 9]
    assert orc.encode_block_bits(1, z) == (0, "1110" "11110" "00101" "011" "10")


def test_slice_and_macroblock_header(ref, orc):
    import ctypes as C
    for strip in (0, 1, 119, 254, 255, 300):
        b = orc.OrcBits()
        L = orc.lib()
        L.orc_bits_init(C.byref(b))
        L.orc_bits_put(C.byref(b), 0x000001, 24)
        L.orc_bits_put(C.byref(b), (strip + 1) & 0xFF, 8)
        L.orc_bits_put(C.byref(b), 1, 5)
        L.orc_bits_put(C.byref(b), 0, 1)
        L.orc_bits_put(C.byref(b), 3, 2)
        raw = bytes(bytearray(b.buf[i] for i in range(5)))
        mine = "".join(f"{x:08b}" for x in raw)[:40]
        L.orc_bits_free(C.byref(b))
        assert mine == ref.slice_and_mb_bits(strip)
    assert ref.slice_and_mb_bits(0) == f"{0x000001010b:040b}"


# ------------------------------------------------------------------------------------------------
# end to end through the reference driver binaries
# ------------------------------------------------------------------------------------------------

def _make_folder(tmp, name, frames, quality=90):
    from PIL import Image
    d = tmp / name
    (d / "images").mkdir(parents=True)
    (d / "bit").mkdir()
    for i, a in enumerate(frames):
        Image.fromarray(a).save(str(d / "images" / f"f{i:03d}.jpg"), quality=quality)
    return d


def _synthetic(rng, n, W, H, kind):
    out = []
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(n):
        if kind == "noise":
            a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        else:
            a = np.stack([(xx * 255 // max(W - 1, 1) + 13 * k) % 256, (yy * 255 // max(H - 1, 1)) % 256,
                          ((xx + yy) // 2 + 40 * k) % 256], -1).astype(np.int32)
            a = np.clip(a + rng.integers(-12, 13, a.shape), 0, 255).astype(np.uint8)
        out.append(a)
    return out


def _check_folder(ref, orc, d, qf, modes=("strict", "full")):
    names, frames = ref.dump_rgb(str(d / "images"), str(d / "dump"))
    assert frames, "no frames decoded"
    H, W, Cn = frames[0].shape
    rgb = np.concatenate([f.reshape(-1) for f in frames])
    for mode in modes:
        video = d / "bit" / f"{mode}_{qf}.mpeg"
        assert ref.run_encoder(str(d / "images"), str(d / "bit"), str(video), qf, mode) == 0
        theirs = video.read_bytes()
        mine = orc.encode_sequence(rgb, len(frames), W, H, qf, orc.MODE_STRICT if mode == "strict" else orc.MODE_FULL, Cn)
        assert mine == theirs, (mode, qf, len(mine), len(theirs))
    # .bit side files (written by either run; identical)
    for i, f in enumerate(frames):
        Y, Cb, Cr = orc.convert(f, Cn)
        want = struct.pack("<ii", W, H) + Y.tobytes() + Cb.tobytes() + Cr.tobytes()
        assert (d / "bit" / f"image_{i + 1}.bit").read_bytes() == want
    return names, frames


@pytest.mark.parametrize("W,H,kind,n", [(352, 288, "noise", 4), (352, 288, "smooth", 3), (96, 144, "noise", 2),
                                        (112, 160, "smooth", 2), (400, 600, "smooth", 2), (1920, 1080, "smooth", 1),
                                        (360, 250, "noise", 1)])
def test_driver_end_to_end(ref, orc, tmp_path, W, H, kind, n):
    rng = np.random.default_rng(W * 7 + H)
    d = _make_folder(tmp_path, "e2e", _synthetic(rng, n, W, H, kind))
    _check_folder(ref, orc, d, 12)


@pytest.mark.parametrize("qf", [1, 5, 30, 49, 50, 75, 100])
def test_driver_quality_sweep(ref, orc, tmp_path, qf):
    rng = np.random.default_rng(qf)
    # smooth content keeps |level| < 256 at qf=100 (noise would make the reference segfault, vlc.c:349)
    d = _make_folder(tmp_path, "q", _synthetic(rng, 2, 176, 144, "smooth"))
    _check_folder(ref, orc, d, qf)


def test_driver_300_frames_hour_wrap(ref, orc, tmp_path):
    # 300 frames: uint8 hour wraps at 256, the GOP field keeps 5 bits (encoder.h:42, mpeg1_enc.c:109)
    rng = np.random.default_rng(300)
    base = _synthetic(rng, 3, 96, 144, "smooth")
    d = _make_folder(tmp_path, "wrap", [base[i % 3] for i in range(300)])
    _check_folder(ref, orc, d, 12, modes=("strict",))


def test_driver_sample_images(ref, orc, tmp_path):
    """The reference's own sample data (images.zip: 30 JPEGs 400x600, 3 distinct)."""
    import zipfile
    z = "/root/reference/images.zip"
    if not os.path.exists(z):
        pytest.skip("images.zip absent")
    d = tmp_path / "sample"
    (d / "bit").mkdir(parents=True)
    with zipfile.ZipFile(z) as zf:
        members = [m for m in zf.namelist() if m.lower().endswith((".jpg", ".jpeg")) and "__MACOSX" not in m]
        for m in members:
            target = d / "images" / os.path.basename(m)
            target.parent.mkdir(exist_ok=True)
            target.write_bytes(zf.read(m))
    _check_folder(ref, orc, d, 12)


def test_4k_strip_byte_and_dimension_wrap(ref, orc, tmp_path):
    # 3840x2160: SEQ w&0xFF = 0, h&0xFF = 112, strip start codes up to 0xF0 (SURVEY §8 grammar)
    rng = np.random.default_rng(4)
    d = _make_folder(tmp_path, "uhd", _synthetic(rng, 1, 3840, 2160, "smooth"), quality=75)
    _check_folder(ref, orc, d, 12)


def test_more_than_255_strips_wraps_the_slice_byte(ref, orc, tmp_path):
    # 4128 wide = 258 strips: the uint8 vertical position (mpeg1_blk.c slice start code) passes 0xFF, strip 255 is
    # written as 00 00 01 00 and strips 256, 257 reuse 01, 02 — the reference does this, so the oracle must too
    rng = np.random.default_rng(41)
    d = _make_folder(tmp_path, "wide", _synthetic(rng, 2, 4128, 32, "noise"), quality=90)
    _check_folder(ref, orc, d, 12, modes=("full",))

