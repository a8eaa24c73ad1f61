"""The oracle against the committed golden vectors (tests/golden/, produced by the real reference via
tests/golden/make_goldens.py).  Runs anywhere gcc is (CPU only) — this is what pins the oracle on
machines where /root/reference does not exist (the GPU box)."""
import hashlib
import struct

import numpy as np
import pytest

import golden_io as G


def test_kat_blocks(orc):
    z = G.load("kat_blocks.npz")
    blocks, dct = z["blocks"], z["dct"].astype(np.int32)
    for i in range(len(blocks)):
        assert np.array_equal(orc.fdct(blocks[i]), dct[i]), i
    for qf in (12, 50, 100):
        q = orc.scale_qmatrix(qf)
        zz = z[f"zz_q{qf}"].astype(np.int32)
        for i in range(len(blocks)):
            assert np.array_equal(orc.quant_zigzag(dct[i], q), zz[i]), (qf, i)
        if qf == 100:
            continue
        for luma, name in ((1, "luma"), (0, "chroma")):
            want = G.unpack_bits(z[f"bits_q{qf}_{name}"], z[f"offs_q{qf}_{name}"])
            for i in range(len(blocks)):
                assert orc.encode_block_bits(luma, zz[i]) == (0, want[i]), (qf, name, i)


def test_kat_vlc(orc):
    z = G.load("kat_vlc.npz")
    zz = z["zz"].astype(np.int32)
    for i in range(len(zz)):
        pairs, n = orc.run_length(zz[i])
        assert n == z["npairs"][i]
        assert np.array_equal(pairs[:2 * n + 2], z["pairs"][i][:2 * n + 2].astype(np.int32))
    for luma, name in ((1, "luma"), (0, "chroma")):
        want = G.unpack_bits(z[f"bits_{name}"], z[f"offs_{name}"])
        for i in range(len(zz)):
            assert orc.encode_block_bits(luma, zz[i]) == (0, want[i]), (name, i)


def test_qmatrix(orc):
    qm = np.load(G.GOLDEN + "/qmatrix.npy")
    for k, qf in enumerate(range(-1, 103)):
        assert np.array_equal(orc.scale_qmatrix(qf), qm[k].astype(np.int32)), qf


def test_colour_sample_and_subsample(orc):
    z = G.load("kat_colour.npz")
    Y, Cb, Cr = orc.convert(z["rgb"])
    assert np.array_equal(Y, z["Y"]) and np.array_equal(Cb, z["Cb"]) and np.array_equal(Cr, z["Cr"])
    a, b = orc.subsample(z["sub_in_cb"], z["sub_in_cr"], int(z["sub_w"]), int(z["sub_h"]))
    assert np.array_equal(a, z["sub_cb"]) and np.array_equal(b, z["sub_cr"])


def test_colour_exhaustive_sha(orc):
    want = G.load_json("colour_exhaustive.json")["sha256"]
    hs = [hashlib.sha256() for _ in range(3)]
    g, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for r0 in range(0, 256, 16):
        rgb = np.empty((16, 256, 256, 3), np.uint8)
        rgb[..., 0] = np.arange(r0, r0 + 16, dtype=np.uint8)[:, None, None]
        rgb[..., 1] = g
        rgb[..., 2] = b
        for h, plane in zip(hs, orc.convert(rgb)):
            h.update(plane.tobytes())
    assert [h.hexdigest() for h in hs] == [want["Y"], want["Cb"], want["Cr"]]


@pytest.mark.parametrize("fn", G.E2E_FILES)
def test_end_to_end_files(orc, fn):
    z = G.load(fn)
    rgb = z["rgb"]
    n, H, W, C = rgb.shape
    for _, qf, mode, _, want in G.e2e_cases([fn]):
        m = orc.MODE_STRICT if mode == "strict" else orc.MODE_FULL
        assert orc.encode_sequence(rgb, n, W, H, qf, m, C) == want, (fn, qf, mode)
        # the frame-parallel batch entry gives the same bytes
        body, sizes = orc.encode_frames(rgb, n, W, H, 0, qf, m, C, threads=3)
        assert orc.file_prolog() + body == want and int(sizes.sum()) == len(body)
    for i in range(n):  # .bit side files
        Y, Cb, Cr = orc.convert(rgb[i], C)
        blob = struct.pack("<ii", W, H) + Y.tobytes() + Cb.tobytes() + Cr.tobytes()
        assert hashlib.sha256(blob).hexdigest() == str(z["bit_sha256"][i])


def test_300_frames_hour_wrap(orc):
    z = G.load("e2e_300_wrap.npz")
    idx = z["frame_index"]
    rgb = z["rgb"][idx]
    n, H, W, C = rgb.shape
    assert n == 300
    want = z["mpeg_strict_q12"].tobytes()
    assert orc.encode_sequence(rgb, n, W, H, 12, orc.MODE_STRICT, C, threads=4) == want
    # frames 0 and 256 of the same picture differ only through the global index (hour = i & 0xff)
    a = orc.encode_frame(rgb[0], W, H, 0, 12, orc.MODE_STRICT, C)
    b = orc.encode_frame(rgb[0], W, H, 256, 12, orc.MODE_STRICT, C)
    assert a == b


def test_packet_length_wraps_mod_65536(orc):
    # 1080p FULL payload is > 64 KiB: the 16-bit length field wraps (encoder.h:448-453)
    z = G.load("e2e_1080p.npz")
    want = z["mpeg_full_q12"].tobytes()
    payload = len(want) - 27 - 44 - 4
    assert payload > 65535
    assert struct.unpack(">H", want[27 + 4:27 + 6])[0] == (payload + 36) & 0xFFFF


def test_geometry_errors(orc):
    rgb = np.zeros((64, 64, 3), np.uint8)
    with pytest.raises(ValueError):
        orc.encode_frame(rgb, 64, 64, 0, 12, orc.MODE_STRICT)  # 96x144 region does not fit
    assert len(orc.encode_frame(rgb, 64, 64, 0, 12, orc.MODE_FULL)) > 48
    with pytest.raises(ValueError):
        orc.encode_frame(np.zeros((144, 96, 2), np.uint8), 96, 144, 0, 12, orc.MODE_STRICT, channels=2)


def test_synth_generator_is_stable(orc):
    f = orc.synth_frames(2, 16, 16, seed=504)
    assert f.shape == (2, 16, 16, 3)
    assert hashlib.sha256(f.tobytes()).hexdigest()[:16] == hashlib.sha256(orc.synth_frames(2, 16, 16, seed=504).tobytes()).hexdigest()[:16]
    assert not np.array_equal(f[0], f[1])
    # splitmix64 known answer: state 0 -> first output 0xE220A8397B1DCDAF
    z = orc.synth_frames(1, 8, 1, seed=0, first_index=0, channels=1)
    assert int.from_bytes(z.tobytes(), "little") == 0xE220A8397B1DCDAF
