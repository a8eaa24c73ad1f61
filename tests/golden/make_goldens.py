#!/usr/bin/env python3
"""Generates tests/golden/*.npz|json from the REAL reference (oracle/_ref, built by
`make -C oracle _ref` from /root/reference's own sources).  Authoring container only.

Everything written here is DATA: inputs (pixels as decoded by the reference's stb_image, blocks,
coefficient arrays) and the outputs the reference produced for them.  No reference source text.

    python tests/golden/make_goldens.py
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_ffi as ref  # noqa: E402

from PIL import Image  # noqa: E402


def pack_bits(strings):
    """list of '0101..' strings -> (uint8 blob, uint32 bit offsets[n+1])."""
    offs = np.zeros(len(strings) + 1, np.uint32)
    for i, s in enumerate(strings):
        offs[i + 1] = offs[i] + len(s)
    allbits = "".join(strings)
    allbits += "0" * (-len(allbits) % 8)
    blob = np.frombuffer(int(allbits, 2).to_bytes(len(allbits) // 8, "big") if allbits else b"", np.uint8)
    return blob, offs


def smooth_frames(rng, n, W, H, amp=12):
    yy, xx = np.mgrid[0:H, 0:W]
    out = []
    for k in range(n):
        a = np.stack([(xx * 255 // max(W - 1, 1) + 13 * k) % 256, (yy * 255 // max(H - 1, 1)) % 256,
                      ((xx + yy) // 2 + 40 * k) % 256], -1).astype(np.int32)
        out.append(np.clip(a + rng.integers(-amp, amp + 1, a.shape), 0, 255).astype(np.uint8))
    return out


def coarse_frames(rng, n, W, H, cell=8, amp=40):
    """Gradient + piecewise-constant noise (cell x cell): rich coefficients, compresses well."""
    yy, xx = np.mgrid[0:H, 0:W]
    out = []
    for k in range(n):
        a = np.stack([(xx * 255 // max(W - 1, 1) + 13 * k) % 256, (yy * 255 // max(H - 1, 1)) % 256,
                      ((xx + yy) // 2 + 40 * k) % 256], -1).astype(np.int32)
        nz = rng.integers(-amp, amp + 1, ((H + cell - 1) // cell, (W + cell - 1) // cell, 3))
        nz = np.repeat(np.repeat(nz, cell, 0), cell, 1)[:H, :W]
        out.append(np.clip(a + nz, 0, 255).astype(np.uint8))
    return out


def run_folder(frames_for_files, qfs, modes, jpeg_quality=90):
    """Writes JPEGs, runs the reference drivers, returns dict of results."""
    tmp = tempfile.mkdtemp(prefix="ec504_gold_")
    try:
        os.makedirs(os.path.join(tmp, "images"))
        os.makedirs(os.path.join(tmp, "bit"))
        for i, a in enumerate(frames_for_files):
            Image.fromarray(a).save(os.path.join(tmp, "images", f"f{i:03d}.jpg"), quality=jpeg_quality)
        names, frames = ref.dump_rgb(os.path.join(tmp, "images"), os.path.join(tmp, "dump"))
        res = {"names": names, "frames": frames, "mpeg": {}}
        for qf in qfs:
            for mode in modes:
                v = os.path.join(tmp, "bit", "v.mpeg")
                rc = ref.run_encoder(os.path.join(tmp, "images"), os.path.join(tmp, "bit"), v, qf, mode)
                assert rc == 0, (qf, mode, rc)
                res["mpeg"][(qf, mode)] = open(v, "rb").read()
        res["bit_sha256"] = [hashlib.sha256(open(os.path.join(tmp, "bit", f"image_{i + 1}.bit"), "rb").read()).hexdigest()
                             for i in range(len(frames))]
        return res
    finally:
        shutil.rmtree(tmp)


def save_e2e(name, res, extra=None):
    frames = res["frames"]
    d = {"names": np.array(res["names"]), "rgb": np.stack(frames), "bit_sha256": np.array(res["bit_sha256"])}
    for (qf, mode), data in res["mpeg"].items():
        d[f"mpeg_{mode}_q{qf}"] = np.frombuffer(data, np.uint8)
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(HERE, name), **d)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in d.items() if k.startswith("mpeg")})


def main():
    assert ref.ensure_built(), "reference sources not available"
    rng = np.random.default_rng(504)

    # 1. block-level known answers: u8 block -> FDCT -> quantise+zigzag (qf 12/50/100) -> block bits
    blocks = [np.full(64, v, np.uint8) for v in (0, 1, 128, 255)]
    checker = ((np.arange(64) // 8 + np.arange(64) % 8) % 2 * 255).astype(np.uint8)
    blocks += [checker, 255 - checker]
    for k in range(1018):
        kind = k % 4
        if kind == 0:
            blocks.append(rng.integers(0, 256, 64, dtype=np.uint8))
        elif kind == 1:
            blocks.append((rng.integers(0, 200) + rng.integers(0, 8, 64)).astype(np.uint8))
        elif kind == 2:
            blocks.append((rng.integers(0, 2, 64) * 255).astype(np.uint8))
        else:
            gx, gy = rng.integers(-8, 9, 2)
            i, j = np.divmod(np.arange(64), 8)
            blocks.append(np.clip(128 + gx * j + gy * i + rng.integers(-3, 4, 64), 0, 255).astype(np.uint8))
    blocks = np.stack(blocks)
    dct = np.stack([ref.fast_dct(b) for b in blocks])
    assert np.array_equal(dct, np.round(dct))
    kat = {"blocks": blocks, "dct": dct.astype(np.int16)}
    for qf in (12, 50, 100):
        zz = np.stack([ref.quant_zigzag(d, qf) for d in dct])
        kat[f"zz_q{qf}"] = zz.astype(np.int16)
        if qf != 100:  # qf 100 on noise blocks has |level| >= 256 -> reference segfaults (vlc.c:349)
            for luma in (1, 0):
                blob, offs = pack_bits([ref.block_bits(luma, z) for z in zz])
                kat[f"bits_q{qf}_{'luma' if luma else 'chroma'}"] = blob
                kat[f"offs_q{qf}_{'luma' if luma else 'chroma'}"] = offs
    np.savez_compressed(os.path.join(HERE, "kat_blocks.npz"), **kat)

    # 2. hand-made coefficient arrays -> run-length pairs + block bits (VLC corner cases)
    cases = [np.zeros(64, np.int32)]
    for dcv in (1, -1, 31, -31, 32, 255, 256, 300, -300, 2042, -2042):
        z = np.zeros(64, np.int32); z[0] = dcv; cases.append(z)
    for pos in (1, 2, 3, 17, 32, 33, 63):
        for lv in (1, -1, 2, 3, 18, 39, 40, 41, 127, -127, 128, -128, 200, -200, 255, -255):
            z = np.zeros(64, np.int32); z[pos] = lv; cases.append(z)
    z = np.zeros(64, np.int32); z[[0, 2, 5, 6]] = [30, -2, 1, 9]; cases.append(z)
    cases.append(np.ones(64, np.int32))
    z = np.zeros(64, np.int32); z[1::2] = 255; cases.append(z)
    z = np.zeros(64, np.int32); z[0] = -5; z[2::2] = -200; cases.append(z)
    for _ in range(600):
        z = np.zeros(64, np.int32)
        mask = rng.random(64) < rng.choice([0.02, 0.1, 0.3, 0.6])
        mag = rng.choice([2, 5, 41, 130, 256])
        z[mask] = rng.integers(-mag + 1, mag, mask.sum())
        if rng.random() < 0.3:
            z[0] = 0
        cases.append(z)
    cases = np.stack(cases)
    vlc = {"zz": cases.astype(np.int16)}
    pairs = np.stack([ref.run_length(z)[:130] for z in cases])
    npairs = np.array([int(np.count_nonzero(z)) for z in cases])
    for i, n in enumerate(npairs):
        pairs[i, 2 * n + 2:] = 0  # beyond the (0,0) terminator the reference leaves garbage
    vlc["pairs"] = pairs.astype(np.int16)
    vlc["npairs"] = npairs.astype(np.int16)
    for luma in (1, 0):
        blob, offs = pack_bits([ref.block_bits(luma, z) for z in cases])
        vlc[f"bits_{'luma' if luma else 'chroma'}"] = blob
        vlc[f"offs_{'luma' if luma else 'chroma'}"] = offs
    np.savez_compressed(os.path.join(HERE, "kat_vlc.npz"), **vlc)

    # 3. scaled quantiser matrices for qf -1..102
    np.save(os.path.join(HERE, "qmatrix.npy"), np.stack([ref.scale_qmatrix(q) for q in range(-1, 103)]).astype(np.int16))

    # 4. colour conversion: SHA-256 of the exhaustive 2^24 table (r-major, then g, then b) + a sample
    hs = [hashlib.sha256() for _ in range(3)]
    g, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    for r0 in range(0, 256, 16):
        rgb = np.empty((16, 256, 256, 3), np.uint8)
        rgb[..., 0] = np.arange(r0, r0 + 16, dtype=np.uint8)[:, None, None]
        rgb[..., 1] = g
        rgb[..., 2] = b
        for h, plane in zip(hs, ref.convert(rgb)):
            h.update(plane.tobytes())
    sample = rng.integers(0, 256, (4096, 3), dtype=np.uint8)
    sy, scb, scr = ref.convert(sample)
    cb = rng.integers(0, 256, 64 * 48, dtype=np.uint8)
    cr = rng.integers(0, 256, 64 * 48, dtype=np.uint8)
    sub = ref.subsample(cb, cr, 64, 48)
    json.dump({"order": "for r in 0..255: for g in 0..255: for b in 0..255",
               "sha256": {"Y": hs[0].hexdigest(), "Cb": hs[1].hexdigest(), "Cr": hs[2].hexdigest()}},
              open(os.path.join(HERE, "colour_exhaustive.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(HERE, "kat_colour.npz"), rgb=sample, Y=sy, Cb=scb, Cr=scr,
                        sub_in_cb=cb, sub_in_cr=cr, sub_w=64, sub_h=48, sub_cb=sub[0], sub_cr=sub[1])

    # 5. end to end: BASELINE config 0 (4 x CIF random-pixel JPEGs), STRICT and FULL, qf 12
    noise = [rng.integers(0, 256, (288, 352, 3), dtype=np.uint8) for _ in range(4)]
    save_e2e("e2e_cif_noise.npz", run_folder(noise, [12], ["strict", "full"], jpeg_quality=95))

    # 6. quality sweep on smooth QCIF (noise at qf 100 makes the reference segfault)
    save_e2e("e2e_qcif_quality.npz", run_folder(smooth_frames(rng, 2, 176, 144), [1, 5, 12, 30, 49, 50, 75, 100],
                                                ["strict", "full"]))

    # 7. 300 frames from 3 distinct 96x144 pictures: hour (u8) wraps at 256, GOP keeps 5 bits
    base = smooth_frames(rng, 3, 96, 144)
    res = run_folder([base[i % 3] for i in range(300)], [12], ["strict"])
    distinct, index = [], []
    for f in res["frames"]:
        for k, dfr in enumerate(distinct):
            if np.array_equal(dfr, f):
                index.append(k)
                break
        else:
            distinct.append(f)
            index.append(len(distinct) - 1)
    res_small = dict(res, frames=distinct)
    save_e2e("e2e_300_wrap.npz", res_small, {"frame_index": np.array(index, np.int16)})

    # 8. odd geometry: 360x250 (W/2 even, H not a multiple of 16), 400x600 portrait (sample-data size)
    save_e2e("e2e_360x250.npz", run_folder(smooth_frames(rng, 2, 360, 250), [12], ["strict", "full"]))
    save_e2e("e2e_400x600.npz", run_folder(smooth_frames(rng, 1, 400, 600), [12], ["strict", "full"]))

    # 9. one 1080p picture (BASELINE configs 1-2 geometry), STRICT and FULL
    save_e2e("e2e_1080p.npz", run_folder(smooth_frames(rng, 1, 1920, 1080, amp=2), [12], ["strict", "full"], jpeg_quality=50))

    for f in sorted(os.listdir(HERE)):
        print(f"{os.path.getsize(os.path.join(HERE, f)):>10}  {f}")


if __name__ == "__main__":
    main()
