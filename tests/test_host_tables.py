"""Host-checkable arithmetic facts the kernels rely on."""
import numpy as np


def test_reciprocal_quantiser_is_exact():
    """k_encode_strips quantises with one fp32 multiply by rq = fl((1/d)(1+2^-20)) and a truncating
    convert.  Check trunc(n * rq) == trunc(n / d) (C division) for every divisor the scaled matrix can
    hold (1..4150 = round(83*50)) and every |n| < 2^15 (FDCT outputs of u8 blocks stay below 2^13)."""
    n = np.arange(0, 1 << 15, dtype=np.int32)
    nf = n.astype(np.float32)
    for d in range(1, 4151):
        rq = np.float32((1.0 / d) * (1.0 + 1.0 / 1048576.0))
        got = (nf * rq).astype(np.int32)          # fp32 multiply, truncation
        assert np.array_equal(got, n // d), d
        # negative side: fp32 multiply and C truncation are both sign-symmetric
        assert np.array_equal((-nf * rq).astype(np.int32), -(n // d)), d
        # the tile kernels multiply in round-toward-minus-infinity mode (m1v_kernels.hip, pixel_stage_rounds_down): the exact
        # product (24 x 24 bits: exact in float64) rounded DOWN to fp32, then truncated
        for sign in (1.0, -1.0):
            exact = (sign * nf).astype(np.float64) * np.float64(rq)
            down = exact.astype(np.float32)
            down = np.where(down.astype(np.float64) > exact, np.nextafter(down, np.float32(-np.inf)), down)
            assert np.array_equal(down.astype(np.int32), (sign * (n // d)).astype(np.int32)), (d, sign)


def test_fdct_output_range(orc):
    """|coefficient| < 2^13 for u8 input (needed by the int16 staging and the reciprocal quantiser)."""
    rng = np.random.default_rng(0)
    worst = 0
    blocks = [np.full(64, 255, np.uint8), np.zeros(64, np.uint8)]
    i, j = np.divmod(np.arange(64), 8)
    for u in range(8):
        for v in range(8):  # sign patterns of every basis function maximise that coefficient
            pat = np.cos((2 * i + 1) * u * np.pi / 16) * np.cos((2 * j + 1) * v * np.pi / 16)
            blocks.append(((pat > 0) * 255).astype(np.uint8))
            blocks.append(((pat < 0) * 255).astype(np.uint8))
    blocks += [rng.integers(0, 2, 64).astype(np.uint8) * 255 for _ in range(2000)]
    worst_ac = 0
    for b in blocks:
        d = orc.fdct(b)
        worst = max(worst, int(np.abs(d).max()))
        worst_ac = max(worst_ac, int(np.abs(d[1:]).max()))
    assert worst < (1 << 13), worst
    # The dense kernel stages AC levels as single bytes whenever 128 * (smallest AC divisor) >= 1024
    # (m1v_create): that needs every |AC coefficient| < 1024.  An AC coefficient is a zero-mean linear form of
    # the pixels with sum |b| <= 8 (Cauchy-Schwarz, orthonormal basis), so |AC| <= 127.5 * 8 + 2 (the
    # reference's rounding bias) = 1022, reached by the sign patterns of (0,4), (4,0), (4,4), which the integer
    # FDCT computes with adds and shifts only.
    assert worst_ac == 1022, worst_ac


def test_narrow_staging_threshold(orc):
    """Quality factors for which no AC level can reach +-128 (one byte per staged level is exact)."""
    for qf in range(1, 101):
        q = orc.scale_qmatrix(qf)
        narrow = int(q[1:].min()) >= 8
        assert narrow == (qf <= 76), qf


def test_colour_fast_path_proof_over_all_rgb_triples(tmp_path):
    """The kernel's fp32 colour fast path (value and uncertainty flag read from the float's bit pattern, +256 so the
    exponent is fixed) restated on the host and checked for all 2^24 triples x 3 components against the reference's
    fp64 expression: every pixel it does not flag must already be right (tools/colour_fast_proof.c, 0.5 s) — in round-to-nearest
    (run kernel, plane conversion) and in round-toward-minus-infinity (tile kernels)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "proof")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-frounding-math", os.path.join(root, "tools", "colour_fast_proof.c"), "-o", exe, "-lm"],
                   check=True)
    p = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0 and "wrong: 0" in p.stdout, p.stdout
    # the tile kernels evaluate the three fmas rounded toward minus infinity: same proof in that mode
    q = subprocess.run([exe, "down"], stdout=subprocess.PIPE, text=True)
    assert q.returncode == 0 and "wrong: 0" in q.stdout, q.stdout


def test_fp32_fdct_is_exact(tmp_path):
    """The kernel's FDCT runs in fp32 (ec504_imageencoder_amd/csrc/fdct_f32.h).  tools/fdct_f32_proof.cpp instantiates the
    same header (a) with a checked number type that computes every add / multiply / fma / floor exactly and fails if any
    result is not an fp32 value, (b) with float, compared with an integer restatement of image_processing.c:192-307 — on
    constant blocks, the sign patterns that extremise every linear form of both passes, and 100k random blocks; both forms of the two
    (x * 181) >> 17 outputs of the row pass (integer multiplier; one multiply rounded toward minus infinity)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fdct_proof")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", os.path.join(root, "ec504_imageencoder_amd", "csrc"),
                    os.path.join(root, "tools", "fdct_f32_proof.cpp"), "-o", exe], check=True)
    p = subprocess.run([exe, "100000"], stdout=subprocess.PIPE, text=True)
    assert p.returncode == 0 and "wrong coefficients 0  inexact operations 0" in p.stdout, p.stdout
