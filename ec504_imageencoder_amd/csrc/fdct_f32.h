// ec504_imageencoder_amd/csrc/fdct_f32.h — the reference's integer FDCT (image_processing.c:192-307) restated in
// fp32 arithmetic that is EXACT, shared by the device code (m1v_kernels.hip) and the host-side proof
// (tools/fdct_f32_proof.cpp, tests/test_host_tables.py).
//
// Why floats for an integer transform: the round-1 butterflies were built on v_mad_i32_i24 / v_mul_i32_i24, and a 24-bit
// integer multiply costs 20-45 cycles inside a mixed instruction stream on gfx950 (tools/ubench/stream_probe.hip,
// profiles/r02_stream_probe.txt) where a float fma costs ~2.  The same butterfly network in fp32 measures 22 % fewer cycles
// in a synthetic stream of the kernel's instruction mix (tools/ubench/synth_rows.hip), and 4 % in the kernel (x1.105 ->
// x1.148 of round 1, profiles/r02_ab_history.txt).
//
// Why it is exact: every value the network holds is  N * 2^-s  with an integer |N| < 2^24 (s = 0, 10 or 13), so
// every add, multiply and fma below returns its exact result without rounding:
//   * inputs of the row pass are raw pixels 256 + u8; every register of the butterfly is a linear form in the pixel
//     DIFFERENCES (the bias cancels) with coefficient sum <= 3710, i.e. |N| <= 3710 * 255 < 2^20, except the two plain
//     sums t0 (<= 8 * 511) and t1;
//   * inputs of the column pass are row-pass outputs, |v| <= 2040 after the bias cancels (row 0: sums of eight u8;
//     others <= 1020): |N| <= 3710 * 2040 < 2^23;
//   * scaled constants c * 2^-s are exact (c < 2^11) and a product of an exact constant and an exact input is an
//     exact real the fma adds to the third operand before its single rounding — which does not round, because the sum
//     is again N * 2^-s with |N| < 2^24.
//   floor(N * 2^-s) is the reference's arithmetic right shift.  The two row outputs (t * 181) >> 17 exceed 24 bits in the
//   product: see fdct_row_f for the two forms (integer multiplier, or one multiply rounded toward minus infinity).  tools/fdct_f32_proof.cpp instantiates the network with a checked
//   number type (every operation computed exactly, failing if a result is not an fp32 value) on worst-case and random
//   blocks, and compares the float instantiation with the integer network.
#pragma once
#include <math.h>

#ifndef M1V_HD
#define M1V_HD inline
#endif

namespace m1vf {

// The network is written over a number type F: float on the device; the proof instantiates it a second time with a
// checked type that computes every operation exactly and fails if a result is not representable in fp32.
M1V_HD float fma_(float a, float b, float c) { return fmaf(a, b, c); }
M1V_HD float floor_(float a) { return floorf(a); }
M1V_HD int to_int(float a) { return (int)a; }
// a * k rounded toward minus infinity.  Device: a plain multiply — the caller's wave runs its fp32 arithmetic in that mode
// (pixel_stage_rounding() in m1v_kernels.hip, first statement of every kernel that instantiates fdct_row_f<F, true>).
M1V_HD float mul_down(float a, float k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return a * k;
#else
    const double p = (double)a * (double)k; // exact: 24 x 24 bits
    float f = (float)p;
    if ((double)f > p) f = nextafterf(f, -INFINITY);
    return f;
#endif
}
M1V_HD int mulhi_(int a, int b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __mulhi(a, b);
#else
    return (int)(((long long)a * (long long)b) >> 32);
#endif
}

// raw pixels carry this bias through the transform; only the DC sum keeps it (64 * kPxBiasF, removed in fdct_col_f)
constexpr float kPxBiasF = 256.0f;

// One 8-point pass.  t[0], t[1] unscaled; t[2..7] scaled by 2^-SH.  ROUND (column pass): + 2.0 = (16384 >> 13) on
// t[2..5], the rounding offset the reference adds before its >> 13.
template <int SH, bool ROUND, typename F>
M1V_HD void butterfly8f(const F v0, const F v1, const F v2, const F v3, const F v4, const F v5, const F v6, const F v7,
                        F t[8]) {
    constexpr float K = 1.0f / (float)(1 << SH);
    const F c1 = F(1004.0f * K), s1 = F(200.0f * K), c3 = F(851.0f * K), s3 = F(569.0f * K), r2c6 = F(554.0f * K),
            r2s6 = F(1337.0f * K), two = F(2.0f);
    const F a0 = v0 + v7, d0 = v0 - v7;
    const F a1 = v1 + v6, d1 = v1 - v6;
    const F a2 = v2 + v5, d2 = v2 - v5;
    const F a3 = v3 + v4, d3 = v3 - v4;
    const F e0 = a0 + a3, e3 = a0 - a3;
    const F e1 = a1 + a2, e2 = a1 - a2;
    t[0] = e0 + e1; // x6
    t[1] = e0 - e1; // x4
    const F e23 = e2 + e3;
    const F m78 = ROUND ? fma_(r2c6, e23, two) : r2c6 * e23;
    t[3] = fma_(-(r2s6 + r2c6), e2, m78); // x7
    t[2] = fma_(r2s6 - r2c6, e3, m78);    // x8
    const F d12 = d1 + d2, d03 = d0 + d3;
    const F m12 = c1 * d12;
    const F f2 = fma_(-(s1 + c1), d2, m12);
    const F f1 = fma_(s1 - c1, d1, m12);
    const F m03 = c3 * d03;
    const F f3 = fma_(-(s3 + c3), d3, m03);
    const F f0 = fma_(s3 - c3, d0, m03);
    const F g5 = f0 + f2, g0 = f0 - f2;
    const F g2 = f3 + f1, g3 = f3 - f1;
    const F g2r = ROUND ? g2 + two : g2;
    t[4] = g2r - g5;
    t[5] = g2r + g5;
    t[6] = g3;
    t[7] = g0;
}

// Row pass (image_processing.c:198-250): 8 raw pixels (kPxBiasF + value) -> out[0..7], integers held in floats.
// out[0] carries 8 * kPxBiasF.
// The two outputs (x * 181) >> 17 need 28 bits in the product:
//   DOWN = false  through the integer multiplier: the high half of x * (181 << 15) = (x * 181 * 2^15) >> 32 is the product and
//                 the shift in one v_mul_hi_i32 (a v_mul_i32_i24 inside a stream of float instructions costs 20-45 cycles on
//                 gfx950, tools/ubench/stream_probe.hip; v_mul_lo_u32 + shift measured 3 % slower than this in the kernel);
//   DOWN = true   x * 181 / 2^17 = t * (181 / 128) as ONE float multiply rounded TOWARD MINUS INFINITY: the rounded product
//                 cannot fall below the integer under the exact value (that integer is a float <= the exact value) and is
//                 never above the exact value, so its floor is the exact value's floor.  Needs the wave in that rounding mode;
//                 every other operation of both passes is exact and does not care.  Two conversions and the integer multiply
//                 per output less (profiles/r03_ab_history.txt).
template <typename F, bool DOWN = false>
M1V_HD void fdct_row_f(const F p[8], F out[8]) {
    F t[8];
    butterfly8f<10, false, F>(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], t);
    out[0] = t[0];
    out[4] = t[1];
    out[2] = floor_(t[2]); // x8 >> 10
    out[6] = floor_(t[3]); // x7 >> 10
    out[7] = floor_(t[4]); // (x2 - x5) >> 10
    out[1] = floor_(t[5]); // (x2 + x5) >> 10
    if (DOWN) {
        out[3] = floor_(mul_down(t[6], F(181.0f / 128.0f)));
        out[5] = floor_(mul_down(t[7], F(181.0f / 128.0f)));
    } else {
        out[3] = F((float)mulhi_(to_int(t[6] * F(1024.0f)), 181 << 15));
        out[5] = F((float)mulhi_(to_int(t[7] * F(1024.0f)), 181 << 15));
    }
}

// Column pass (image_processing.c:253-305): rows[0..7][i] -> dct_block[0..7][i].  dc_bias8 = (what the column's plain
// sum carries on top of the reference's value) / 8: 8 * kPxBiasF for column 0 of a block of raw pixels, else 0.
template <typename F>
M1V_HD void fdct_col_f(const F r0, const F r1, const F r2, const F r3, const F r4, const F r5, const F r6, const F r7,
                       F c[8], const float dc_bias8) {
    F t[8];
    butterfly8f<13, true, F>(r0, r1, r2, r3, r4, r5, r6, r7, t);
    c[0] = floor_(fma_(t[0], F(0.125f), F(2.0f - dc_bias8))); // (x6 + 16) >> 3
    c[4] = floor_(fma_(t[1], F(0.125f), F(2.0f)));            // (x4 + 16) >> 3
    c[2] = floor_(t[2]);                                      // (x8 + 16384) >> 13
    c[6] = floor_(t[3]);
    c[7] = floor_(t[4]);
    c[1] = floor_(t[5]);
    // ((x >> 8) * 181 + 8192) >> 12: x >> 8 = floor(t * 2^-13 * 32); (A * 181 + 8192) < 2^22 is exact in the fma
    c[3] = floor_(fma_(floor_(t[6] * F(32.0f)), F(181.0f / 4096.0f), F(2.0f)));
    c[5] = floor_(fma_(floor_(t[7] * F(32.0f)), F(181.0f / 4096.0f), F(2.0f)));
}

} // namespace m1vf
