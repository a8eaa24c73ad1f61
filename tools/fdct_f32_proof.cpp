// tools/fdct_f32_proof.cpp — host-side check that the fp32 FDCT of ec504_imageencoder_amd/csrc/fdct_f32.h is the
// reference's integer FDCT (image_processing.c:192-307), bit for bit.
//   1. the network instantiated with a CHECKED number type: every add / multiply / fma / floor is computed exactly (in
//      double, whose 53 bits hold every product and sum that occurs) and the run fails if any result is not an fp32
//      value — i.e. if the float instantiation could have rounded anywhere;
//   2. the float instantiation against an integer restatement of the reference's two passes.
// Blocks: constant, the +-sign patterns of all 64 basis functions (they maximise each coefficient), their row/column
// restrictions, random 0/255 blocks, random bytes.
//     g++ -O2 -ffp-contract=off -I ec504_imageencoder_amd/csrc tools/fdct_f32_proof.cpp -o build/fdct_f32_proof && build/fdct_f32_proof
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

static long long g_inexact = 0;
static double g_max_abs = 0;
struct Chk {
    double v;
    Chk() : v(0) {}
    explicit Chk(float f) : v(f) {}
    static Chk make(double r, const char *op) {
        if ((double)(float)r != r) {
            if (g_inexact < 10) printf("INEXACT %s: %.17g is not an fp32 value\n", op, r);
            g_inexact++;
        }
        if (fabs(r) > g_max_abs) g_max_abs = fabs(r);
        Chk c;
        c.v = r;
        return c;
    }
};
static inline Chk operator+(Chk a, Chk b) { return Chk::make(a.v + b.v, "add"); }
static inline Chk operator-(Chk a, Chk b) { return Chk::make(a.v - b.v, "sub"); }
static inline Chk operator*(Chk a, Chk b) { return Chk::make(a.v * b.v, "mul"); }
static inline Chk operator-(Chk a) { Chk c; c.v = -a.v; return c; }
namespace m1vf {
static inline Chk fma_(Chk a, Chk b, Chk c) { return Chk::make(a.v * b.v + c.v, "fma"); } // exact: < 2^53
static inline Chk floor_(Chk a) { return Chk::make(floor(a.v), "floor"); }
static inline int to_int(Chk a) {
    if (a.v != floor(a.v)) { printf("to_int of a non-integer %.17g\n", a.v); g_inexact++; }
    return (int)a.v;
}
// fdct_row_f<F, true>: the product taken in round-toward-minus-infinity mode = the exact product, rounded down to fp32
static inline Chk mul_down(Chk a, Chk k) {
    const double r = a.v * k.v;
    float f = (float)r;
    if ((double)f > r) f = nextafterf(f, -INFINITY);
    Chk c;
    c.v = f;
    return c;
}
} // namespace m1vf
#include "fdct_f32.h"

// ---- integer restatement of image_processing.c:192-307 (row pass :198-250, column pass :253-305) ----
static void butterfly_int(const int v[8], int t[8]) {
    const int c1 = 1004, s1 = 200, c3 = 851, s3 = 569, r2c6 = 554, r2s6 = 1337;
    int x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3], x4 = v[4], x5 = v[5], x6 = v[6], x7 = v[7], x8;
    // stage 1
    x8 = x7 + x0; x0 -= x7; x7 = x1 + x6; x1 -= x6; x6 = x2 + x5; x2 -= x5; x5 = x3 + x4; x3 -= x4;
    // stage 2
    x4 = x8 + x5; x8 -= x5; x5 = x7 + x6; x7 -= x6;
    x6 = c1 * (x1 + x2); x2 = (-s1 - c1) * x2 + x6; x1 = (s1 - c1) * x1 + x6;
    x6 = c3 * (x0 + x3); x3 = (-s3 - c3) * x3 + x6; x0 = (s3 - c3) * x0 + x6;
    // stage 3
    x6 = x4 + x5; x4 -= x5;
    x5 = r2c6 * (x7 + x8); x7 = (-r2s6 - r2c6) * x7 + x5; x8 = (r2s6 - r2c6) * x8 + x5;
    x5 = x0 + x2; x0 -= x2; x2 = x3 + x1; x3 -= x1;
    t[0] = x6; t[1] = x4; t[2] = x8; t[3] = x7; t[4] = x2 - x5; t[5] = x2 + x5; t[6] = x3; t[7] = x0;
}
static void fdct_int(const unsigned char px[64], int out[64]) {
    const int r2 = 181;
    int rows[64];
    for (int i = 0; i < 8; i++) {
        int v[8], t[8];
        for (int j = 0; j < 8; j++) v[j] = px[i * 8 + j];
        butterfly_int(v, t);
        int *o = &rows[i * 8];
        o[0] = t[0]; o[4] = t[1]; o[2] = t[2] >> 10; o[6] = t[3] >> 10; o[7] = t[4] >> 10; o[1] = t[5] >> 10;
        o[3] = (t[6] * r2) >> 17; o[5] = (t[7] * r2) >> 17;
    }
    for (int i = 0; i < 8; i++) {
        int v[8], t[8];
        for (int r = 0; r < 8; r++) v[r] = rows[r * 8 + i];
        butterfly_int(v, t);
        out[0 * 8 + i] = (t[0] + 16) >> 3; out[4 * 8 + i] = (t[1] + 16) >> 3;
        out[2 * 8 + i] = (t[2] + 16384) >> 13; out[6 * 8 + i] = (t[3] + 16384) >> 13;
        out[7 * 8 + i] = (t[4] + 16384) >> 13; out[1 * 8 + i] = (t[5] + 16384) >> 13;
        out[3 * 8 + i] = ((t[6] >> 8) * r2 + 8192) >> 12; out[5 * 8 + i] = ((t[7] >> 8) * r2 + 8192) >> 12;
    }
}

template <typename F, bool DOWN>
static void fdct_f(const unsigned char px[64], double out[64], double (*val)(F)) {
    F rows[64];
    for (int i = 0; i < 8; i++) {
        F p[8];
        for (int j = 0; j < 8; j++) p[j] = F(m1vf::kPxBiasF + (float)px[i * 8 + j]);
        m1vf::fdct_row_f<F, DOWN>(p, &rows[i * 8]);
    }
    for (int i = 0; i < 8; i++) {
        F c[8];
        m1vf::fdct_col_f<F>(rows[0 * 8 + i], rows[1 * 8 + i], rows[2 * 8 + i], rows[3 * 8 + i], rows[4 * 8 + i], rows[5 * 8 + i],
                            rows[6 * 8 + i], rows[7 * 8 + i], c, i == 0 ? 8.0f * m1vf::kPxBiasF : 0.0f);
        for (int u = 0; u < 8; u++) out[u * 8 + i] = val(c[u]);
    }
}
static double val_f(float x) { return x; }
static double val_c(Chk x) { return x.v; }

static long long g_blocks = 0, g_wrong = 0;
static void check(const unsigned char px[64]) {
    int want[64];
    double got_f[64], got_c[64], got_fd[64], got_cd[64];
    fdct_int(px, want);
    fdct_f<float, false>(px, got_f, val_f);   // both forms of the two (x * 181) >> 17 outputs (fdct_row_f)
    fdct_f<Chk, false>(px, got_c, val_c);
    fdct_f<float, true>(px, got_fd, val_f);
    fdct_f<Chk, true>(px, got_cd, val_c);
    g_blocks++;
    for (int k = 0; k < 64; k++)
        if (got_f[k] != (double)want[k] || got_c[k] != (double)want[k] || got_fd[k] != (double)want[k] || got_cd[k] != (double)want[k]) {
            if (g_wrong < 10)
                printf("WRONG coefficient %d: int %d float %.9g checked %.9g, rounded-down form %.9g checked %.9g\n", k, want[k], got_f[k], got_c[k],
                       got_fd[k], got_cd[k]);
            g_wrong++;
        }
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}

int main(int argc, char **argv) {
    long n_random = argc > 1 ? atol(argv[1]) : 200000;
    unsigned char px[64];
    for (int v = 0; v < 256; v += 51) { for (int k = 0; k < 64; k++) px[k] = (unsigned char)v; check(px); }
    for (int u = 0; u < 8; u++)
        for (int w = 0; w < 8; w++)
            for (int lo = 0; lo < 2; lo++)
                for (int mode = 0; mode < 3; mode++) { // 2-D basis sign pattern, rows only, columns only
                    for (int i = 0; i < 8; i++)
                        for (int j = 0; j < 8; j++) {
                            double a = cos((2 * i + 1) * u * M_PI / 16), b = cos((2 * j + 1) * w * M_PI / 16);
                            double s = mode == 0 ? a * b : (mode == 1 ? a : b);
                            px[i * 8 + j] = ((s > 0) != (lo != 0)) ? 255 : 0;
                        }
                    check(px);
                }
    // vertices of the input cube that extremise the linear forms: every 0/255 pattern of a row (row pass, exhaustive),
    // and for each row-pass output i the blocks whose rows are the pattern maximising or minimising out[i], in all 256
    // max/min combinations over the eight rows (column pass)
    for (int pat = 0; pat < 256; pat++) {
        for (int i = 0; i < 8; i++)
            for (int j = 0; j < 8; j++) px[i * 8 + j] = ((pat >> j) & 1) ? 255 : 0;
        check(px);
    }
    for (int i = 0; i < 8; i++) {
        int best[2] = {0, 0}, bestv[2] = {-(1 << 30), 1 << 30};
        for (int pat = 0; pat < 256; pat++) {
            unsigned char row[64] = {0};
            int out[64];
            for (int j = 0; j < 8; j++) row[j] = ((pat >> j) & 1) ? 255 : 0; // only row 0 matters for the row pass of row 0
            // row-pass output i of this pattern, through the integer restatement
            int v[8], t[8];
            for (int j = 0; j < 8; j++) v[j] = row[j];
            butterfly_int(v, t);
            int o[8] = {t[0], t[5] >> 10, t[2] >> 10, (t[6] * 181) >> 17, t[1], (t[7] * 181) >> 17, t[3] >> 10, t[4] >> 10};
            (void)out;
            if (o[i] > bestv[0]) { bestv[0] = o[i]; best[0] = pat; }
            if (o[i] < bestv[1]) { bestv[1] = o[i]; best[1] = pat; }
        }
        for (int sel = 0; sel < 256; sel++) {
            for (int r = 0; r < 8; r++)
                for (int j = 0; j < 8; j++) px[r * 8 + j] = ((best[(sel >> r) & 1] >> j) & 1) ? 255 : 0;
            check(px);
        }
    }
    for (long n = 0; n < n_random; n++) {
        int kind = n % 4;
        for (int k = 0; k < 64; k++) {
            uint32_t r = rnd();
            px[k] = kind == 0 ? (unsigned char)r : kind == 1 ? ((r & 1) ? 255 : 0) : kind == 2 ? (unsigned char)(r % 3 * 127 + (r >> 8) % 2)
                                                                                              : (unsigned char)(128 + (int)(r % 9) - 4 + ((k & 9) ? 100 : -100));
        }
        check(px);
    }
    printf("blocks %lld  wrong coefficients %lld  inexact operations %lld  largest |value| %.0f (2^24 = 16777216)\n", g_blocks,
           g_wrong, g_inexact, g_max_abs);
    return (g_wrong || g_inexact) ? 1 : 0;
}
