/* tools/colour_fast_proof.c — exhaustive host-side proof of the kernel's fp32 colour fast path
 * (m1v_kernels.hip: component_t / clear_fraction / convert_row).  For every (r,g,b) and each of Y, Cb, Cr:
 *     t    = fmaf(r, kr, fmaf(g, kg, fmaf(b, kb, k0 + 256 + eps)))          (same order, same constants)
 *     p    = t with its low 15 mantissa bits cleared (= 256 + trunc(t - 256));   d = t - p  (exact)
 * a pixel is "flagged" (redone in fp64 by the kernel) when d < LOW; for every unflagged pixel p - 256 must equal the
 * reference's fp64, unfused, left-to-right value truncated to int (image_processing.c:104-106), and t must lie in
 * [256, 512) for EVERY pixel (the bit trick assumes that exponent).
 *     gcc -O2 -ffp-contract=off -frounding-math tools/colour_fast_proof.c -o build/colour_fast_proof -lm && build/colour_fast_proof [down]
 */
#include <fenv.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define EPS 1.5e-4f
#define LOW (10.0f / 32768.0f) /* kFracLow */

int main(int argc, char **argv) {
    /* "down": the three fmas rounded toward minus infinity (the encode kernels' pixel stage runs in that mode, fdct_f32.h);
     * the reference's fp64 expression stays in round-to-nearest */
    const int down = argc > 1 && strcmp(argv[1], "down") == 0;
    const double k0d[3] = {0.0, 128.0, 128.0};
    const double krd[3] = {0.299, -0.168736, 0.5}, kgd[3] = {0.587, -0.331264, -0.418688}, kbd[3] = {0.114, 0.5, -0.081312};
    const float krf[3] = {0.299f, -0.168736f, 0.5f}, kgf[3] = {0.587f, -0.331264f, -0.418688f},
                kbf[3] = {0.114f, 0.5f, -0.081312f};
    long long flagged[3] = {0, 0, 0}, wrong = 0, outside = 0;
    for (int c = 0; c < 3; c++) {
        const float k0 = (float)k0d[c] + 256.0f + EPS;
        for (int r = 0; r < 256; r++)
            for (int g = 0; g < 256; g++)
                for (int b = 0; b < 256; b++) {
                    volatile double acc = k0d[c] + krd[c] * (double)r;
                    acc = acc + kgd[c] * (double)g;
                    acc = acc + kbd[c] * (double)b;
                    int want = (int)acc;
                    if (down) fesetround(FE_DOWNWARD);
                    volatile float t = fmaf((float)b, kbf[c], k0);
                    t = fmaf((float)g, kgf[c], t);
                    t = fmaf((float)r, krf[c], t);
                    if (down) fesetround(FE_TONEAREST);
                    uint32_t bits;
                    { float tt = t; memcpy(&bits, &tt, 4); }
                    bits &= 0xffff8000u;
                    float p;
                    memcpy(&p, &bits, 4);
                    if (t < 256.0f || t >= 512.0f) outside++;
                    if (t - p < LOW) {
                        flagged[c]++;
                        continue;
                    }
                    if ((int)(p - 256.0f) != want) {
                        if (wrong < 10)
                            printf("WRONG comp %d rgb %d %d %d: t=%.9g p-256=%d want %d\n", c, r, g, b, t, (int)(p - 256.0f), want);
                        wrong++;
                    }
                }
    }
    printf("flagged: Y %lld  Cb %lld  Cr %lld of 16777216 each; t outside [256, 512): %lld; wrong: %lld\n",
           flagged[0], flagged[1], flagged[2], outside, wrong);
    return wrong != 0 || outside != 0;
}
