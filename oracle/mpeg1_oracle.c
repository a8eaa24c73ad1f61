/* oracle/mpeg1_oracle.c — TEST INFRASTRUCTURE ONLY (see mpeg1_oracle.h).
 *
 * CPU restatement of the reference's MPEG-1 I-frame path.  Every function cites the reference
 * file:line (under /root/reference) whose observable behaviour it restates.  The code here is
 * written from scratch: tables are stored as (code, length) integers, the bit buffer is a plain
 * append-only writer, the quantiser matrix is scaled once per call and nothing prints or leaks.
 * Built with -ffp-contract=off: the colour conversion must not be fused (SURVEY §7 hard part 2).
 */
#define _GNU_SOURCE
#include "mpeg1_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Tables                                                                                     */
/* ------------------------------------------------------------------------------------------ */

/* Default intra matrix, image_processing.c:17-26 (ISO 11172-2 default). */
static const uint8_t kIntraQ[64] = {
     8, 16, 19, 22, 26, 27, 29, 34,   16, 16, 22, 24, 27, 29, 34, 37,
    19, 22, 26, 27, 29, 34, 34, 38,   22, 22, 26, 27, 29, 34, 37, 40,
    22, 26, 27, 29, 32, 35, 40, 48,   26, 27, 29, 32, 35, 40, 48, 58,
    26, 27, 29, 34, 38, 46, 56, 69,   27, 29, 35, 38, 46, 56, 69, 83};

/* Scan position of natural-order coefficient [i][j], image_processing.c:28-37. */
static const uint8_t kScanPos[64] = {
     0,  1,  5,  6, 14, 15, 27, 28,    2,  4,  7, 13, 16, 26, 29, 42,
     3,  8, 12, 17, 25, 30, 41, 43,    9, 11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54,   20, 22, 33, 38, 46, 51, 55, 60,
    21, 34, 37, 47, 50, 56, 59, 61,   35, 36, 48, 49, 57, 58, 62, 63};

/* DC size codes (code, bits) for sizes 0..8: vlc.c:121-131 (luma), :134-144 (chroma). */
static const uint8_t kDcLumaCode[9]   = {0x4, 0x0, 0x1, 0x5, 0x6, 0xE, 0x1E, 0x3E, 0x7E};
static const uint8_t kDcLumaBits[9]   = {3, 2, 2, 3, 3, 4, 5, 6, 7};
static const uint8_t kDcChromaCode[9] = {0x0, 0x1, 0x2, 0x6, 0xE, 0x1E, 0x3E, 0x7E, 0xFE};
static const uint8_t kDcChromaBits[9] = {2, 2, 2, 3, 4, 5, 6, 7, 8};

/* Run/level codes WITHOUT sign bit, vlc.c:176-288, in the reference's order (run-major, then
 * level).  Row r of the table starts at kAcFirst[r] (vlc.c:172-174); row 0 starts at level 2,
 * every other row at level 1.  Entry (16,2) is the reference's 15-bit deviation (vlc.c:271). */
static const uint8_t kAcFirst[33] = {
    0, 39, 57, 62, 66, 69, 72, 75, 77, 79, 81, 83, 85, 87, 89, 91, 93,
    95, 96, 97, 98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110};
static const uint8_t kAcCode[110] = {
    0x04, 0x05, 0x06, 0x26, 0x21, 0x0a, 0x1d, 0x18, 0x13, 0x10, 0x1a, 0x19,
    0x18, 0x17, 0x1f, 0x1e, 0x1d, 0x1c, 0x1b, 0x1a, 0x19, 0x18, 0x17, 0x16,
    0x15, 0x14, 0x13, 0x12, 0x11, 0x10, 0x18, 0x17, 0x16, 0x15, 0x14, 0x13,
    0x12, 0x11, 0x10, 0x03, 0x06, 0x25, 0x0c, 0x1b, 0x16, 0x15, 0x1f, 0x1e,
    0x1d, 0x1c, 0x1b, 0x1a, 0x19, 0x13, 0x12, 0x11, 0x10, 0x05, 0x04, 0x0b,
    0x14, 0x14, 0x07, 0x24, 0x1c, 0x13, 0x06, 0x0f, 0x12, 0x07, 0x09, 0x12,
    0x05, 0x1e, 0x14, 0x04, 0x15, 0x07, 0x11, 0x05, 0x11, 0x27, 0x10, 0x23,
    0x1a, 0x22, 0x19, 0x20, 0x18, 0x0e, 0x17, 0x0d, 0x16, 0x08, 0x15, 0x1f,
    0x1a, 0x19, 0x17, 0x16, 0x1f, 0x1e, 0x1d, 0x1c, 0x1b, 0x1f, 0x1e, 0x1d,
    0x1c, 0x1b};
static const uint8_t kAcBits[110] = {
    4, 5, 7, 8, 8, 10, 12, 12, 12, 12, 13, 13, 13, 13, 14, 14, 14, 14, 14, 14, 14, 14,
    14, 14, 14, 14, 14, 14, 14, 14, 15, 15, 15, 15, 15, 15, 15, 15, 15, 3, 6, 8, 10, 12,
    13, 13, 15, 15, 15, 15, 15, 15, 15, 16, 16, 16, 16, 4, 7, 10, 12, 13, 5, 8, 12, 13,
    5, 10, 12, 6, 10, 13, 6, 12, 16, 6, 12, 7, 12, 7, 13, 8, 13, 8, 16, 8, 16, 8,
    16, 10, 16, 10, 16, 10, 15, 12, 12, 12, 12, 12, 13, 13, 13, 13, 13, 16, 16, 16, 16, 16};

/* ------------------------------------------------------------------------------------------ */
/* Bit buffer                                                                                 */
/* ------------------------------------------------------------------------------------------ */

void orc_bits_init(orc_bits *b) {
    b->cap_bytes = 4096;
    b->buf = (uint8_t *)calloc(b->cap_bytes, 1);
    b->nbits = 0;
}

void orc_bits_free(orc_bits *b) {
    free(b->buf);
    b->buf = NULL;
    b->cap_bytes = b->nbits = 0;
}

/* Append the low n bits of value, most significant first.  Equivalent to the reference's
 * bitvector_put_bit / put_byte_off / concat chains (bit_vector.c:13-115): every one of them
 * appends at the cursor, which always equals the fill level on this path. */
void orc_bits_put(orc_bits *b, uint32_t value, int n) {
    if (n <= 0) return;
    size_t need = (b->nbits + (size_t)n + 7) >> 3;
    if (need + 8 > b->cap_bytes) {
        size_t ncap = b->cap_bytes * 2;
        while (need + 8 > ncap) ncap *= 2;
        b->buf = (uint8_t *)realloc(b->buf, ncap);
        memset(b->buf + b->cap_bytes, 0, ncap - b->cap_bytes);
        b->cap_bytes = ncap;
    }
    for (int k = n - 1; k >= 0; k--) {
        if ((value >> k) & 1u) b->buf[b->nbits >> 3] |= (uint8_t)(0x80u >> (b->nbits & 7));
        b->nbits++;
    }
}

static void bits_align_zero(orc_bits *b) { /* encoder.h:442-443 */
    while (b->nbits & 7) orc_bits_put(b, 0, 1);
}

/* ------------------------------------------------------------------------------------------ */
/* Pixel stage                                                                                */
/* ------------------------------------------------------------------------------------------ */

void orc_region(int mode, int W, int H, int *x_extent, int *y_extent) {
    if (mode == ORC_MODE_FULL) {
        *x_extent = W & ~15;
        *y_extent = H & ~15;
    } else { /* encoder.h:238 "96", :248 "144" */
        *x_extent = 96;
        *y_extent = 144;
    }
}

/* image_processing.c:104-106 — fp64, evaluated left to right, never fused, truncated to u8. */
void orc_convert_rgb_to_ycbcr(const uint8_t *data, int channels, size_t npx,
                              uint8_t *Y, uint8_t *Cb, uint8_t *Cr) {
    for (size_t i = 0; i < npx; i++) {
        const uint8_t *p = data + i * (size_t)channels;
        double r = p[0], g = p[1], b = p[2];
        double y = 0.299 * r;
        y = y + 0.587 * g;
        y = y + 0.114 * b;
        double cb = 128 - 0.168736 * r;
        cb = cb - 0.331264 * g;
        cb = cb + 0.5 * b;
        double cr = 128 + 0.5 * r;
        cr = cr - 0.418688 * g;
        cr = cr - 0.081312 * b;
        Y[i] = (uint8_t)y;
        Cb[i] = (uint8_t)cb;
        Cr[i] = (uint8_t)cr;
    }
}

/* image_processing.c:114-133 — integer mean of each 2x2, truncated. */
void orc_subsample_420(const uint8_t *Cb, const uint8_t *Cr, int W, int H,
                       uint8_t *Cb_sub, uint8_t *Cr_sub) {
    int sw = W / 2;
    for (int y = 0; y + 1 < H; y += 2) {
        for (int x = 0; x + 1 < W; x += 2) {
            size_t a = (size_t)y * W + x, c = (size_t)(y + 1) * W + x;
            size_t o = (size_t)(y / 2) * sw + (x / 2);
            Cb_sub[o] = (uint8_t)((Cb[a] + Cb[a + 1] + Cb[c] + Cb[c + 1]) / 4);
            Cr_sub[o] = (uint8_t)((Cr[a] + Cr[a + 1] + Cr[c] + Cr[c + 1]) / 4);
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* Block stage                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* The 8-point butterfly network shared by both passes of fast_DCT (image_processing.c:208-239 and
 * :263-294; constants :40-46).  v[0..7] in, the nine intermediates t[] out:
 *   t0=x6 t1=x4 t2=x8 t3=x7 t4=x2-x5 t5=x2+x5 t6=x3 t7=x0   (names as in the reference). */
static inline void butterfly8(const int32_t v[8], int32_t t[8]) {
    enum { c1 = 1004, s1 = 200, c3 = 851, s3 = 569, r2c6 = 554, r2s6 = 1337 };
    int32_t a0 = v[0] + v[7], d0 = v[0] - v[7];
    int32_t a1 = v[1] + v[6], d1 = v[1] - v[6];
    int32_t a2 = v[2] + v[5], d2 = v[2] - v[5];
    int32_t a3 = v[3] + v[4], d3 = v[3] - v[4];

    int32_t e0 = a0 + a3, e3 = a0 - a3;          /* x4, x8 after stage 2 */
    int32_t e1 = a1 + a2, e2 = a1 - a2;          /* x5, x7 after stage 2 */
    int32_t m12 = c1 * (d1 + d2);
    int32_t f2 = (-s1 - c1) * d2 + m12;          /* x2 */
    int32_t f1 = (s1 - c1) * d1 + m12;           /* x1 */
    int32_t m03 = c3 * (d0 + d3);
    int32_t f3 = (-s3 - c3) * d3 + m03;          /* x3 */
    int32_t f0 = (s3 - c3) * d0 + m03;           /* x0 */

    t[0] = e0 + e1;                              /* x6 */
    t[1] = e0 - e1;                              /* x4 */
    int32_t m78 = r2c6 * (e2 + e3);
    t[3] = (-r2s6 - r2c6) * e2 + m78;            /* x7 */
    t[2] = (r2s6 - r2c6) * e3 + m78;             /* x8 */
    int32_t g5 = f0 + f2, g0 = f0 - f2;          /* x5, x0 */
    int32_t g2 = f3 + f1, g3 = f3 - f1;          /* x2, x3 */
    t[4] = g2 - g5;
    t[5] = g2 + g5;
    t[6] = g3;
    t[7] = g0;
}

void orc_fdct(const uint8_t in[64], int32_t out[64]) {
    enum { r2 = 181 };
    int32_t rows[64], v[8], t[8];
    for (int i = 0; i < 8; i++) { /* image_processing.c:198-250 */
        for (int j = 0; j < 8; j++) v[j] = in[i * 8 + j];
        butterfly8(v, t);
        rows[i * 8 + 0] = t[0];
        rows[i * 8 + 4] = t[1];
        rows[i * 8 + 2] = t[2] >> 10;
        rows[i * 8 + 6] = t[3] >> 10;
        rows[i * 8 + 7] = t[4] >> 10;
        rows[i * 8 + 1] = t[5] >> 10;
        rows[i * 8 + 3] = (t[6] * r2) >> 17;
        rows[i * 8 + 5] = (t[7] * r2) >> 17;
    }
    for (int i = 0; i < 8; i++) { /* image_processing.c:253-305 */
        for (int j = 0; j < 8; j++) v[j] = rows[j * 8 + i];
        butterfly8(v, t);
        out[0 * 8 + i] = (t[0] + 16) >> 3;
        out[4 * 8 + i] = (t[1] + 16) >> 3;
        out[2 * 8 + i] = (t[2] + 16384) >> 13;
        out[6 * 8 + i] = (t[3] + 16384) >> 13;
        out[7 * 8 + i] = (t[4] + 16384) >> 13;
        out[1 * 8 + i] = (t[5] + 16384) >> 13;
        out[3 * 8 + i] = ((t[6] >> 8) * r2 + 8192) >> 12;
        out[5 * 8 + i] = ((t[7] >> 8) * r2 + 8192) >> 12;
    }
}

/* image_processing.c:314-343.  The scale factor is a float; Q*sf is int*float -> float, the
 * division by 100.0 promotes to double, round() is half-away-from-zero. */
void orc_scale_qmatrix(int quality_factor, int32_t q[64]) {
    if (quality_factor < 1) quality_factor = 1;
    if (quality_factor > 100) quality_factor = 100;
    float sf;
    if (quality_factor < 50) sf = (float)(5000.0 / quality_factor);
    else sf = (float)(200.0 - 2 * quality_factor);
    for (int k = 0; k < 64; k++) {
        float prod = (float)kIntraQ[k] * sf;
        int v = (int)round((double)prod / 100.0);
        q[k] = v < 1 ? 1 : v;
    }
}

/* image_processing.c:367 `(int)(round(d)/q)` with d integral == C truncating division; then the
 * scatter of image_processing.c:373-381.  equalize_coefficients (:385-398) is the identity. */
void orc_quant_zigzag(const int32_t dct[64], const int32_t q[64], int32_t zz[64]) {
    for (int k = 0; k < 64; k++) zz[kScanPos[k]] = dct[k] / q[k];
}

int orc_run_length(const int32_t zz[64], int32_t pairs[130]) {
    int n = 0, zeros = 0;
    for (int i = 0; i < 64; i++) {
        if (zz[i] != 0) {
            pairs[2 * n] = zz[i];
            pairs[2 * n + 1] = zeros;
            zeros = 0;
            n++;
        } else {
            zeros++;
        }
    }
    pairs[2 * n] = 0; /* image_processing.c:748: gcc stores [n]=0 and [n+1]=0 */
    pairs[2 * n + 1] = 0;
    return n;
}

/* vlc.c:315-385 with first == 0 (image_processing.c:411-416 always passes 0).
 * `run` is the reference's argument (zeros before the coefficient, >= 1 here). */
static int put_ac(orc_bits *out, int run, int level) {
    int negative = level < 0;
    int L = negative ? -level : level;
    int r = run - 1;                               /* vlc.c:326 */
    if (r == 0 && L == 1) {                        /* vlc.c:329-334: "11" */
        orc_bits_put(out, 0x3, 2);
        return ORC_OK;
    }
    if (r <= 31) {                                 /* vlc.c:335-339: index (L-1) into row r   */
        int idx = L - 1;                           /* row 0 starts at level 2 => off by one   */
        int row_len = kAcFirst[r + 1] - kAcFirst[r];
        if (idx < row_len) {
            int e = kAcFirst[r] + idx;
            orc_bits_put(out, kAcCode[e], kAcBits[e]);
            return ORC_OK;                         /* no sign bit: vlc.c:344 is commented out */
        }
    }
    if (L >= 256 || r >= 64) return ORC_E_UNENCODABLE; /* vlc.c:349 -> NULL -> segfault      */
    orc_bits_put(out, 0x01, 6);                    /* escape "000001", vlc.c:351              */
    orc_bits_put(out, (uint32_t)r & 0x3f, 6);      /* vlc.c:354                               */
    if (L < 128) {                                 /* vlc.c:357-363                           */
        uint8_t e = (uint8_t)(L & 0x7f);
        if (negative) e = (uint8_t)(~e + 1);
        orc_bits_put(out, e, 8);
    } else {                                       /* vlc.c:364-375                           */
        uint8_t e = (uint8_t)L;
        if (negative) e = (uint8_t)(~e + 1);
        orc_bits_put(out, negative ? 0x80 : 0x00, 8);
        orc_bits_put(out, e, 8);
    }
    return ORC_OK;
}

/* image_processing.c:400-433: walk (level, run) pairs, stop at the first with run==0 or level==0. */
static int put_ac_pairs(orc_bits *out, const int32_t *pairs) {
    for (int k = 0; k < 64; k++) {
        int level = pairs[2 * k], run = pairs[2 * k + 1];
        if (run == 0 || level == 0) break;
        int rc = put_ac(out, run, level);
        if (rc != ORC_OK) return rc;
    }
    return ORC_OK;
}

int orc_encode_block(int is_luma, const int32_t zz[64], orc_bits *out) {
    int32_t pairs[130];
    orc_run_length(zz, pairs);
    int rc;
    if (pairs[0] != 0 && pairs[1] == 0) {          /* mpeg1_blk.c:73: DC != 0                 */
        int coe = pairs[0] < 0 ? -pairs[0] : pairs[0];
        int sz = 1;                                /* mpeg1_blk.c:77-83: highest set bit among */
        for (int i = 1; i <= 8; i++)               /* bits 0..7, 1-based, default 1           */
            if (coe & (1 << (i - 1))) sz = i;
        if (is_luma) orc_bits_put(out, kDcLumaCode[sz], kDcLumaBits[sz]);       /* vlc.c:146 */
        else orc_bits_put(out, kDcChromaCode[sz], kDcChromaBits[sz]);
        if (pairs[0] < 0) coe ^= 1 << (sz - 1);    /* mpeg1_blk.c:87-89                       */
        orc_bits_put(out, (uint32_t)(coe & 0xff), sz); /* mpeg1_blk.c:91: low sz bits         */
        rc = put_ac_pairs(out, pairs + 2);         /* mpeg1_blk.c:93                          */
    } else {
        if (is_luma) orc_bits_put(out, 0x4, 3);    /* "100", mpeg1_blk.c:99                   */
        else orc_bits_put(out, 0x0, 2);            /* "00",  mpeg1_blk.c:101                  */
        rc = put_ac_pairs(out, pairs);             /* mpeg1_blk.c:104                         */
    }
    if (rc != ORC_OK) return rc;
    orc_bits_put(out, 0x2, 2);                     /* EOB "10", mpeg1_blk.c:115-117           */
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Frame stage                                                                                */
/* ------------------------------------------------------------------------------------------ */

static int check_geometry(int W, int H, int channels, int mode, int *xe, int *ye) {
    if (W <= 0 || H <= 0 || channels < 3) return ORC_E_ARG;
    orc_region(mode, W, H, xe, ye);
    if (*xe > W || *ye > H) return ORC_E_ARG; /* the reference would read outside the planes */
    return ORC_OK;
}

/* extract_8x8_block, image_processing.c:138-150. */
static inline void cut_block(const uint8_t *plane, int stride, int x0, int y0, uint8_t blk[64]) {
    for (int i = 0; i < 8; i++)
        memcpy(blk + i * 8, plane + (size_t)(y0 + i) * stride + x0, 8);
}

typedef int (*block_sink)(void *ctx, int is_luma, const int32_t zz[64]);

/* The slice -> macroblock -> block walk of encoder.h:238-440 over already converted planes.
 * Calls on_strip(s) before each strip, on_mb() before each macroblock, sink() per block and
 * on_strip_end() after each strip. */
typedef struct {
    void *ctx;
    void (*on_strip)(void *ctx, int strip);
    void (*on_mb)(void *ctx);
    block_sink sink;
    void (*on_strip_end)(void *ctx);
} walk_ops;

static int walk_picture(const uint8_t *Y, const uint8_t *Cb, const uint8_t *Cr, int W,
                        int xe, int ye, const int32_t q[64], const walk_ops *ops) {
    uint8_t blk[64];
    int32_t dct[64], zz[64];
    for (int x = 0; x < xe; x += 16) {
        if (ops->on_strip) ops->on_strip(ops->ctx, x / 16);
        for (int y = 0; y < ye; y += 16) {
            if (ops->on_mb) ops->on_mb(ops->ctx);
            for (int b = 0; b < 4; b++) { /* encoder.h:263-341 */
                cut_block(Y, W, x + (b % 2) * 8, y + (b / 2) * 8, blk);
                orc_fdct(blk, dct);
                orc_quant_zigzag(dct, q, zz);
                int rc = ops->sink(ops->ctx, 1, zz);
                if (rc != ORC_OK) return rc;
            }
            /* encoder.h:347-348: FULL-resolution Cb/Cr planes addressed with stride W/2 */
            const uint8_t *planes[2] = {Cb, Cr};
            for (int c = 0; c < 2; c++) {
                cut_block(planes[c], W / 2, x / 2, y / 2, blk);
                orc_fdct(blk, dct);
                orc_quant_zigzag(dct, q, zz);
                int rc = ops->sink(ops->ctx, 0, zz);
                if (rc != ORC_OK) return rc;
            }
        }
        if (ops->on_strip_end) ops->on_strip_end(ops->ctx);
    }
    return ORC_OK;
}

/* --- coefficient dump (BASELINE config 2) --- */
static int coeff_sink(void *ctx, int is_luma, const int32_t zz[64]) {
    (void)is_luma;
    int32_t **cur = (int32_t **)ctx;
    memcpy(*cur, zz, 64 * sizeof(int32_t));
    *cur += 64;
    return ORC_OK;
}

int orc_frame_coefficients(const uint8_t *rgb, int W, int H, int channels, int quality_factor,
                           int mode, int32_t *coeffs) {
    int xe, ye, rc = check_geometry(W, H, channels, mode, &xe, &ye);
    if (rc != ORC_OK) return rc;
    size_t npx = (size_t)W * H;
    uint8_t *planes = (uint8_t *)malloc(3 * npx);
    if (!planes) return ORC_E_ARG;
    orc_convert_rgb_to_ycbcr(rgb, channels, npx, planes, planes + npx, planes + 2 * npx);
    int32_t q[64];
    orc_scale_qmatrix(quality_factor, q);
    int32_t *cur = coeffs;
    walk_ops ops = {&cur, NULL, NULL, coeff_sink, NULL};
    rc = walk_picture(planes, planes + npx, planes + 2 * npx, W, xe, ye, q, &ops);
    free(planes);
    return rc;
}

/* --- bitstream --- */
static void bs_on_strip(void *ctx, int strip) { /* mpeg1_slice, mpeg1_blk.c:12-16 */
    orc_bits *b = (orc_bits *)ctx;
    orc_bits_put(b, 0x000001, 24);
    orc_bits_put(b, (uint32_t)(strip + 1) & 0xff, 8); /* uint8_t vertical_pos wraps          */
    orc_bits_put(b, 1, 5);                            /* quant_scale = 1 (encoder.h:51)      */
    orc_bits_put(b, 0, 1);
}
static void bs_on_mb(void *ctx) { /* encode_macroblock_header_i(1,...), mpeg1_blk.c:38-51 */
    orc_bits_put((orc_bits *)ctx, 0x3, 2); /* increment 1 -> "1", intra type -> "1" */
}
static int bs_sink(void *ctx, int is_luma, const int32_t zz[64]) {
    return orc_encode_block(is_luma, zz, (orc_bits *)ctx);
}
static void bs_on_strip_end(void *ctx) { bits_align_zero((orc_bits *)ctx); }

/* mpeg1_enc.c:59-64 / :67-71 — one 5-byte timestamp field. */
static void put_ts(uint8_t *o, uint8_t prefix, uint32_t v) {
    o[0] = (uint8_t)(prefix | ((v & 0xe0000000u) >> 28));
    o[1] = (uint8_t)((v & 0x1fe00000u) >> 21);
    o[2] = (uint8_t)(0x01 | ((v & 0x001fc000u) >> 13));
    o[3] = (uint8_t)((v & 0x00003fc0u) >> 6);
    o[4] = (uint8_t)(0x01 | ((v & 0x0000003fu) << 1));
}

/* The 44 fixed bytes in front of a frame's strips: packet (mpeg1_enc.c:47-77), sequence
 * (:81-94), GOP (:103-113) and picture (:120-129) headers with the driver's constants
 * (encoder.h:37-63,186-187,200-230,475-484).  The packet length field is left zero. */
static void frame_headers(uint8_t h[44], int W, int H, int frame_index) {
    uint8_t hour = (uint8_t)frame_index;   /* encoder.h:42,475-484: hour++ every frame, u8 */
    uint8_t w8 = (uint8_t)W, h8 = (uint8_t)H; /* encoder.h:186-187 */
    memset(h, 0, 44);
    /* packet */
    h[2] = 0x01; h[3] = 0xe0;
    uint32_t dts = (uint32_t)(1 + 3600 * (int)hour);
    dts = (uint32_t)((double)dts * 1.2);   /* mpeg1_enc.c:57 */
    dts += 0xbeef;
    put_ts(h + 6, 0x31, dts);
    dts -= 0xbeef;
    put_ts(h + 11, 0x11, dts);
    /* sequence */
    uint8_t *s = h + 16;
    uint16_t w = w8, hh = h8;
    s[2] = 0x01; s[3] = 0xb3;
    s[4] = (uint8_t)((w & 0xff0) >> 4);
    s[5] = (uint8_t)(((w & 0xf) << 4) | ((hh & 0xf00) >> 8));
    s[6] = (uint8_t)(hh & 0xff);
    s[7] = (uint8_t)(((1 & 0xf) << 4) | (4 & 0xf)); /* aspect 1, frame rate 4 */
    s[8] = 0xff; s[9] = 0xff; s[10] = 0xe0;
    s[11] = (uint8_t)((3 & 0x1f) << 3);              /* yby_size 3 */
    /* GOP: minute = second = 0 (reset every frame), num_pic 0, closed 1, broken 0 */
    uint8_t *g = h + 28;
    g[2] = 0x01; g[3] = 0xb8;
    g[4] = (uint8_t)((hour & 0x1f) << 2);
    g[5] = 0x08;
    g[6] = 0x00;
    g[7] = 0x40;
    /* picture: temporal_ref 0, type 1 (I), vbv_delay 0xffff */
    uint8_t *p = h + 36;
    p[2] = 0x01; p[3] = 0x00;
    p[4] = 0x00;
    p[5] = (uint8_t)((1 << 3) | ((0xffff & 0xe000) >> 13));
    p[6] = (uint8_t)((0xffff & 0x1fe0) >> 5);
    p[7] = (uint8_t)((0xffff & 0x1f) << 3);
}

size_t orc_frame_bound(int W, int H, int mode) {
    int xe, ye;
    orc_region(mode, W, H, &xe, &ye);
    size_t strips = (size_t)xe / 16, mbs = (size_t)ye / 16;
    size_t strip_bits = 38 + mbs * (2 + 6 * 886);
    return 44 + strips * ((strip_bits + 7) / 8) + 4;
}

long orc_encode_frame(const uint8_t *rgb, int W, int H, int channels, int frame_index,
                      int quality_factor, int mode, uint8_t *out, size_t cap) {
    int xe, ye, rc = check_geometry(W, H, channels, mode, &xe, &ye);
    if (rc != ORC_OK) return rc;
    size_t npx = (size_t)W * H;
    uint8_t *planes = (uint8_t *)malloc(3 * npx);
    if (!planes) return ORC_E_ARG;
    orc_convert_rgb_to_ycbcr(rgb, channels, npx, planes, planes + npx, planes + 2 * npx);
    int32_t q[64];
    orc_scale_qmatrix(quality_factor, q);

    orc_bits bits;
    orc_bits_init(&bits);
    walk_ops ops = {&bits, bs_on_strip, bs_on_mb, bs_sink, bs_on_strip_end};
    rc = walk_picture(planes, planes + npx, planes + 2 * npx, W, xe, ye, q, &ops);
    free(planes);
    if (rc != ORC_OK) { orc_bits_free(&bits); return rc; }

    size_t payload = bits.nbits >> 3;              /* bitvector_fwrite, bit_vector.c:136-139 */
    size_t total = 44 + payload + 4;
    if (total > cap) { orc_bits_free(&bits); return ORC_E_NOSPACE; }
    frame_headers(out, W, H, frame_index);
    memcpy(out + 44, bits.buf, payload);
    /* encoder.h:448-453: (u16)(ftell - (frame_start+4)) - 4, big-endian, at frame_start+4 */
    uint16_t fwd = (uint16_t)((44 + payload) - 4);
    fwd = (uint16_t)(fwd - 4);
    out[4] = (uint8_t)(fwd >> 8);
    out[5] = (uint8_t)(fwd & 0xff);
    /* encoder.h:456-458 writes 4 uninitialised stack bytes; observed 00 00 00 00 */
    memset(out + 44 + payload, 0, 4);
    orc_bits_free(&bits);
    return (long)total;
}

size_t orc_file_prolog(uint8_t out[27]) {
    /* mpeg1_file_header(2202035), mpeg1_enc.c:7-21 */
    static const uint8_t pack[9] = {0x00, 0x00, 0x01, 0xba, 0x21, 0x00, 0x01, 0x00, 0x01};
    memcpy(out, pack, 9);
    uint32_t rate = (2202035u & 0x3fffffu) | 0x400000u;
    rate = (rate << 1) | 1u;
    out[9] = (uint8_t)(rate >> 16);
    out[10] = (uint8_t)(rate >> 8);
    out[11] = (uint8_t)rate;
    /* mpeg1_sys_header(2202035, 0xe6), mpeg1_enc.c:24-44 */
    uint8_t *s = out + 12;
    s[0] = 0x00; s[1] = 0x00; s[2] = 0x01; s[3] = 0xbb;
    s[4] = 0x00; s[5] = 0x09;
    s[6] = (uint8_t)(rate >> 16); s[7] = (uint8_t)(rate >> 8); s[8] = (uint8_t)rate;
    s[9] = 0x00; s[10] = 0x21; s[11] = 0xff;
    s[12] = 0xe0; s[13] = 0xe0; s[14] = 0xe6;
    return 27;
}

/* --- frame-parallel batch --- */
typedef struct {
    const uint8_t *rgb;
    int n_frames, W, H, channels, first_index, qf, mode, n_threads, tid;
    uint8_t **bufs;
    long *sizes;
    size_t bound;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *j = (batch_job *)arg;
    size_t fbytes = (size_t)j->W * j->H * j->channels;
    for (int f = j->tid; f < j->n_frames; f += j->n_threads) {
        j->bufs[f] = (uint8_t *)malloc(j->bound);
        j->sizes[f] = orc_encode_frame(j->rgb + fbytes * f, j->W, j->H, j->channels,
                                       j->first_index + f, j->qf, j->mode, j->bufs[f], j->bound);
        if (j->sizes[f] > 0) { /* shrink so that many frames do not hold worst-case buffers */
            uint8_t *s = (uint8_t *)realloc(j->bufs[f], (size_t)j->sizes[f]);
            if (s) j->bufs[f] = s;
        }
    }
    return NULL;
}

long orc_encode_frames(const uint8_t *rgb, int n_frames, int W, int H, int channels,
                       int first_frame_index, int quality_factor, int mode, int n_threads,
                       uint8_t *out, size_t cap, uint64_t *frame_sizes) {
    int xe, ye, rc = check_geometry(W, H, channels, mode, &xe, &ye);
    if (rc != ORC_OK) return rc;
    if (n_frames <= 0) return 0;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_frames) n_threads = n_frames;
    uint8_t **bufs = (uint8_t **)calloc((size_t)n_frames, sizeof *bufs);
    long *sizes = (long *)calloc((size_t)n_frames, sizeof *sizes);
    batch_job *jobs = (batch_job *)calloc((size_t)n_threads, sizeof *jobs);
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof *th);
    for (int t = 0; t < n_threads; t++) {
        jobs[t] = (batch_job){rgb, n_frames, W, H, channels, first_frame_index, quality_factor,
                              mode, n_threads, t, bufs, sizes, orc_frame_bound(W, H, mode)};
        if (n_threads == 1) batch_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    if (n_threads > 1)
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    long total = 0;
    for (int f = 0; f < n_frames && total >= 0; f++) {
        if (sizes[f] < 0) total = sizes[f];
        else if ((size_t)total + (size_t)sizes[f] > cap) total = ORC_E_NOSPACE;
        else {
            memcpy(out + total, bufs[f], (size_t)sizes[f]);
            if (frame_sizes) frame_sizes[f] = (uint64_t)sizes[f];
            total += sizes[f];
        }
    }
    for (int f = 0; f < n_frames; f++) free(bufs[f]);
    free(bufs); free(sizes); free(jobs); free(th);
    return total;
}

int orc_write_bit_file(const char *path, const uint8_t *Y, const uint8_t *Cb, const uint8_t *Cr,
                       int W, int H) {
    FILE *f = fopen(path, "wb");
    if (!f) return ORC_E_ARG;
    int32_t w = W, h = H;
    size_t npx = (size_t)W * H;
    fwrite(&w, sizeof w, 1, f);
    fwrite(&h, sizeof h, 1, f);
    fwrite(Y, 1, npx, f);
    fwrite(Cb, 1, npx, f);
    fwrite(Cr, 1, npx, f);
    fclose(f);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Synthetic input                                                                            */
/* ------------------------------------------------------------------------------------------ */

static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_synth_frame(uint8_t *rgb, size_t nbytes, uint64_t seed, uint64_t frame_index) {
    uint64_t base = seed + frame_index * 0x9E3779B97F4A7C15ull;
    size_t words = nbytes >> 3;
    for (size_t j = 0; j < words; j++) {
        uint64_t v = splitmix64(base + j);
        memcpy(rgb + 8 * j, &v, 8); /* little-endian host */
    }
    if (nbytes & 7) {
        uint64_t v = splitmix64(base + words);
        memcpy(rgb + 8 * words, &v, nbytes & 7);
    }
}
