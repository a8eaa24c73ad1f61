/* oracle/mpeg1_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C11, gcc) of the reference's MPEG-1 I-frame hot path
 * (eburhansjah/ec504_ImageEncoder; citations are file:line into /root/reference).  It is the
 * checker the HIP path is compared against and the "port" CPU baseline of bench.py.  It is NEVER
 * linked into, imported by or executed from the product (ec504_imageencoder_amd/, libencoder.so).
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks this restatement against the real
 * reference compiled from its own sources (oracle/_ref, authoring container only) and
 * tests/test_oracle_golden.py checks it against the committed vectors in tests/golden/ that were
 * produced by that reference build (tests/golden/make_goldens.py).
 */
#ifndef MPEG1_ORACLE_H
#define MPEG1_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Region the macroblock loops cover (SURVEY §7 "STRICT vs FULL").
 *   STRICT: x in [0,96), y in [0,144)  — the literals at include/encoder.h:238,248
 *   FULL  : x in [0, W & ~15), y in [0, H & ~15) — the reference with those two literals restored */
enum { ORC_MODE_STRICT = 0, ORC_MODE_FULL = 1 };

enum {
    ORC_OK = 0,
    ORC_E_ARG = -1,          /* bad dimensions / channels / region outside the picture            */
    ORC_E_UNENCODABLE = -2,  /* an emitted AC level has |level| >= 256: vlc.c:349 returns NULL and
                                bitvector_concat(dest, NULL) dereferences it (reference segfaults)   */
    ORC_E_NOSPACE = -3       /* caller's output buffer too small                                   */
};

void orc_region(int mode, int W, int H, int *x_extent, int *y_extent);

/* image_processing.c:68-110.  `data` interleaved, `channels` >= 3 bytes per pixel. */
void orc_convert_rgb_to_ycbcr(const uint8_t *data, int channels, size_t npx,
                              uint8_t *Y, uint8_t *Cb, uint8_t *Cr);
/* image_processing.c:114-133 (even W and H only; the reference reads out of bounds otherwise). */
void orc_subsample_420(const uint8_t *Cb, const uint8_t *Cr, int W, int H,
                       uint8_t *Cb_sub, uint8_t *Cr_sub);
/* image_processing.c:192-307.  in[i*8+j] row-major pixels, out[u*8+i] = dct_block[u][i]. */
void orc_fdct(const uint8_t in[64], int32_t out[64]);
/* image_processing.c:314-343. */
void orc_scale_qmatrix(int quality_factor, int32_t q[64]);
/* image_processing.c:349-381: quantise (C truncating division) then scatter by ZIGZAG_ORDER. */
void orc_quant_zigzag(const int32_t dct[64], const int32_t q[64], int32_t zz[64]);
/* image_processing.c:703-751: pairs[2k]=level, pairs[2k+1]=zeros before it; (0,0) terminator.
 * `pairs` must hold 130 ints.  Returns the number of pairs. */
int orc_run_length(const int32_t zz[64], int32_t pairs[130]);

/* MSB-first append-only bit buffer (bit_vector.c put/concat semantics). */
typedef struct {
    uint8_t *buf;
    size_t   cap_bytes;
    size_t   nbits;
} orc_bits;
void orc_bits_init(orc_bits *b);
void orc_bits_free(orc_bits *b);
void orc_bits_put(orc_bits *b, uint32_t value, int n); /* n <= 32, value's low n bits, MSB first */

/* mpeg1_blk.c:67-117 + image_processing.c:400-433 + vlc.c:146-157,315-385:
 * DC size/value (or "100"/"00"), the AC codes up to the first pair with run 0, then EOB "10". */
int orc_encode_block(int is_luma, const int32_t zz[64], orc_bits *out);

/* The 64 zigzag-ordered quantised levels of every block the driver visits, in emission order:
 * strip (x) major, macroblock (y) next, then Y0 Y1 Y2 Y3 Cb Cr  (encoder.h:238-423).
 * coeffs holds n_strips*n_mbrows*6*64 int32.  (BASELINE config 2.) */
int orc_frame_coefficients(const uint8_t *rgb, int W, int H, int channels, int quality_factor,
                           int mode, int32_t *coeffs);

/* One frame record:  PKT(16) SEQ(12) GOP(8) PIC(8) strips 00 00 00 00  (encoder.h:196-458).
 * frame_index is the GLOBAL index i of the frame loop (drives `hour`).  Returns bytes written
 * into out (<= cap) or a negative ORC_E_*. */
long orc_encode_frame(const uint8_t *rgb, int W, int H, int channels, int frame_index,
                      int quality_factor, int mode, uint8_t *out, size_t cap);
/* Upper bound of orc_encode_frame's output for a picture of this size. */
size_t orc_frame_bound(int W, int H, int mode);

/* PACK(12)+SYS(15) written once per file (encoder.h:86-89, mpeg1_enc.c:7-44).  Returns 27. */
size_t orc_file_prolog(uint8_t out[27]);

/* n_frames frames, contiguous in `rgb` (W*H*channels each) -> contiguous frame records in out;
 * frame_sizes[f] receives each record's byte count.  n_threads >= 1 frame-parallel workers
 * (1 = the reference's execution model).  Returns total bytes or negative ORC_E_*. */
long orc_encode_frames(const uint8_t *rgb, int n_frames, int W, int H, int channels,
                       int first_frame_index, int quality_factor, int mode, int n_threads,
                       uint8_t *out, size_t cap, uint64_t *frame_sizes);

/* image_processing.c:753-787: int32 W, int32 H, Y, full-res Cb, full-res Cr. */
int orc_write_bit_file(const char *path, const uint8_t *Y, const uint8_t *Cb, const uint8_t *Cr,
                       int W, int H);

/* Synthetic frame generator shared (by definition, not by code) with the HIP fill kernel:
 * byte k of frame f = byte (k & 7) (little-endian) of splitmix64(seed + f*0x9E3779B97F4A7C15 + (k >> 3)). */
void orc_synth_frame(uint8_t *rgb, size_t nbytes, uint64_t seed, uint64_t frame_index);

#ifdef __cplusplus
}
#endif
#endif
