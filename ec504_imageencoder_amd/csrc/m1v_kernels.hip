// ec504_imageencoder_amd/csrc/m1v_kernels.hip — MI355X (gfx950 / CDNA4) kernels and the C-ABI of
// include/mpeg1_hip.h.  Written for gfx950 only: 64-wide waves, LDS-staged bit packing, 5 waves per SIMD.
//
// Data layout in HBM
//   input    n_frames x (H x W x C) interleaved u8, as the reference's Image::data (jpeg_handler.h:6-11)
//   scratch  one compact slot per unit of the encode kernel (a tile of 8 strips x 4 macroblock rows, or a run of 256 blocks)
//            + an overflow arena of worst-case slots; a strip = 16-pixel-wide COLUMN of macroblocks (encoder.h:238
//            iterates x outermost) and is byte aligned (encoder.h:442), so strips are independent units of bit packing;
//            a segment table (bits, where) per (frame, segment, strip); per-strip bit counters, per-frame byte totals
//   output   contiguous frame records  PKT SEQ GOP PIC strips 00000000  (encoder.h:196-458)
//
// Kernels
//   k_encode_tiles    (m1v_tiles.h) the encode kernel of every 3-channel picture: a workgroup per tile of 8 strips x 4
//                     macroblock rows, pixels in as whole 128-byte lines by LDS-DMA, one lane per 8x8 block
//   k_encode_dense    the run kernel (4-channel pictures; round 2's hot kernel).  A frame's blocks, in
//                     stream order, are cut into runs of T consecutive blocks (default 256); one
//                     workgroup per (frame, run), one LANE per 8x8 block (Y0..Y3, Cb, Cr of each
//                     macroblock down the strip).  Per lane: 8 rows x 24 B of RGB -> component (three
//                     fp32 FMAs; an unfused fp64 fix-up where the fp32 result is within 2 eps of an
//                     integer, so results equal the reference's fp64 arithmetic bit for bit) ->
//                     two-pass integer FDCT in registers -> quantise -> zigzag positions staged in LDS
//                     as int8 (min AC divisor >= 8) or int16 -> DC/AC code words.  Bit lengths are
//                     prefix-summed across the workgroup (DPP wave scan, one barrier), the bits are
//                     OR-ed into an LDS image of the run (<= 2 byte-aligned strip segments) and stored
//                     once.  Workgroups are dealt to XCDs so one frame's runs share an L2.
//   k_encode_strips   one 64-lane workgroup per (frame, strip) for small pictures (< 64 blocks/strip)
//   k_dense_frame_layout   run kernel only: run metadata -> segment table, strip bit counts, frame bytes
//   k_assemble        (m1v_assemble.h) ONE launch behind every encode kernel: frame and strip offsets, the strips' segments
//                     at their final bit positions, frame headers, 16-bit length back-patch, trailer, sizes, status
//   k_coefficients    FDCT+quant+zigzag only (BASELINE config 2)
//   k_convert, k_subsample, k_synth   plane conversion / 4:2:0 / synthetic input
//
// Reference citations are file:line under /root/reference.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <algorithm>
#include <vector>

#include "../../include/mpeg1_hip.h"

#define M1V_HD __host__ __device__ __forceinline__
#include "fdct_f32.h"

#pragma clang fp contract(off) // colour conversion must stay unfused (image_processing.c:104-106)

namespace {

// ------------------------------------------------------------------------------------------------
// constant geometry
// ------------------------------------------------------------------------------------------------
constexpr int kWave = 64;
// VLC table, one private copy per wave in LDS (192 words: a wave fills and reads only its own copy, so the waves of a
// workgroup need not meet between the pixel stage and the entropy stage):
//   [0, 32)    per run row r = run - 1:  first entry | entries << 8   (vlc.c:172-174, the offset index)
//   [32, 142)  the 110 run/level entries in the reference's order, (bits << 16) | code  (vlc.c:176-288)
//   [144, 153) luma DC size codes, [160, 169) chroma DC size codes              (vlc.c:121-144)
constexpr int kAcRows = 32;
constexpr int kVlcRowInfo = 0, kVlcEntries = 32, kVlcDcLuma = 144, kVlcDcChroma = 160, kVlcWords = 192;
constexpr int kMaxBlockBits = 886;
constexpr int kDefaultLdsWords = 4096;            // 16 KiB strip image in LDS (strip-per-workgroup kernel)

// zigzag position of natural-order coefficient [u][i] (image_processing.c:28-37)
__host__ __device__ constexpr int scan_pos(int k) {
    constexpr int t[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42,
                           3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
                           10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
                           21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
    return t[k];
}
// natural-order index of zigzag position p
__host__ __device__ constexpr int scan_inv(int p) {
    for (int k = 0; k < 64; k++)
        if (scan_pos(k) == p) return k;
    return -1;
}

// LDS staging of one block's 64 quantised levels, "zigzag class" layout.  A block owns kStageStride consecutive
// words (an odd stride: the lanes of a wave hit different banks).  Narrow form (one BYTE per level, exact whenever no
// AC level can reach +-128, which the host decides from the quantiser alone): word (p & 7) + 8 (p >> 5) holds the
// levels at zigzag positions p, p+8, p+16, p+24 in its four bytes.  Wide form (int16): word (p & 15) + 16 (p >> 5)
// holds positions p and p+16.  The point of the layout: the "is this byte non-zero" flags of one word, computed for
// all four bytes at once, land on zigzag positions that differ by 8, so ONE shift puts them in place in the 64-bit
// non-zero mask (stage_nonzero_mask) — 1.25 instructions per coefficient instead of a compare, a select and an OR.
constexpr int kStageStride8 = 17, kStageStride16 = 33;
__host__ __device__ constexpr int stage_byte8(int p) { return ((p & 7) + 8 * (p >> 5)) * 4 + ((p >> 3) & 3); }
__host__ __device__ constexpr int stage_byte16(int p) { return ((p & 15) + 16 * (p >> 5)) * 4 + 2 * ((p >> 4) & 1); }

// Device-resident tables, built by m1v_create.
struct Tables {
    float rq[64];               // inflated reciprocal of the scaled quantiser, natural order [u][i]
    float rq_t[64];             // the same, transposed [i][u]: one 32-byte scalar load per column pass
    uint32_t vlc[kVlcWords];    // see kVlc*
    uint8_t hdr[256][44];       // PKT SEQ GOP PIC for hour = 0..255, length field zero
};

struct Geometry {
    int W, H, C;
    int n_strips, n_mbrows;     // x_extent/16, y_extent/16
    int half_w;                 // W / 2 (stride of the chroma quirk, encoder.h:347)
    uint32_t strip_cap;         // bytes of one scratch slot (multiple of 16)
    unsigned long long frame_bytes;
};

#ifdef M1V_STAMPS
// Diagnostic build only (tools/stamps.py): per-phase cycle sums, lane 0 of every wave adds the cycles it
// spent between two stamps into stamps[phase].  Never compiled into the shipped library.
#define STAMP(ph)                                                                                  \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long now_ = __builtin_amdgcn_s_memtime();                                    \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                        \
        if ((threadIdx.x & 63) == 0) atomicAdd(&a.stamps[ph], now_ - stamp_t_);                    \
        stamp_t_ = __builtin_amdgcn_s_memtime();                                                   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define STAMP_INIT() unsigned long long stamp_t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#elif defined(M1V_MARKS)
// Analysis build only (tools/isa_phases.py): comment markers in the .s between phases.
#define STAMP(ph) asm volatile("; PHASE_MARK " #ph ::: "memory")
#define STAMP_INIT() asm volatile("; PHASE_MARK start" ::: "memory")
#else
#define STAMP(ph) do { } while (0)
#define STAMP_INIT() do { } while (0)
#endif

struct EncodeArgs {
    Geometry g;
    const uint8_t *rgb;
    const Tables *tab;
    uint8_t *scratch;           // [frame][strip][strip_cap]
    uint2 *seg;                 // [frame][strip]: (bits of the strip, where it starts in scratch: 4-byte words) — one segment per strip
    unsigned long long *strip_ctr;   // [frame][strip]: bits of the strip
    unsigned long long *frame_bytes; // [frame]: every strip adds its bytes (zero before the batch: k_assemble of the batch before)
    uint32_t *status;
    int n_frames;
    int threads;                // workgroup size
    int lds_words;              // capacity of the LDS strip image
    unsigned long long *stamps; // diagnostic builds only
};

// ------------------------------------------------------------------------------------------------
// pixel stage
// ------------------------------------------------------------------------------------------------

// One colour component, exactly as image_processing.c:104-106 evaluates it: fp64, left to right,
// one rounding per operation, truncation to u8.  (k0,kr,kg,kb) select Y / Cb / Cr:
//   Y  = 0.299 r + 0.587 g + 0.114 b              -> (0,   .299,     .587,     .114)
//   Cb = 128 - 0.168736 r - 0.331264 g + 0.5 b     -> (128, -.168736, -.331264, .5)
//   Cr = 128 + 0.5 r - 0.418688 g - 0.081312 b     -> (128, .5,       -.418688, -.081312)
// a - c*x == a + (-c)*x and 0 + c*x == c*x hold exactly in IEEE arithmetic.
__device__ __forceinline__ int component_fp64(int r, int g, int b, double k0, double kr, double kg,
                                              double kb) {
    double acc = k0 + kr * (double)r;
    acc = acc + kg * (double)g;
    acc = acc + kb * (double)b;
    return (int)acc;
}

struct CompCoef {
    double k0, kr, kg, kb;
};
__device__ __forceinline__ CompCoef comp_coef(int comp) { // 0 = Y, 1 = Cb, 2 = Cr
    CompCoef c;
    c.k0 = comp == 0 ? 0.0 : 128.0;
    c.kr = comp == 0 ? 0.299 : (comp == 1 ? -0.168736 : 0.5);
    c.kg = comp == 0 ? 0.587 : (comp == 1 ? -0.331264 : -0.418688);
    c.kb = comp == 0 ? 0.114 : (comp == 1 ? 0.5 : -0.081312);
    return c;
}

// fp32 coefficients of the same three formulas, for the fast path below.  k0 carries +256 + kEps.
constexpr float kEps = 1.5e-4f;
struct CompCoefF {
    float k0, kr, kg, kb;
    CompCoef d; // the reference's fp64 coefficients of the same component, for the pixels the fp32 form cannot decide
};
__device__ __forceinline__ CompCoefF comp_coef_f(int comp) {
    CompCoefF c;
    c.k0 = (comp == 0 ? 0.0f : 128.0f) + 256.0f + kEps;
    c.kr = comp == 0 ? 0.299f : (comp == 1 ? -0.168736f : 0.5f);
    c.kg = comp == 0 ? 0.587f : (comp == 1 ? -0.331264f : -0.418688f);
    c.kb = comp == 0 ? 0.114f : (comp == 1 ? 0.5f : -0.081312f);
    c.d = comp_coef(comp);
    return c;
}
// The same for a wave whose lanes all convert luma (first = true) or convert Cb in lanes 0-31 and Cr in lanes 32-63 (the
// waves of a tile, m1v_tiles.h): a scalar branch and, on the chroma side, one select per register — a third of the vector
// instructions the general three-way selects of comp_coef_f take per wave (12 registers; the fp64 ones are loop invariants
// the compiler sets up in front of the rows either way).
__device__ __forceinline__ CompCoefF comp_coef_wave(bool luma_wave, int lane) {
    CompCoefF c;
    if (luma_wave) {
        asm volatile(""); // keeps the branch: two arms of selects would be merged back into three-way selects
        c = comp_coef_f(0);
    } else {
        asm volatile("");
        const CompCoefF cb = comp_coef_f(1), cr = comp_coef_f(2);
        const bool hi = lane >= 32;
        c.k0 = cb.k0;
        c.kr = hi ? cr.kr : cb.kr;
        c.kg = hi ? cr.kg : cb.kg;
        c.kb = hi ? cr.kb : cb.kb;
        c.d.k0 = cb.d.k0;
        c.d.kr = hi ? cr.d.kr : cb.d.kr;
        c.d.kg = hi ? cr.d.kg : cb.d.kg;
        c.d.kb = hi ? cr.d.kb : cb.d.kb;
    }
    return c;
}

// Same value as component_fp64 for every (r,g,b), at fp32 cost, and already the float the fp32 FDCT (fdct_f32.h) takes.
// The exact rational value x of a formula is a multiple of 1e-6 in [0, 255.5].  Three fp32 FMAs starting from
// k0 + 256 + eps give t = x + 256 + eps + e with |e| < 1e-4 < eps (three half-ulps of 2^-15, three coefficient
// roundings, the rounding of the constant); the reference's fp64 result differs from x by < 1e-12.  t lies in
// [256, 512) for every input, so the float's exponent is fixed and its low 15 mantissa bits are the fraction:
//     raw pixel      p = t with those 15 bits cleared = 256 + trunc(t - 256)     (one v_and_b32; kPxBiasF + value)
//     fraction       d = t - p                                                   (exact)
// If d >= kFracLow (> 2 eps) then x >= trunc(t - 256) + d - eps - 1e-4 lies at least 0.5e-4 above that integer and,
// as d < 1, at least eps - 1e-4 below the next one: trunc(fp64 result) == trunc(t - 256).  Otherwise (x an exact
// integer — where fp64 rounding decides the byte — or less than 1.5e-4 above one: 0.16 % of the pixels) the lane
// re-evaluates the reference's fp64 expression.  The row's smallest d is reduced with v_min3_f32 (two pixels per
// instruction) and compared once per row.  Raw pixels keep their bias through the FDCT: every multiplier input of the
// butterflies is a difference (bias cancels), only the DC sum carries 64 * kPxBiasF, removed in fdct_col_f.
// Proof over all 2^24 triples x 3 components: tools/colour_fast_proof.c (host) and the GPU tests.
constexpr float kFracLow = 10.0f / 32768.0f;
__device__ __forceinline__ float component_t(uint32_t r, uint32_t g, uint32_t b, const CompCoefF &k) {
    float t = fmaf((float)b, k.kb, k.k0);
    t = fmaf((float)g, k.kg, t);
    return fmaf((float)r, k.kr, t);
}
__device__ __forceinline__ float clear_fraction(float t) { return __uint_as_float(__float_as_uint(t) & 0xffff8000u); }
// raw pixel (kPxBiasF + value) of one component, per-pixel branch (plane conversion and the byte-load input mode)
__device__ __forceinline__ float component_raw(uint32_t r, uint32_t g, uint32_t b, const CompCoefF &k) {
    const float t = component_t(r, g, b, k);
    float p = clear_fraction(t);
    if (t - p < kFracLow) {
        const CompCoef &d = k.d;
        p = m1vf::kPxBiasF + (float)component_fp64((int)r, (int)g, (int)b, d.k0, d.kr, d.kg, d.kb);
    }
    return p;
}

#define M1V_CONST_AS __attribute__((address_space(4)))


// The tile kernels run their fp32 arithmetic rounded TOWARD MINUS INFINITY: fdct_row_f<float, true> takes two floors of
// products that round (fdct_f32.h) and needs that mode; the colour sums are proven for it as well as for the default
// (tools/colour_fast_proof.c down: the same pixels are flagged, none is wrong), the reciprocal quantiser likewise
// (tests/test_host_tables.py), and everything else in the stage is exact.  MODE[1:0] = 2; the fp64 mode bits (the colour
// fallback re-evaluates the reference's expression) stay at round-to-nearest.  The frame's base pointer passes through the
// statement, so no load of a pixel — and no arithmetic on one — can be scheduled in front of the switch; so do the two tile
// coordinates, the results of the kernel's last integer divisions (the compiler expands those through v_rcp_iflag_f32 and a
// float multiply: they stay in the default mode; tests/test_abi.py checks the code object for both).
// The run kernel keeps the default mode and the integer form: measured 1 % faster there (profiles/r03_ab_history.txt).
__device__ __forceinline__ const uint8_t *pixel_stage_rounds_down(const uint8_t *frame_base, int &u0, int &u1) {
    unsigned long long p = (unsigned long long)(uintptr_t)frame_base;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2" : "+s"(p), "+s"(u0), "+s"(u1));
    return (const uint8_t *)(uintptr_t)p;
}

struct __attribute__((aligned(4))) Row24 {
    uint32_t d[6];
};

// 8 pixels of one block row (24 or 32 bytes already in registers) -> 8 raw pixels.  One "is any pixel of this row
// uncertain?" branch per row instead of one per pixel: the branch is taken by about half of the waves, and then only
// the flagged pixels redo the fp64 expression.
template <int BPP, bool LEAN, typename RowT>
__device__ __forceinline__ void convert_row(const RowT &v, const CompCoefF &k, float out[8]) {
    auto chan = [&](int j, int ch) -> uint32_t {
        int byte = BPP * j + ch;
        return (v.d[byte >> 2] >> ((byte & 3) * 8)) & 0xffu;
    };
    float lowest = 1.0f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const float t0 = component_t(chan(j, 0), chan(j, 1), chan(j, 2), k);
        const float t1 = component_t(chan(j + 1, 0), chan(j + 1, 1), chan(j + 1, 2), k);
        out[j] = clear_fraction(t0);
        out[j + 1] = clear_fraction(t1);
        lowest = fminf(fminf(lowest, t0 - out[j]), t1 - out[j + 1]);
    }
    if (lowest < kFracLow) { // rare: redo the row's flagged pixels in the reference's arithmetic
        // Written as a recomputation from the row's bytes (same instructions, same values); on the main path (aligned
        // 3-byte pixels) the compiler instead keeps the eight sums alive across the branch (measured 8 % faster than
        // recomputing: the branch is taken for half of the rows).  The input modes with more raw registers per row (4-byte
        // pixels, funnel-shifted rows) cannot afford those eight registers inside the 96-VGPR budget (200+ B of scratch
        // per lane): LEAN hides the bytes' origin behind an empty asm and so forces the recomputation.
        RowT w = v;
        if constexpr (LEAN) {
#pragma unroll
            for (int i = 0; i < (int)(sizeof(RowT) / 4); i++) asm("" : "+v"(w.d[i]));
        }
        auto chan2 = [&](int j, int ch) -> uint32_t {
            int byte = BPP * j + ch;
            return (w.d[byte >> 2] >> ((byte & 3) * 8)) & 0xffu;
        };
        const CompCoef &d = k.d;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t r = chan2(j, 0), gg = chan2(j, 1), b = chan2(j, 2);
            const float t = component_t(r, gg, b, k);
            if (t - clear_fraction(t) < kFracLow)
                out[j] = m1vf::kPxBiasF + (float)component_fp64((int)r, (int)gg, (int)b, d.k0, d.kr, d.kg, d.kb);
        }
    }
}

// 8 pixels of one block row -> 8 raw pixels.  FAST: C == 3 and the row starts 4-byte aligned.
template <bool FAST>
__device__ __forceinline__ void load_row(const uint8_t *p, int C, const CompCoefF &k, float out[8]) {
    if (FAST) {
        Row24 v = *reinterpret_cast<const Row24 *>(p);
        convert_row<3, false>(v, k, out);
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint8_t *q = p + j * C;
            out[j] = component_raw(q[0], q[1], q[2], k);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// block stage: the reference's integer FDCT (image_processing.c:192-307) in exact fp32 arithmetic, in registers:
// fdct_f32.h (m1vf::fdct_row_f, m1vf::fdct_col_f).
// ------------------------------------------------------------------------------------------------

// Truncating division by the scaled quantiser (image_processing.c:367) as one fp32 multiply by an
// inflated reciprocal rq = fl((1/d)(1+2^-20)): exact for |n| < 2^15, 1 <= d <= 4150
// (tests/test_host_tables.py checks every (n, d) pair on the host).  n arrives as a float holding the integer.
__device__ __forceinline__ int quant(float n, float rq) { return (int)(n * rq); }

// Where block `bidx` of a strip reads its 64 pixels (encoder.h:275-278 luma, :347-348 chroma).
// Returns the index of the first pixel and the row stride, both in pixels.
struct BlockSrc {
    uint32_t first;  // 32-bit on purpose (m1v_create rejects frames of 4 GiB and more): a row's address is then a uniform
    uint32_t stride; // 64-bit frame base + a 32-bit lane offset, which the load instruction adds itself
    int blk;    // 0..5 inside the macroblock: Y0 Y1 Y2 Y3 Cb Cr
    __device__ int comp() const { return blk < 4 ? 0 : blk - 3; } // 0 Y, 1 Cb, 2 Cr (derived: one register less to keep)
};
__device__ __forceinline__ BlockSrc block_source(const Geometry &g, int strip, int bidx) {
    BlockSrc s;
    int mb = bidx / 6;
    s.blk = bidx - mb * 6;
    if (s.blk < 4) {
        int x0 = strip * 16 + (s.blk & 1) * 8;
        int y0 = mb * 16 + (s.blk >> 1) * 8;
        s.first = (uint32_t)y0 * (uint32_t)g.W + (uint32_t)x0;
        s.stride = (uint32_t)g.W;
    } else { // full-resolution Cb/Cr plane addressed with stride W/2 at (x/2, y/2)
        s.first = (uint32_t)(mb * 8) * (uint32_t)g.half_w + (uint32_t)(strip * 8);
        s.stride = (uint32_t)g.half_w;
    }
    return s;
}


// raw pixels of one block -> the 64 quantised levels, natural order q[u*8+i] (BASELINE config 2 kernel)
template <bool FAST>
__device__ __forceinline__ void block_coefficients(const Geometry &g, const uint8_t *frame,
                                                   const BlockSrc &s, const float *rq, int q[64]) {
    float rows[64];
    CompCoefF k = comp_coef_f(s.comp());
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float px[8];
        load_row<FAST>(frame + (size_t)((s.first + (uint32_t)i * s.stride) * (uint32_t)g.C), g.C, k, px);
        m1vf::fdct_row_f<float, false>(px, &rows[i * 8]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float c[8];
        m1vf::fdct_col_f<float>(rows[0 * 8 + i], rows[1 * 8 + i], rows[2 * 8 + i], rows[3 * 8 + i], rows[4 * 8 + i],
                                rows[5 * 8 + i], rows[6 * 8 + i], rows[7 * 8 + i], c, i == 0 ? 8.0f * m1vf::kPxBiasF : 0.0f);
#pragma unroll
        for (int u = 0; u < 8; u++) q[u * 8 + i] = quant(c[u], rq[u * 8 + i]);
    }
}

// ------------------------------------------------------------------------------------------------
// entropy stage
// ------------------------------------------------------------------------------------------------

// One run/level code word (vlc.c:315-385 with the reference's indexing): r = run - 1 >= 0,
// level != 0.  Table code when row r holds an entry for |level| - 1 (for r = 0 that is the entry of |level| + 1, and
// "11" for |level| = 1: the table's first entry is stored as that special case), else the escape "000001" + 6-bit run
// + 8 or 16 bits of level (vlc.c:346-381); |level| >= 256 cannot be coded (the reference dereferences NULL): `bad`.
// Branch-free: both table reads use an index that is always valid and the choice is made by selects, so a wave whose
// lanes disagree executes one instruction stream.  NARROW: |level| < 128 is guaranteed (byte staging).
template <bool NARROW>
__device__ __forceinline__ void ac_code(const uint32_t *vlc, int r, int level, uint32_t &code, uint32_t &bits, uint32_t &bad) {
    const uint32_t L = (uint32_t)(level < 0 ? -level : level);
    const uint32_t info = vlc[kVlcRowInfo + min(r, kAcRows - 1)];
    const bool in_table = r < kAcRows && L - 1u < (info >> 8);
    const uint32_t e = vlc[kVlcEntries + (in_table ? (info & 0xffu) + L - 1u : 0u)];
    const uint32_t head = (1u << 6) | ((uint32_t)r & 0x3fu); // "000001" + 6-bit run
    const uint32_t lo = (uint32_t)level & 0xffu;              // (u8)(+L) or (u8)(-L)
    uint32_t esc = (head << 8) | lo, esc_bits = 20;
    if (!NARROW) {
        const bool wide = L >= 128u;
        esc = wide ? (head << 16) | (level < 0 ? 0x8000u : 0u) | lo : esc;
        esc_bits = wide ? 28 : 20;
        bad |= (!in_table && L >= 256u) ? 1u : 0u;
    }
    code = in_table ? (e & 0xffffu) : esc;
    bits = in_table ? (e >> 16) : esc_bits;
}

// Pass 1 of a block: all its bits in a 64-bit register (when they fit) and their count.
//   hdr/hlen : DC part (mpeg1_blk.c:73-102), with the macroblock header "11" (mpeg1_blk.c:38-51) in front for
//              block 0
//   emit     : bit p set = the AC coefficient at zigzag position p is coded.  VLC_encode stops at the first pair
//              with run 0 (image_processing.c:421): that is the first p >= 1 with both p-1 and p non-zero.
// The register simply keeps shifting: if the count ends above 64 its content is meaningless and the block is walked
// again by pass 2 (walk_codes); no per-code bookkeeping.
template <bool NARROW, typename Fetch>
__device__ __forceinline__ void block_bits_pass1(uint32_t hdr, int hlen, bool dc_nonzero, unsigned long long emit,
                                                 const uint32_t *vlc, Fetch fetch, unsigned long long &acc, int &tot,
                                                 uint32_t &bad) {
    acc = hdr;
    tot = hlen;
    int prev = dc_nonzero ? 0 : -1;
    while (emit) {
        const int p = __builtin_ctzll(emit);
        emit &= emit - 1;
        const int r = p - prev - 2; // (zeros before this coefficient, image_processing.c:716-722) - 1, vlc.c:326
        prev = p;
        uint32_t code, bits;
        ac_code<NARROW>(vlc, r, fetch(p), code, bits, bad);
        acc = (acc << bits) | code;
        tot += (int)bits;
    }
    acc = (acc << 2) | 0x2u; // EOB "10", mpeg1_blk.c:115-117
    tot += 2;
}

// The same walk, code word by code word into `sink` (pass 2 of the rare blocks that exceed 64 bits).
template <bool NARROW, typename Fetch, typename Sink>
__device__ __forceinline__ void walk_codes(uint32_t hdr, int hlen, bool dc_nonzero, unsigned long long emit,
                                           const uint32_t *vlc, Fetch fetch, Sink &sink) {
    sink(hdr, hlen);
    int prev = dc_nonzero ? 0 : -1;
    uint32_t bad = 0;
    while (emit) {
        const int p = __builtin_ctzll(emit);
        emit &= emit - 1;
        const int r = p - prev - 2;
        prev = p;
        uint32_t code, bits;
        ac_code<NARROW>(vlc, r, fetch(p), code, bits, bad);
        sink(code, (int)bits);
    }
    sink(0x2u, 2);
}

// MSB-first OR of `bits` code bits at absolute bit position `pos` of a zero-initialised word image.
// Words are kept big-endian-logical (bit 31 = earliest bit); SWAP stores them byte-swapped so that a
// little-endian memory image is already the byte stream (used by the global-memory fallback).
template <bool SWAP>
__device__ __forceinline__ void or_code(uint32_t *img, uint32_t pos, uint32_t code, int bits) {
    uint32_t w = pos >> 5, sh = pos & 31u;
    unsigned long long v = (unsigned long long)code << (64 - bits - (int)sh);
    uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    if (SWAP) {
        hi = __builtin_bswap32(hi);
        lo = __builtin_bswap32(lo);
    }
    if (hi) atomicOr(&img[w], hi);
    if (lo) atomicOr(&img[w + 1], lo);
}

// exclusive prefix sum over the workgroup; every thread gets its offset, `total` the grand total
__device__ __forceinline__ uint32_t block_scan_exclusive(uint32_t v, uint32_t *wave_sums, int nthreads,
                                                         uint32_t &total) {
    int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        uint32_t o = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += o;
    }
    if (lane == kWave - 1) wave_sums[wave] = incl;
    __syncthreads();
    int nw = nthreads >> 6;
    if (wave == 0) {
        uint32_t s = lane < nw ? wave_sums[lane] : 0;
        uint32_t si = s;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            uint32_t o = __shfl_up(si, d, kWave);
            if (lane >= d) si += o;
        }
        if (lane < nw) wave_sums[lane] = si - s; // exclusive
        if (lane == nw - 1) wave_sums[16] = si;
    }
    __syncthreads();
    uint32_t off = wave_sums[wave] + incl - v;
    total = wave_sums[16];
    __syncthreads();
    return off;
}

// Inclusive prefix sum across the 64 lanes of a wave with DPP row shifts / row broadcasts (gfx9 family):
// no index registers, 7 VALU adds.  row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
__device__ __forceinline__ uint32_t wave_scan_inclusive(uint32_t v) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xe, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xc, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
// inclusive prefix sum across the first 16 lanes only (one DPP row)
__device__ __forceinline__ uint32_t row_scan_inclusive(uint32_t v) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xe, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xc, false);
    return x;
}

// Same, with ONE barrier: every wave redoes the (<= 16 entry) scan of the wave totals itself.  `ws` must
// hold 2 x 16 words; callers alternate `parity` so that a buffer is rewritten only after another barrier.
__device__ __forceinline__ uint32_t block_scan_exclusive_1b(uint32_t v, uint32_t *ws, int parity, int nthreads,
                                                            uint32_t &total) {
    int lane = threadIdx.x & (kWave - 1);
    int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t incl = wave_scan_inclusive(v);
    uint32_t *buf = ws + parity * 16;
    if (lane == kWave - 1) buf[wave] = incl;
    __syncthreads();
    int nw = nthreads >> 6;
    uint32_t s = lane < nw ? buf[lane] : 0;
    uint32_t si = row_scan_inclusive(s);
    uint32_t wave_off = (uint32_t)__builtin_amdgcn_readlane((int)(si - s), wave);
    total = (uint32_t)__builtin_amdgcn_readlane((int)si, nw - 1);
    return wave_off + incl - v;
}

// XCD-aware (frame, strip) of a workgroup: consecutive workgroup ids round-robin over the 8 XCDs,
// so give each XCD whole frames (its L2 then sees every 128-byte line of the frame once).
__device__ __forceinline__ void frame_strip_of(unsigned b, int n_frames, int n_strips, int &frame,
                                               int &strip) {
    unsigned per_group = 8u * (unsigned)n_strips;
    unsigned full = (unsigned)n_frames / 8u;
    if (b < full * per_group) {
        unsigned g = b / per_group, r = b - g * per_group;
        frame = (int)(g * 8u + (r & 7u));
        strip = (int)(r >> 3);
    } else {
        unsigned t = b - full * per_group;
        frame = (int)(full * 8u + t / (unsigned)n_strips);
        strip = (int)(t % (unsigned)n_strips);
    }
}

// Division of a workgroup index by a launch constant without the compiler's float-reciprocal expansion (15 vector
// instructions per division and wave): q = high half of n * m with m = floor(2^32 / d) + 1, exact while n * d < 2^32 — the
// host checks that for the largest index of the launch and passes m = 0 (plain division) otherwise.  Scalar multiplies only.
struct DivMagic {
    uint32_t d, m;
};
M1V_HD DivMagic div_magic(uint32_t d, unsigned long long n_max) {
    DivMagic r;
    r.d = d;
    r.m = (d >= 2 && n_max * d < (1ull << 32)) ? (uint32_t)((1ull << 32) / d) + 1u : 0u;
    return r;
}
__device__ __forceinline__ uint32_t udiv(uint32_t n, const DivMagic &k) { return k.m ? __umulhi(n, k.m) : n / k.d; }
// frame_strip_of with the launch's three divisors prepared: group = 8 * per_frame, per_frame
__device__ __forceinline__ void frame_unit_of(uint32_t b, int n_frames, const DivMagic &group, const DivMagic &per_frame, int &frame,
                                              int &unit) {
    const uint32_t full = (uint32_t)n_frames / 8u;
    if (b < full * group.d) {
        const uint32_t gq = udiv(b, group), r = b - gq * group.d;
        frame = (int)(gq * 8u + (r & 7u));
        unit = (int)(r >> 3);
    } else {
        const uint32_t t = b - full * group.d, q = udiv(t, per_frame);
        frame = (int)(full * 8u + q);
        unit = (int)(t - q * per_frame.d);
    }
}

#ifndef M1V_WAVES_PER_EU
#define M1V_WAVES_PER_EU 5
#endif
#ifndef M1V_DENSE_KEEP
#define M1V_DENSE_KEEP 8 // row-pass outputs of the run kernels stay unpacked (RowStore)
#endif

// ---- pieces shared by the two encode kernels -----------------------------------------------------

// Issue the 8 row loads (8 x 24 B) of one block.
// Input modes of the dense kernel: 0 = byte loads (4 channels, or a buffer that is not 4-byte aligned),
// 1 = 3 channels and every block row starts on a 4-byte boundary (width % 8 == 0): 24-byte loads,
// 2 = 3 channels, block rows start anywhere in a 4-byte aligned buffer: 28 bytes from the dword at or below the row
//     start, funnel-shifted by the lane's misalignment (v_alignbyte_b32).
struct __attribute__((aligned(4))) Row28 {
    uint32_t d[7];
};
__device__ __forceinline__ void load_block_rows(const uint8_t *fbase, const BlockSrc &src, Row28 raw[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint8_t *p = fbase + (size_t)((src.first + (uint32_t)i * src.stride) * 3u);
        const uint32_t m = (uint32_t)(uintptr_t)p & 3u;
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p - m);
#pragma unroll
        for (int k = 0; k < 6; k++) raw[i].d[k] = q[k];
        // the seventh dword holds row bytes only when the row is misaligned; an aligned row must not touch it (it can lie
        // beyond the end of the buffer), a misaligned one shares it with a byte of the row, i.e. with a mapped page
        raw[i].d[6] = q[m ? 6 : 5];
    }
}
//     3 = 4 channels in a 4-byte aligned buffer: 32-byte rows (always aligned); rows 0..3 are requested up front, rows
//     4..7 from inside the row loop (64 raw registers would not fit next to the row-pass values).
struct __attribute__((aligned(4))) Row32 {
    uint32_t d[8];
};
__device__ __forceinline__ void load_block_rows(const uint8_t *fbase, const BlockSrc &src, Row32 raw[8]) {
#pragma unroll
    for (int i = 0; i < 4; i++)
        raw[i] = *reinterpret_cast<const Row32 *>(fbase + (size_t)((src.first + (uint32_t)i * src.stride) * 4u));
}
__device__ __forceinline__ Row24 row_bytes(const Row24 &v, const uint8_t *) { return v; }
__device__ __forceinline__ Row24 row_bytes(const Row28 &v, const uint8_t *p) {
    const uint32_t m = (uint32_t)(uintptr_t)p & 3u;
    Row24 r;
#pragma unroll
    for (int k = 0; k < 6; k++) r.d[k] = __builtin_amdgcn_alignbyte(v.d[k + 1], v.d[k], m);
    return r;
}
__device__ __forceinline__ void load_block_rows(const uint8_t *fbase, const BlockSrc &src, Row24 raw[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++)
        raw[i] = *reinterpret_cast<const Row24 *>(fbase + (size_t)((src.first + (uint32_t)i * src.stride) * 3u));
}

// rows: convert + row pass as each row's bytes arrive; columns: column pass + quantise + stage in LDS +
// non-zero mask, one column at a time (nothing but rows[] stays live).  Returns the DC level.
// The 64 outputs of the row pass, held until the column pass.  Columns KEEP..7 are stored as f16 PAIRS: every row output
// except column 0 (the plain sum, which also carries the pixel bias 8 * 256) is an integer of magnitude <= 1020, the sum
// without its bias one of magnitude <= 2040 (fdct_f32.h; tools/fdct_f32_proof.cpp checks the bounds), and f16 holds every
// integer up to 2048 exactly, so packing (v_cvt_pkrtz_f16_f32) and unpacking (v_cvt_f32_f16) lose nothing.  Column 0 always
// stays a float (KEEP >= 2).  All eight columns packed (66 VGPRs, 6 waves per SIMD) measured 0.3-2.6 % slower in a sustained
// run than all eight unpacked at 5 waves (profiles/r03_ab_history.txt) and was removed in round 4.
typedef __fp16 m1v_h2 __attribute__((ext_vector_type(2)));
template <int KEEP>
struct RowStore {
    static_assert(KEEP >= 2 && KEEP <= 8 && (8 - KEEP) % 2 == 0, "pairs of columns are packed; column 0 stays a float");
    static constexpr float kBias0 = 8.0f * m1vf::kPxBiasF; // what column 0 still carries on top of the sum
    float f[8][KEEP + 1];
    m1v_h2 h[8][(8 - KEEP) / 2 + 1];
    __device__ __forceinline__ void put(int r, const float out[8]) {
#pragma unroll
        for (int c = 0; c < KEEP; c++) f[r][c] = out[c];
#pragma unroll
        for (int c = KEEP; c < 8; c += 2) {
            m1v_h2 v = __builtin_amdgcn_cvt_pkrtz(out[c], out[c + 1]);
            // pinned here (volatile statements keep their order, and the next row's LDS read is one): left to itself the
            // scheduler sinks all packing behind the last row and the unpacked values spill
            asm volatile("" : "+v"(v));
            h[r][(c - KEEP) / 2] = v;
        }
    }
    __device__ __forceinline__ float get(int r, int c) const {
        if (c < KEEP) return f[r][c];
        return (float)h[r][(c - KEEP) / 2][(c - KEEP) & 1];
    }
};

template <int FAST, bool STAGE8, typename RowT>
__device__ __forceinline__ int block_to_stage(const Geometry &g, const uint8_t *fbase, const BlockSrc &src,
                                              const RowT raw[8], const float *rq_global, uint32_t *blk, uint32_t &lds_addr) {
    // constant address space: the table then stays a scalar load behind the volatile statements of RowStore::put
    const M1V_CONST_AS float *rq_t = reinterpret_cast<const M1V_CONST_AS float *>(reinterpret_cast<uintptr_t>(rq_global));
    RowStore<M1V_DENSE_KEEP> rows;
    CompCoefF k = comp_coef_f(src.comp());
    Row32 late[FAST == 3 ? 4 : 1];
    (void)late;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float px[8];
        if constexpr (FAST == 3) {
            if (i == 1) { // rows 4..7: requested once row 0 has been consumed
#pragma unroll
                for (int r = 0; r < 4; r++)
                    late[r] = *reinterpret_cast<const Row32 *>(fbase + (size_t)((src.first + (uint32_t)(r + 4) * src.stride) * 4u));
            }
            convert_row<4, true>(i < 4 ? raw[i] : late[i - 4], k, px);
        } else if constexpr (FAST != 0) {
            convert_row<3, FAST == 2>(row_bytes(raw[i], fbase + (size_t)((src.first + (uint32_t)i * src.stride) * 3u)), k, px);
        } else {
            load_row<false>(fbase + (size_t)((src.first + (uint32_t)i * src.stride) * (uint32_t)g.C), g.C, k, px);
        }
        float ro[8];
        m1vf::fdct_row_f<float, false>(px, ro); // default rounding mode: see pixel_stage_rounds_down
        rows.put(i, ro);
    }
    int dc = 0;
    lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)blk;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float c[8];
        m1vf::fdct_col_f<float>(rows.get(0, i), rows.get(1, i), rows.get(2, i), rows.get(3, i), rows.get(4, i), rows.get(5, i),
                                rows.get(6, i), rows.get(7, i), c, i == 0 ? RowStore<M1V_DENSE_KEEP>::kBias0 : 0.0f);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = quant(c[u], rq_t[i * 8 + u]);
            const int p = scan_pos(u * 8 + i);
            if (p == 0) dc = q; // the DC level may not fit the staged width: it stays in a register
            // one LDS byte (halfword) store per level, straight from the converted register: no packing on the vector
            // ALU.  Inline asm: plain stores are merged back into packed words by the compiler, and volatile ones (or
            // volatile asm) turn the scalar loads of the quantiser table into per-lane vector loads.  The asm
            // statements are pure as far as the compiler knows; the address register, passed in-out, chains them (and the
            // reads of stage_nonzero_mask) in program order.
            if (STAGE8)
                asm("ds_write_b8 %0, %1 offset:%2" : "+v"(lds_addr) : "v"(q), "n"(stage_byte8(p)));
            else
                asm("ds_write_b16 %0, %1 offset:%2" : "+v"(lds_addr) : "v"(q), "n"(stage_byte16(p)));
        }
    }
    return dc;
}

// Bit p set = the staged level at zigzag position p is non-zero.  Byte (or halfword) granular "non-zero" flags of a
// whole word: ((w & 0x7f..) + 0x7f..) | w has the top bit of every non-zero field set; the layout (stage_byte8/16)
// makes one shift put a word's flags on their zigzag positions.  Position 0 (DC) is the caller's.
template <bool STAGE8>
__device__ __forceinline__ unsigned long long stage_nonzero_mask(const uint32_t *blk, uint32_t lds_addr) {
    constexpr int kWords = STAGE8 ? 16 : 32;
    uint32_t w[kWords];
    // The staging stores are inline asm (block_to_stage), so the reads are too, chained behind them by lds_addr.  Eight
    // ds_read2_b32 AND their s_waitcnt in ONE statement: the compiler's waitcnt pass does not see LDS operations inside asm,
    // so nothing may sit between a read and the wait that covers it (a register copy there would copy stale bits).
    unsigned long long pr[kWords / 2];
#pragma unroll
    for (int j = 0; j < kWords / 2; j += 8)
        asm("ds_read2_b32 %0, %8 offset0:%9 offset1:%10\n\tds_read2_b32 %1, %8 offset0:%11 offset1:%12\n\t"
            "ds_read2_b32 %2, %8 offset0:%13 offset1:%14\n\tds_read2_b32 %3, %8 offset0:%15 offset1:%16\n\t"
            "ds_read2_b32 %4, %8 offset0:%17 offset1:%18\n\tds_read2_b32 %5, %8 offset0:%19 offset1:%20\n\t"
            "ds_read2_b32 %6, %8 offset0:%21 offset1:%22\n\tds_read2_b32 %7, %8 offset0:%23 offset1:%24\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(pr[j]), "=&v"(pr[j + 1]), "=&v"(pr[j + 2]), "=&v"(pr[j + 3]), "=&v"(pr[j + 4]), "=&v"(pr[j + 5]),
              "=&v"(pr[j + 6]), "=&v"(pr[j + 7]), "+v"(lds_addr)
            : "n"(2 * j), "n"(2 * j + 1), "n"(2 * j + 2), "n"(2 * j + 3), "n"(2 * j + 4), "n"(2 * j + 5), "n"(2 * j + 6),
              "n"(2 * j + 7), "n"(2 * j + 8), "n"(2 * j + 9), "n"(2 * j + 10), "n"(2 * j + 11), "n"(2 * j + 12),
              "n"(2 * j + 13), "n"(2 * j + 14), "n"(2 * j + 15));
#pragma unroll
    for (int j = 0; j < kWords / 2; j++) {
        w[2 * j] = (uint32_t)pr[j];
        w[2 * j + 1] = (uint32_t)(pr[j] >> 32);
    }
    uint32_t half[2] = {0u, 0u};
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (STAGE8) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t t = ((w[h * 8 + j] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[h * 8 + j];
                half[h] |= (t >> (7 - j)) & (0x01010101u << j);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const uint32_t t = ((w[h * 16 + j] & 0x7fff7fffu) + 0x7fff7fffu) | w[h * 16 + j];
                half[h] |= (t >> (15 - j)) & (0x00010001u << j);
            }
        }
    }
    return ((unsigned long long)half[1] << 32) | half[0];
}

template <bool STAGE8>
__device__ __forceinline__ int fetch_level(const uint32_t *blk, int p) {
    if (STAGE8) {
        const uint32_t w = blk[(p & 7) + 8 * (p >> 5)];
        return (int)(w << (24 - 8 * ((p >> 3) & 3))) >> 24;
    } else {
        const uint32_t w = blk[(p & 15) + 16 * (p >> 5)];
        return (p & 16) ? ((int)w >> 16) : ((int)(w << 16) >> 16);
    }
}

// DC part of a block (mpeg1_blk.c:73-102) with the macroblock header "11" (mpeg1_blk.c:38-51) in front
// for block 0 of a macroblock.
__device__ __forceinline__ void dc_header(int dc, bool luma, int blk, const uint32_t *vlc, uint32_t &hdr,
                                          int &hlen) {
    if (dc != 0) {
        int coe = dc < 0 ? -dc : dc;
        int low = coe & 0xff;
        int sz = low ? 32 - __builtin_clz((unsigned)low) : 1;
        uint32_t e = vlc[(luma ? kVlcDcLuma : kVlcDcChroma) + sz];
        if (dc < 0) coe ^= 1 << (sz - 1);
        hdr = ((e & 0xffffu) << sz) | ((uint32_t)coe & 0xffu & ((1u << sz) - 1u));
        hlen = (int)(e >> 16) + sz;
    } else { // "100" / "00"
        hdr = luma ? 0x4u : 0x0u;
        hlen = luma ? 3 : 2;
    }
    if (blk == 0) { // macroblock_address_increment "1" + type "1"
        hdr |= 3u << hlen;
        hlen += 2;
    }
}

// AC positions VLC_encode codes: the non-zeros below the first position whose predecessor is non-zero too
__device__ __forceinline__ unsigned long long emit_set(unsigned long long nz) {
    unsigned long long stop = nz & (nz << 1);
    unsigned long long below = stop ? ((stop & (~stop + 1)) - 1) : ~0ull;
    return nz & ~1ull & below;
}

// One lane's block bits: the first 64 in a register (the common case is the whole block), the count always.
struct BlockBits {
    unsigned long long acc;
    int tot;
    __device__ bool spilled() const { return tot > 64; }
};

// OR a block's bits into a zeroed word image at bit offset `off` (LDS image: logical big-endian words;
// global image: byte-swapped so that memory is already the byte stream).
template <bool GLOBAL, typename Walk>
__device__ __forceinline__ void put_block(uint32_t *img, uint32_t off, const BlockBits &b, Walk walk) {
    if (!b.spilled()) {
        unsigned long long A = b.acc << (64 - b.tot);
        uint32_t w = off >> 5, sh = off & 31u;
        uint32_t w0 = (uint32_t)(A >> (32 + sh));
        uint32_t w1 = (uint32_t)(A >> sh);
        uint32_t w2 = sh ? ((uint32_t)A << (32 - sh)) : 0u;
        if (GLOBAL) {
            w0 = __builtin_bswap32(w0);
            w1 = __builtin_bswap32(w1);
            w2 = __builtin_bswap32(w2);
        }
        if (w0) atomicOr(&img[w], w0);
        if (w1) atomicOr(&img[w + 1], w1);
        if (w2) atomicOr(&img[w + 2], w2);
    } else {
        uint32_t pos = off;
        auto sink = [&](uint32_t code, int bits) {
            or_code<GLOBAL>(img, pos, code, bits);
            pos += bits;
        };
        walk(sink);
    }
}

// slice header (mpeg1_blk.c:12-16): 00 00 01, strip+1 (uint8 wrap), quant_scale = 1 in 5 bits, a 0 bit: 38 bits
__device__ __forceinline__ uint32_t slice_word0(int strip) { return 0x00000100u | ((uint32_t)(strip + 1) & 0xffu); }
constexpr uint32_t kSliceWord1 = 0x08000000u;

// ---- kernel 1: one workgroup per strip (pictures with fewer than 64 blocks per strip, e.g. the
//      reference's 96x144 region: 54 blocks) ---------------------------------------------------------
template <bool FAST>
__global__ __launch_bounds__(kWave) void k_encode_strips(EncodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const Geometry &g = a.g;
    const int T = a.threads; // == 64
    const int tid = threadIdx.x;
    uint32_t *vlc = lds;                           // kVlcWords (one wave)
    uint32_t *wave_sums = vlc + kVlcWords;         // 32
    uint32_t *stage = wave_sums + 32;              // T blocks x kStageStride16 words (int16 levels)
    uint32_t *image = stage + kStageStride16 * T;  // a.lds_words

    int frame, strip;
    frame_strip_of(blockIdx.x, a.n_frames, g.n_strips, frame, strip);
    const uint8_t *fbase = a.rgb + (unsigned long long)frame * g.frame_bytes;
    uint32_t *slot32 = reinterpret_cast<uint32_t *>(
        a.scratch + ((unsigned long long)frame * g.n_strips + strip) * g.strip_cap);

    const int blocks_per_strip = g.n_mbrows * 6;
    const bool valid = tid < blocks_per_strip;
    BlockSrc src;
    Row24 raw[8];
    if (valid) {
        src = block_source(g, strip, tid);
        if (FAST) load_block_rows(fbase, src, raw);
    }
    for (int i = tid; i < kVlcWords; i += T) vlc[i] = a.tab->vlc[i];
    for (int i = tid; i < a.lds_words; i += T) image[i] = 0;
    __syncthreads();

    unsigned long long nz = 0;
    int dc = 0;
    uint32_t *blk = stage + tid * kStageStride16;
    if (valid) {
        uint32_t lds_addr;
        dc = block_to_stage<FAST ? 1 : 0, false>(g, fbase, src, raw, a.tab->rq_t, blk, lds_addr);
        nz = (stage_nonzero_mask<false>(blk, lds_addr) & ~1ull) | (dc != 0 ? 1ull : 0ull);
    }
    auto fetch = [&](int p) -> int { return fetch_level<false>(blk, p); };

    uint32_t hdr = 0, bad = 0;
    int hlen = 0;
    unsigned long long emit = 0;
    BlockBits bb = {0, 0};
    if (valid) {
        dc_header(dc, src.blk < 4, src.blk, vlc, hdr, hlen);
        emit = emit_set(nz);
        block_bits_pass1<false>(hdr, hlen, dc != 0, emit, vlc, fetch, bb.acc, bb.tot, bad);
    }
    uint32_t strip_bits;
    uint32_t off = block_scan_exclusive_1b((uint32_t)bb.tot, wave_sums, 0, T, strip_bits) + 38;
    uint32_t end_bits = 38 + strip_bits;
    const bool global_mode = ((end_bits + 63) >> 5) > (uint32_t)a.lds_words; // image too large for LDS
    uint32_t *img = image;
    if (global_mode) {
        uint32_t cap_words = g.strip_cap >> 2;
        for (uint32_t i = tid; i < cap_words; i += T) slot32[i] = 0;
        __syncthreads();
        img = slot32;
    }
    if (tid == 0) {
        uint32_t h0 = slice_word0(strip), h1 = kSliceWord1;
        atomicOr(&img[0], global_mode ? __builtin_bswap32(h0) : h0);
        atomicOr(&img[1], global_mode ? __builtin_bswap32(h1) : h1);
    }
    if (valid) {
        auto walk = [&](auto &sink) { walk_codes<false>(hdr, hlen, dc != 0, emit, vlc, fetch, sink); };
        if (global_mode)
            put_block<true>(slot32, off, bb, walk);
        else
            put_block<false>(image, off, bb, walk);
    }
    __syncthreads();
    // store the strip (zero bits pad it to a byte, encoder.h:442-443)
    if (!global_mode) {
        uint32_t nwords = (end_bits + 31) >> 5;
        for (uint32_t i = tid; i < nwords; i += T) slot32[i] = __builtin_bswap32(image[i]);
    }
    if (tid == 0) {
        const unsigned long long idx = (unsigned long long)frame * g.n_strips + strip;
        a.seg[idx] = make_uint2(end_bits, (uint32_t)((idx * g.strip_cap) >> 2));
        a.strip_ctr[idx] = end_bits;
        atomicAdd(&a.frame_bytes[frame], (unsigned long long)((end_bits + 7) >> 3)); // zero bits pad the strip to a byte, encoder.h:442-443
    }
    if (bad) atomicOr(a.status, (uint32_t)M1V_STATUS_UNENCODABLE);
}

// ---- kernel 2 (the dominant one): dense runs of blocks -----------------------------------------------
// The blocks of a frame in emission order (strip-major, then macroblock, then Y0 Y1 Y2 Y3 Cb Cr) are cut
// into runs of T consecutive blocks, one workgroup (T lanes, no idle lane) per run.  T <= blocks per
// strip, so a run touches at most two strips: segment 0 (lanes < nA) continues or starts strip s0,
// segment 1 (lanes >= nA) starts strip s0+1.  Each segment is packed on its own from a word boundary of
// the workgroup's image (with the 38-bit slice header in front when it starts a strip); k_assemble
// later concatenates the segments of a strip with the necessary bit shift.
struct DenseArgs {
    Geometry g;
    const uint8_t *rgb;
    const Tables *tab;
    uint8_t *scratch;       // [frame][run][slot_bytes] compact slots (a run whose image fits the LDS image), then the overflow
                            // arena: arena_slots x run_cap, handed out by an atomic counter to the runs that build in global memory
    uint32_t *run_meta;     // [frame][run][4]: bits of segment 0, bits of segment 1, first word of segment 1, where the run's
                            // bytes are (offset from `scratch` in 4-byte words)
    uint32_t *arena_next;   // the counter (cleared by the assemble kernel of the batch before)
    uint32_t slot_bytes, arena_slots;
    unsigned long long arena_off; // byte offset of the arena inside scratch
    uint32_t *status;
    int n_frames;
    int threads;            // T
    int runs_per_frame;
    int lds_words;          // capacity of the LDS image of the run's bits
    int zero_iters;         // ceil(lds_words / T): rounds of T words that clear it
    uint32_t run_cap;       // bytes of one arena slot: the worst case of a run
    unsigned long long *stamps;
};

template <int FAST, bool STAGE8>
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(M1V_WAVES_PER_EU, M1V_WAVES_PER_EU)))
void k_encode_dense(DenseArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const Geometry &g = a.g;
    const int T = a.threads;
    const int tid = threadIdx.x;
    constexpr int kStride = STAGE8 ? kStageStride8 : kStageStride16;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t *vlc = lds + wave * kVlcWords;        // this wave's private copy of the VLC table
    uint32_t *wave_sums = lds + (T >> 6) * kVlcWords; // 32 (16 wave totals, [16] = prefix inside the boundary wave)
    uint32_t *stage = wave_sums + 32;              // T blocks x kStride words
    uint32_t *image = stage + kStride * T;         // a.lds_words

    int frame, run;
    frame_strip_of(blockIdx.x, a.n_frames, a.runs_per_frame, frame, run);
    const uint8_t *fbase = a.rgb + (unsigned long long)frame * g.frame_bytes;
    const unsigned long long run_index = (unsigned long long)frame * a.runs_per_frame + run;
    uint32_t *slot32 = reinterpret_cast<uint32_t *>(a.scratch + run_index * a.slot_bytes); // compact slot (common case)

    STAMP_INIT();
    const int bps = g.n_mbrows * 6;                 // blocks per strip
    const int nb = g.n_strips * bps;                // blocks per frame
    const int first = run * T;                      // first block of the run
    const int s0 = first / bps;                     // strip of segment 0 (wave-uniform)
    const int pos0 = first - s0 * bps;              // position of the run's first block inside strip s0
    const int nA = min(T, bps - pos0);              // lanes of segment 0
    const int gb = first + tid;
    const bool valid = gb < nb;
    const bool in_b = tid >= nA;                    // lane belongs to segment 1 (strip s0 + 1)
    const bool has_b = first + nA < nb && nA < T;

    // ---- where this lane's block lies: before any vector load is in flight.  (With the table loads issued first, the
    //      compiler's v_mad_u64_u32 of the block arithmetic took the register of a pending table word as the unused upper
    //      half of its 64-bit addend, and every pixel load waited for that word: one full memory latency per workgroup.)
    BlockSrc src;
    using RowT = typename std::conditional<FAST == 3, Row32, typename std::conditional<FAST == 2, Row28, Row24>::type>::type;
    RowT raw[8];
    {
        // a run touches at most two strips (T <= blocks per strip), so the lane's strip needs no division; lanes past the
        // end of the frame (last run only) re-load the frame's last block
        const int strip = valid ? s0 + (in_b ? 1 : 0) : g.n_strips - 1;
        const int bidx = valid ? (in_b ? tid - nA : pos0 + tid) : bps - 1;
        src = block_source(g, strip, bidx);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- table loads first, then every pixel load of this lane's block: the in-order vmcnt lets the
    //      tables be consumed while the pixels are still in flight ----
    uint32_t vlcv[kVlcWords / kWave];
#pragma unroll
    for (int j = 0; j < kVlcWords / kWave; j++) vlcv[j] = a.tab->vlc[lane + j * kWave];
    // keep the table loads in front of the pixel loads (the scheduler otherwise hoists the pixel loads); a scheduling
    // barrier, not a memory clobber: a clobber would turn the later scalar table loads into vector loads
    __builtin_amdgcn_sched_barrier(0);
    // Every lane loads (lanes past the end of the frame — last run only — re-load the frame's last block): outside any
    // branch the sixteen loads stay countable, so the waits for the table words below are vmcnt(16) and the rows are
    // consumed as they arrive (vmcnt(14), (12), ...) instead of after the last one.
    {
        if (FAST) load_block_rows(fbase, src, raw);
    }

    // ---- workgroup prologue, under the latency of those loads ----
#pragma unroll
    for (int j = 0; j < kVlcWords / kWave; j++) vlc[lane + j * kWave] = vlcv[j];
#pragma unroll 1
    for (int k = 0; k < a.zero_iters; k++) image[k * T + tid] = 0; // the allocation is rounded up to a multiple of T words
    STAMP(0);

    unsigned long long nz = 0;
    int dc = 0;
    uint32_t *blk = stage + tid * kStride;
    if (valid) {
        uint32_t lds_addr;
        dc = block_to_stage<FAST, STAGE8, RowT>(g, fbase, src, raw, a.tab->rq_t, blk, lds_addr);
        nz = (stage_nonzero_mask<STAGE8>(blk, lds_addr) & ~1ull) | (dc != 0 ? 1ull : 0ull);
    }
    // No barrier here: pass 1 below reads only the lane's own staged levels and the wave's own copy of the VLC table.
    // The image zeroed in the prologue is first touched in pass 2, behind the barrier of the scan.
    STAMP(2);
    auto fetch = [&](int p) -> int { return fetch_level<STAGE8>(blk, p); };

    uint32_t hdr = 0, bad = 0;
    int hlen = 0;
    unsigned long long emit = 0;
    BlockBits bb = {0, 0};
    if (valid) {
        dc_header(dc, src.blk < 4, src.blk, vlc, hdr, hlen);
        emit = emit_set(nz);
        block_bits_pass1<STAGE8>(hdr, hlen, dc != 0, emit, vlc, fetch, bb.acc, bb.tot, bad);
    }
    STAMP(4);

    // ---- exclusive scan of the bit counts; PA = bits of segment 0's blocks ----
    uint32_t incl = wave_scan_inclusive((uint32_t)bb.tot);
    if (lane == kWave - 1) wave_sums[wave] = incl;
    if (tid == nA - 1) wave_sums[16] = incl;       // prefix inside the wave that holds segment 0's last lane
    __syncthreads();
    const int nw = T >> 6;
    uint32_t wsum = lane < nw ? wave_sums[lane] : 0;
    uint32_t wincl = row_scan_inclusive(wsum);
    uint32_t wave_off = (uint32_t)__builtin_amdgcn_readlane((int)(wincl - wsum), wave);
    uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)wincl, nw - 1);
    const int wA = __builtin_amdgcn_readfirstlane((nA - 1) >> 6);
    uint32_t PA = (uint32_t)__builtin_amdgcn_readlane((int)(wincl - wsum), wA) + wave_sums[16];
    uint32_t P = wave_off + incl - (uint32_t)bb.tot;
    STAMP(5);

    const uint32_t origin0 = pos0 == 0 ? 38u : 0u;          // segment 0 starts its strip
    const uint32_t bits0 = origin0 + PA;
    const uint32_t base1 = ((bits0 + 31) >> 5) + 1;         // first image word of segment 1
    const uint32_t bits1 = has_b ? 38u + (total - PA) : 0u;
    const uint32_t end_words = has_b ? base1 + ((bits1 + 31) >> 5) : ((bits0 + 31) >> 5);
    const uint32_t off = in_b ? base1 * 32u + 38u + (P - PA) : origin0 + P;

    auto slice_headers = [&](uint32_t *img, bool swapped) {
        if (pos0 == 0) {
            uint32_t h0 = slice_word0(s0), h1 = kSliceWord1;
            atomicOr(&img[0], swapped ? __builtin_bswap32(h0) : h0);
            atomicOr(&img[1], swapped ? __builtin_bswap32(h1) : h1);
        }
        if (has_b) {
            uint32_t h0 = slice_word0(s0 + 1), h1 = kSliceWord1;
            atomicOr(&img[base1], swapped ? __builtin_bswap32(h0) : h0);
            atomicOr(&img[base1 + 1], swapped ? __builtin_bswap32(h1) : h1);
        }
    };
    auto walk = [&](auto &sink) { walk_codes<STAGE8>(hdr, hlen, dc != 0, emit, vlc, fetch, sink); };
    uint32_t *meta = a.run_meta + run_index * 4;

    // ---- image too large for LDS (rare): build it with global atomics in a zeroed, worst-case sized slot taken from
    //      the overflow arena.  A branch of its own that ends the kernel, so that the common path below keeps its
    //      uniform slot pointer (merging the two pointers into one variable measured 3 % slower on the common path).
    if (end_words + 2 > (uint32_t)a.lds_words) {
        if (tid == 0) wave_sums[17] = atomicAdd(a.arena_next, 1u);
        __syncthreads();
        const uint32_t got = wave_sums[17];
        if (got >= a.arena_slots) { // arena exhausted: the caller re-encodes after m1v_reserve_scratch (M1V_E_SCRATCH)
            if (tid == 0) {
                atomicOr(a.status, (uint32_t)M1V_STATUS_SCRATCH);
                meta[0] = meta[1] = 0;
                meta[2] = meta[3] = 0;
            }
            return;
        }
        const unsigned long long where = a.arena_off + (unsigned long long)got * a.run_cap;
        uint32_t *big = reinterpret_cast<uint32_t *>(a.scratch + where);
        for (uint32_t i = tid; i < (a.run_cap >> 2); i += T) big[i] = 0;
        __syncthreads();
        if (tid == 0) {
            slice_headers(big, true);
            meta[0] = bits0;
            meta[1] = bits1;
            meta[2] = base1;
            meta[3] = (uint32_t)(where >> 2);
        }
        if (valid) put_block<true>(big, off, bb, walk);
        if (bad) atomicOr(a.status, (uint32_t)M1V_STATUS_UNENCODABLE);
        return;
    }

    // ---- common path: OR the bits into the LDS image, store it once to the run's compact slot ----
    if (tid == 0) {
        slice_headers(image, false);
        meta[0] = bits0;
        meta[1] = bits1;
        meta[2] = base1;
        meta[3] = (uint32_t)((run_index * a.slot_bytes) >> 2);
    }
    if (valid) put_block<false>(image, off, bb, walk);
    STAMP(6);
    __syncthreads();
    STAMP(7);
    for (uint32_t i = tid; i < end_words; i += T) slot32[i] = __builtin_bswap32(image[i]);
    if (bad) atomicOr(a.status, (uint32_t)M1V_STATUS_UNENCODABLE);
    STAMP(8);
}

// ------------------------------------------------------------------------------------------------
// layout + gather
// ------------------------------------------------------------------------------------------------
// PKT SEQ GOP PIC in front of a frame's strips with the 16-bit length back-patched (encoder.h:198-230, :448-453:
// (u16)(bytes after the length field's word) - 4) and the four trailing bytes (encoder.h:456-458, observed zero);
// threads 0..47 of the workgroup that assembles the frame's first strips
__device__ __forceinline__ void frame_header_and_trailer(const Tables *tab, uint8_t *out, unsigned long long fo,
                                                         unsigned long long fs, int index, int t) {
    if (t < 44) {
        uint8_t v = tab->hdr[index & 255][t];
        const uint32_t fwd = (uint32_t)((fs - 4ull) - 4ull - 4ull) & 0xffffu;
        if (t == 4) v = (uint8_t)(fwd >> 8);
        if (t == 5) v = (uint8_t)(fwd & 0xff);
        out[fo + t] = v;
    } else if (t < 48) {
        out[fo + fs - 4 + (t - 44)] = 0;
    }
}

#include "m1v_assemble.h"

// ---- run kernels: strips are concatenations of run segments ----------------------------------------
struct DenseGeom {
    int n_frames, n_strips, bps, T, runs_per_frame;
};

// Segment of strip s contributed by run w: which of the run's two segments, its bit count, its bytes.
__device__ __forceinline__ uint32_t dense_segment(const DenseGeom &d, const uint32_t *meta_frame, int w, int s,
                                                  uint32_t &word_off) {
    int s0w = (w * d.T) / d.bps;
    const uint32_t *m = meta_frame + (size_t)w * 4;
    if (s0w == s) {
        word_off = m[3];
        return m[0];
    }
    word_off = m[3] + m[2];
    return m[1];
}

// One workgroup per frame: the run segments of every strip, in the form k_assemble takes them: seg[frame][q][strip] =
// (bits, where), padded with empty segments up to `segs` per strip; the strips' bit totals; the frame's bytes.
__global__ __launch_bounds__(256) void k_dense_frame_layout(DenseGeom d, int segs, const uint32_t *run_meta, uint2 *seg,
                                                            unsigned long long *strip_ctr, unsigned long long *frame_bytes) {
    __shared__ unsigned long long wsum[4];
    const int f = blockIdx.x;
    const uint32_t *mf = run_meta + (size_t)f * d.runs_per_frame * 4;
    unsigned long long bytes = 0;
    for (int s = threadIdx.x; s < d.n_strips; s += 256) {
        const size_t i = (size_t)f * d.n_strips + s;
        const int w_lo = (s * d.bps) / d.T, w_hi = ((s + 1) * d.bps - 1) / d.T; // w_hi - w_lo + 1 <= segs
        uint32_t bits = 0;
        for (int q = 0; q < segs; q++) {
            uint32_t boff = 0, L = 0;
            if (w_lo + q <= w_hi) L = dense_segment(d, mf, w_lo + q, s, boff);
            seg[((size_t)f * segs + q) * d.n_strips + s] = make_uint2(L, boff);
            bits += L;
        }
        strip_ctr[i] = bits;
        bytes += (bits + 7) >> 3; // zero bits pad the strip to a byte, encoder.h:442-443
    }
    bytes = wave_sum_u64(bytes);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = bytes;
    __syncthreads();
    if (threadIdx.x == 0) frame_bytes[f] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

#include "m1v_tiles.h"

// ------------------------------------------------------------------------------------------------
// partial pipelines
// ------------------------------------------------------------------------------------------------
struct CoefArgs {
    Geometry g;
    const uint8_t *rgb;
    const Tables *tab;
    int16_t *out;
    int n_frames;
};

template <bool FAST>
__global__ __launch_bounds__(256) void k_coefficients(CoefArgs a) {
    const Geometry &g = a.g;
    int blocks_per_strip = g.n_mbrows * 6;
    int bidx = blockIdx.x * 256 + threadIdx.x;
    int strip = blockIdx.y, frame = blockIdx.z;
    if (bidx >= blocks_per_strip) return;
    BlockSrc s = block_source(g, strip, bidx);
    int q[64];
    block_coefficients<FAST>(g, a.rgb + (unsigned long long)frame * g.frame_bytes, s, a.tab->rq, q);
    int16_t *o = a.out + (((size_t)frame * g.n_strips + strip) * blocks_per_strip + bidx) * 64;
    uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
#pragma unroll
    for (int m = 0; m < 32; m++)
        o32[m] = ((uint32_t)q[scan_inv(2 * m)] & 0xffffu) | ((uint32_t)q[scan_inv(2 * m + 1)] << 16);
}

__global__ __launch_bounds__(256) void k_convert(const uint8_t *rgb, int C, unsigned long long npx_frame,
                                                 int n_frames, uint8_t *planes) {
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    unsigned long long total = npx_frame * (unsigned long long)n_frames;
    for (; i < total; i += (unsigned long long)gridDim.x * 256) {
        unsigned long long f = i / npx_frame, p = i - f * npx_frame;
        const uint8_t *q = rgb + i * C;
        uint32_t r = q[0], gg = q[1], b = q[2];
        uint8_t *o = planes + f * 3 * npx_frame;
        o[p] = (uint8_t)(int)(component_raw(r, gg, b, comp_coef_f(0)) - m1vf::kPxBiasF);
        o[npx_frame + p] = (uint8_t)(int)(component_raw(r, gg, b, comp_coef_f(1)) - m1vf::kPxBiasF);
        o[2 * npx_frame + p] = (uint8_t)(int)(component_raw(r, gg, b, comp_coef_f(2)) - m1vf::kPxBiasF);
    }
}

// The same planes, four pixels per lane (frames whose pixel count is a multiple of 4, 4-byte aligned buffers): the 12 or
// 16 input bytes arrive as dwords, every byte is converted once and feeds all three components, ONE fraction test covers the
// group's twelve sums (the fp64 re-evaluation runs only behind it), and each plane gets one packed dword store.
// HBM-bound: 3 or 4 bytes in, 3 bytes out per pixel.
template <int C>
__global__ __launch_bounds__(256) void k_convert4(const uint32_t *rgb, unsigned long long groups_frame, int n_frames,
                                                  uint32_t *planes) {
    const CompCoefF ky = comp_coef_f(0), kcb = comp_coef_f(1), kcr = comp_coef_f(2); // uniform: scalar registers
    const unsigned long long total = groups_frame * (unsigned long long)n_frames;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (unsigned long long)gridDim.x * 256) {
        const unsigned long long f = i / groups_frame, gidx = i - f * groups_frame;
        uint32_t w[C];
#pragma unroll
        for (int k = 0; k < C; k++) w[k] = rgb[i * C + k];
        float lowest = 1.0f, t[3][4], p[3][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            auto chan = [&](int ch) -> uint32_t {
                int byte = C * j + ch;
                return (w[byte >> 2] >> ((byte & 3) * 8)) & 0xffu;
            };
            const float r = (float)chan(0), g = (float)chan(1), b = (float)chan(2);
            t[0][j] = fmaf(r, ky.kr, fmaf(g, ky.kg, fmaf(b, ky.kb, ky.k0)));
            t[1][j] = fmaf(r, kcb.kr, fmaf(g, kcb.kg, fmaf(b, kcb.kb, kcb.k0)));
            t[2][j] = fmaf(r, kcr.kr, fmaf(g, kcr.kg, fmaf(b, kcr.kb, kcr.k0)));
#pragma unroll
            for (int c = 0; c < 3; c++) {
                p[c][j] = clear_fraction(t[c][j]);
                lowest = fminf(lowest, t[c][j] - p[c][j]);
            }
        }
        if (lowest < kFracLow) { // rare: some sum of the group is a tie (or next to one) of the reference's formulas
#pragma unroll
            for (int j = 0; j < 4; j++) {
                auto chan = [&](int ch) -> int {
                    int byte = C * j + ch;
                    return (int)((w[byte >> 2] >> ((byte & 3) * 8)) & 0xffu);
                };
#pragma unroll
                for (int c = 0; c < 3; c++)
                    if (t[c][j] - p[c][j] < kFracLow) {
                        const CompCoef d = comp_coef(c);
                        p[c][j] = m1vf::kPxBiasF + (float)component_fp64(chan(0), chan(1), chan(2), d.k0, d.kr, d.kg, d.kb);
                    }
            }
        }
        // p = 256 + value: the value is the top 8 mantissa bits
        const unsigned long long plane_words = groups_frame;
        uint32_t *o = planes + f * 3 * plane_words + gidx;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) packed |= ((__float_as_uint(p[c][j]) >> 15) & 0xffu) << (8 * j);
            o[c * plane_words] = packed;
        }
    }
}

__global__ __launch_bounds__(256) void k_subsample(const uint8_t *cb, const uint8_t *cr, int W, int H,
                                                   uint8_t *cbs, uint8_t *crs) {
    int sw = W / 2, sh = H / 2;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= sw * sh) return;
    int y = (i / sw) * 2, x = (i % sw) * 2;
    size_t a = (size_t)y * W + x, c = (size_t)(y + 1) * W + x;
    cbs[i] = (uint8_t)((cb[a] + cb[a + 1] + cb[c] + cb[c + 1]) / 4);
    crs[i] = (uint8_t)((cr[a] + cr[a + 1] + cr[c] + cr[c + 1]) / 4);
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void k_synth(uint8_t *rgb, unsigned long long bytes_per_frame,
                                               int n_frames, unsigned long long seed,
                                               unsigned long long first_index) {
    unsigned long long words = (bytes_per_frame + 7) >> 3;
    unsigned long long total = words * (unsigned long long)n_frames;
    unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    for (; i < total; i += (unsigned long long)gridDim.x * 256) {
        unsigned long long f = i / words, j = i - f * words;
        unsigned long long v = splitmix64(seed + (first_index + f) * 0x9E3779B97F4A7C15ull + j);
        uint8_t *dst = rgb + f * bytes_per_frame + 8 * j;
        unsigned long long left = bytes_per_frame - 8 * j;
        if (left >= 8 && ((uintptr_t)dst & 7) == 0) {
            *reinterpret_cast<unsigned long long *>(dst) = v;
        } else {
            for (unsigned k = 0; k < 8 && k < left; k++) dst[k] = (uint8_t)(v >> (8 * k));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side of the C-ABI
// ------------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(M1V_E_HIP, #expr ": %s", hipGetErrorString(e_));         \
    } while (0)

// scaled quantiser matrix, image_processing.c:314-343 (float scale factor, double division,
// round half away from zero, floor of 1)
void scaled_matrix(int qf, int q[64]) {
    static const unsigned char base[64] = {
        8,  16, 19, 22, 26, 27, 29, 34, 16, 16, 22, 24, 27, 29, 34, 37, 19, 22, 26, 27, 29, 34,
        34, 38, 22, 22, 26, 27, 29, 34, 37, 40, 22, 26, 27, 29, 32, 35, 40, 48, 26, 27, 29, 32,
        35, 40, 48, 58, 26, 27, 29, 34, 38, 46, 56, 69, 27, 29, 35, 38, 46, 56, 69, 83};
    if (qf < 1) qf = 1;
    if (qf > 100) qf = 100;
    float sf = qf < 50 ? (float)(5000.0 / qf) : (float)(200.0 - 2 * qf);
    for (int k = 0; k < 64; k++) {
        float prod = (float)base[k] * sf;
        int v = (int)round((double)prod / 100.0);
        q[k] = v < 1 ? 1 : v;
    }
}

// The VLC table of the kernels (layout: kVlc*).  Run/level code words without sign bit as the reference stores them
// (vlc.c:176-288), in its order, with its offset index (vlc.c:172-174); the reference's indexing rule (vlc.c:329-339)
// reads row r = run - 1 at |level| - 1, so in row 0 entry idx codes level idx + 2 — except idx 0, which the rule
// replaces by the special "11" and which is therefore stored that way here.  DC size codes: vlc.c:121-144.
void build_vlc_table(uint32_t t[kVlcWords]) {
    static const unsigned char row_len[32] = {39, 18, 5, 4, 3, 3, 3, 2, 2, 2, 2, 2, 2, 2, 2, 2,
                                              2,  1,  1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
    static const unsigned char code[110] = {
        0x04, 0x05, 0x06, 0x26, 0x21, 0x0a, 0x1d, 0x18, 0x13, 0x10, 0x1a, 0x19, 0x18, 0x17, 0x1f, 0x1e,
        0x1d, 0x1c, 0x1b, 0x1a, 0x19, 0x18, 0x17, 0x16, 0x15, 0x14, 0x13, 0x12, 0x11, 0x10, 0x18, 0x17,
        0x16, 0x15, 0x14, 0x13, 0x12, 0x11, 0x10, 0x03, 0x06, 0x25, 0x0c, 0x1b, 0x16, 0x15, 0x1f, 0x1e,
        0x1d, 0x1c, 0x1b, 0x1a, 0x19, 0x13, 0x12, 0x11, 0x10, 0x05, 0x04, 0x0b, 0x14, 0x14, 0x07, 0x24,
        0x1c, 0x13, 0x06, 0x0f, 0x12, 0x07, 0x09, 0x12, 0x05, 0x1e, 0x14, 0x04, 0x15, 0x07, 0x11, 0x05,
        0x11, 0x27, 0x10, 0x23, 0x1a, 0x22, 0x19, 0x20, 0x18, 0x0e, 0x17, 0x0d, 0x16, 0x08, 0x15, 0x1f,
        0x1a, 0x19, 0x17, 0x16, 0x1f, 0x1e, 0x1d, 0x1c, 0x1b, 0x1f, 0x1e, 0x1d, 0x1c, 0x1b};
    static const unsigned char bits[110] = {
        4,  5,  7,  8,  8,  10, 12, 12, 12, 12, 13, 13, 13, 13, 14, 14, 14, 14, 14, 14, 14, 14,
        14, 14, 14, 14, 14, 14, 14, 14, 15, 15, 15, 15, 15, 15, 15, 15, 15, 3,  6,  8,  10, 12,
        13, 13, 15, 15, 15, 15, 15, 15, 15, 16, 16, 16, 16, 4,  7,  10, 12, 13, 5,  8,  12, 13,
        5,  10, 12, 6,  10, 13, 6,  12, 16, 6,  12, 7,  12, 7,  13, 8,  13, 8,  16, 8,  16, 8,
        16, 10, 16, 10, 16, 10, 15, 12, 12, 12, 12, 12, 13, 13, 13, 13, 13, 16, 16, 16, 16, 16};
    memset(t, 0, kVlcWords * sizeof(uint32_t));
    int first = 0;
    for (int r = 0; r < kAcRows; r++) {
        t[kVlcRowInfo + r] = (uint32_t)first | ((uint32_t)row_len[r] << 8);
        first += row_len[r];
    }
    for (int e = 0; e < 110; e++) t[kVlcEntries + e] = ((uint32_t)bits[e] << 16) | code[e];
    t[kVlcEntries] = (2u << 16) | 0x3u; // run 1, |level| 1 -> "11" (vlc.c:329-334 with first == 0)
    static const unsigned char lc[9] = {0x4, 0x0, 0x1, 0x5, 0x6, 0xE, 0x1E, 0x3E, 0x7E};
    static const unsigned char lb[9] = {3, 2, 2, 3, 3, 4, 5, 6, 7};
    static const unsigned char cc[9] = {0x0, 0x1, 0x2, 0x6, 0xE, 0x1E, 0x3E, 0x7E, 0xFE};
    static const unsigned char cb[9] = {2, 2, 2, 3, 4, 5, 6, 7, 8};
    for (int i = 0; i < 9; i++) {
        t[kVlcDcLuma + i] = ((uint32_t)lb[i] << 16) | lc[i];
        t[kVlcDcChroma + i] = ((uint32_t)cb[i] << 16) | cc[i];
    }
}

void put_timestamp(uint8_t *o, uint8_t prefix, uint32_t v) { // mpeg1_enc.c:59-64, :67-71
    o[0] = (uint8_t)(prefix | ((v & 0xe0000000u) >> 28));
    o[1] = (uint8_t)((v & 0x1fe00000u) >> 21);
    o[2] = (uint8_t)(0x01 | ((v & 0x001fc000u) >> 13));
    o[3] = (uint8_t)((v & 0x00003fc0u) >> 6);
    o[4] = (uint8_t)(0x01 | ((v & 0x0000003fu) << 1));
}

// PKT(16) SEQ(12) GOP(8) PIC(8) of the frame whose uint8 `hour` is given (encoder.h:37-63,186-230)
void build_frame_header(uint8_t h[44], int W, int H, int hour) {
    memset(h, 0, 44);
    h[2] = 0x01; h[3] = 0xe0;                                  // packet, stream id 0 (mpeg1_enc.c:47-77)
    uint32_t ts = (uint32_t)(1 + 3600 * hour);
    ts = (uint32_t)((double)ts * 1.2);
    ts += 0xbeef;
    put_timestamp(h + 6, 0x31, ts);
    ts -= 0xbeef;
    put_timestamp(h + 11, 0x11, ts);
    uint8_t *s = h + 16;                                       // sequence (mpeg1_enc.c:81-94)
    unsigned w = (unsigned)W & 0xffu, hh = (unsigned)H & 0xffu; // uint8_t width/height, encoder.h:186-187
    s[2] = 0x01; s[3] = 0xb3;
    s[4] = (uint8_t)((w & 0xff0) >> 4);
    s[5] = (uint8_t)(((w & 0xf) << 4) | ((hh & 0xf00) >> 8));
    s[6] = (uint8_t)(hh & 0xff);
    s[7] = 0x14; s[8] = 0xff; s[9] = 0xff; s[10] = 0xe0; s[11] = 0x18;
    uint8_t *g = h + 28;                                       // GOP (mpeg1_enc.c:103-113)
    g[2] = 0x01; g[3] = 0xb8;
    g[4] = (uint8_t)((hour & 0x1f) << 2);
    g[5] = 0x08; g[6] = 0x00; g[7] = 0x40;
    uint8_t *p = h + 36;                                       // picture (mpeg1_enc.c:120-129)
    p[2] = 0x01; p[3] = 0x00; p[4] = 0x00; p[5] = 0x0f; p[6] = 0xff; p[7] = 0xf8;
}

} // namespace

struct m1v_encoder {
    int device;
    Geometry g;
    int qf, mode, max_frames;
    int threads;       // workgroup size of k_encode_strips
    int lds_words;
    bool dense;        // blocks per strip >= 64: k_encode_dense, else one workgroup per strip
    bool narrow;       // no AC level can reach +-128: one byte per staged level
    int dense_T, runs_per_frame;
    uint32_t run_cap;       // worst-case bytes of one run = one slot of the overflow arena
    uint32_t slot_bytes;    // compact slot of a run (what the LDS image can hold)
    uint32_t arena_slots;
    size_t arena_off;
    int image_words;        // capacity of the dense kernel's LDS image in effect
    bool reserve_worst;     // overflow arena sized for every run (m1v_reserve_scratch, or a forced tiny LDS image)
    size_t scratch_bytes;   // per batch state
    bool pipelined;         // layout + gather of batch k on `side` while batch k+1 encodes on the caller's stream
    unsigned calls;
    hipStream_t side;
    bool fast_ok;      // geometry allows the 4-byte-aligned 24-byte row loads
    int forced_mode;   // test hook (m1v_debug_set_input_mode): -1 = pick by geometry and alignment
    // Which encode kernel serves a batch.  Tiles (k_encode_tiles, m1v_tiles.h): 3-channel pictures, any width and
    // alignment — the default.  Runs (k_encode_dense / k_encode_strips): 4-channel pictures, and whatever the test
    // hooks force (m1v_debug_set_path, a forced input mode, a forced run length).
    int forced_path;   // -1 = by geometry, 0 = runs, 1 = tiles
    int forced_T;      // run length forced by m1v_debug_set_dense_threads (0 = default)
    bool tiles;        // the path configure_path set up
    int tile_cols, tile_rows, tiles_per_frame, tile_ring;
    uint32_t luma_region, chroma_region; // LDS bytes of a wave's ring / staging region
    uint32_t *d_tile_order;  // tile-row processing order of the tile kernel (tile_row_order_for), [tile_rows]
    int tile_order_rows;     // for how many tile rows d_tile_order was built
    size_t meta_bytes, seg_bytes; // sizes of run_meta and seg in effect
    int segs;               // segments per strip: tile rows (tiles), or the most runs a strip can touch (run kernels)
    Tables *d_tab;
    // What k_assemble needs to know about the output (configure_path): strips per workgroup, lanes per segment, LDS image
    int asm_group, asm_lanes_log2, asm_img_words;
    // What an encode kernel adds to and k_assemble reads.  Two sets per Batch, taken in turns: the assemble kernel of call j clears
    // the set call j + 1 will add to (the one call j - 1 used), so no launch and no memset stands between two batches.
    struct Counters {
        unsigned long long *strip_ctr;   // [frame][strip] bits of the strip (tile kernel: + arrivals << 40)
        unsigned long long *frame_bytes; // [frame] bytes of the frame's strips
        uint32_t *words;                 // [0] status bits of the encode kernel, [1] sink for callers without a status word, [2] arena counter
        int dirty_frames;                // frames of the set's last batch that nobody has cleared yet
    };
    // Everything one batch owns between its encode kernel and the end of its assembly.  Two sets, so that in
    // pipelined mode batch k+1 can encode while batch k is still being assembled.
    struct Batch {
        uint8_t *scratch;
        uint32_t *run_meta;     // run kernels: [frame][run][4]
        uint2 *seg;             // [frame][segment][strip] (bits, where): what a strip is concatenated from
        Counters ctr[2];
        unsigned turn;
        hipEvent_t enc_done, gather_done;
        bool gather_pending;
    } batch[2];
    unsigned long long *d_stamps;
    // device-side staging of the host-buffer entry points, kept between calls
    struct HostPath {
        uint8_t *d_in, *d_out, *d_planes;
        unsigned long long *d_meta;
        size_t in_cap, out_cap, planes_cap, meta_cap;
        hipStream_t copy_in, work; // upload stream; convert/encode/download stream
        hipEvent_t uploaded[2];
    } hp;
    // profiling
    bool prof;
    std::vector<hipEvent_t> ev;
    size_t ev_used;
};

template <typename T>
static hipError_t ensure_device(T **p, size_t *cap, size_t need) {
    if (need <= *cap) return hipSuccess;
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    hipError_t err = hipMalloc(p, need);
    if (err == hipSuccess) *cap = need;
    return err;
}

extern "C" {

const char *m1v_last_error(void) { return g_err; }

int m1v_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int m1v_warm_up(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr)); // creates the context
    hipFuncAttributes attr;    // loads this library's code object for the device
    HIP_TRY(hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_assemble<false>)));
    return M1V_OK;
}

size_t m1v_file_prolog(uint8_t out[27]) {
    static const uint8_t pack[9] = {0x00, 0x00, 0x01, 0xba, 0x21, 0x00, 0x01, 0x00, 0x01};
    memcpy(out, pack, 9);
    uint32_t rate = (2202035u & 0x3fffffu) | 0x400000u; // mpeg1_enc.c:14-16
    rate = (rate << 1) | 1u;
    out[9] = (uint8_t)(rate >> 16); out[10] = (uint8_t)(rate >> 8); out[11] = (uint8_t)rate;
    uint8_t *s = out + 12;                                 // mpeg1_enc.c:24-44, packet_num 0xe6
    s[0] = 0; s[1] = 0; s[2] = 1; s[3] = 0xbb; s[4] = 0; s[5] = 9;
    s[6] = (uint8_t)(rate >> 16); s[7] = (uint8_t)(rate >> 8); s[8] = (uint8_t)rate;
    s[9] = 0; s[10] = 0x21; s[11] = 0xff; s[12] = 0xe0; s[13] = 0xe0; s[14] = 0xe6;
    return 27;
}

// Chooses the encode kernel and its geometry and (re)allocates its scratch.  e->forced_T: run length of the run kernels
// (0 = default: 256 blocks = 4 waves, one per SIMD, so that 5 workgroups of 96-VGPR waves share a CU; or the largest
// multiple of 64 that the strip holds when it has fewer than 256 blocks).
// New buffers are allocated FIRST and swapped in, together with the geometry they belong to, only when every allocation
// has succeeded: a failed call (the worst-case arena of m1v_reserve_scratch is large) leaves the encoder as it was.
#ifndef M1V_TILE_RING
#define M1V_TILE_RING 2
#endif
// The order in which a frame's tile rows are processed.  Tile row R (macroblock rows 4R..4R+3) reads its luma from picture
// rows [64R, 64R+64) and — the chroma quirk, encoder.h:347-348 — its chroma from rows [16R, 16R+16), i.e. from one quarter of
// the luma region of tile row R/4.  Every byte of the top quarter of the picture is therefore read twice, once as luma and once
// as some other tile row's chroma, and the second read is an L2 hit only if few tile rows pass between the two (an XCD's L2 holds
// 5.7 tile rows of a 3840x2160 frame, 11 of a 1920x1080 one).  Which read comes first does not matter.  So this is the layout of
// the tree "R is the parent of 4R .. 4R+3" on a line that keeps parents and children close: every node sits in the MIDDLE of
// its children, the two smallest subtrees directly beside it, the larger ones outside.  Top to bottom (round 2) four chroma
// reads in five miss at 4K (HBM traffic 1.22x the algorithmic bytes); a depth-first walk (round 3) still loses the later
// children of every inner node (1.07x); this order loses three edges of 33 (1.02x in an LRU model of the L2, tools/l2_order_sim.py).
static void tile_row_order_for(int tile_rows, std::vector<uint32_t> &order) {
    std::vector<int> size((size_t)tile_rows, 1);
    for (int r = tile_rows - 1; r >= 1; r--) size[(size_t)(r / 4)] += size[(size_t)r]; // children have larger indices than parents
    struct Arrange {
        int tile_rows;
        const std::vector<int> &size;
        std::vector<uint32_t> run(int r) const {
            std::vector<int> ch;
            for (int c = 4 * r; c < 4 * r + 4; c++)
                if (c > 0 && c < tile_rows) ch.push_back(c);
            std::stable_sort(ch.begin(), ch.end(), [&](int x, int y) { return size[(size_t)x] < size[(size_t)y]; });
            std::vector<uint32_t> left, right;
            for (size_t i = 0; i < ch.size(); i++) {
                const std::vector<uint32_t> a = run(ch[i]);
                if (i == 0) left.insert(left.end(), a.begin(), a.end());          // smallest: directly in front of r
                else if (i == 1) right.insert(right.begin(), a.begin(), a.end()); // next: directly behind r
                else if (i == 2) left.insert(left.begin(), a.begin(), a.end());   // the larger ones outside
                else right.insert(right.end(), a.begin(), a.end());
            }
            left.push_back((uint32_t)r);
            left.insert(left.end(), right.begin(), right.end());
            return left;
        }
    };
    order = Arrange{tile_rows, size}.run(0);
}

static int g_fail_alloc_in = 0; // test hook (m1v_debug_fail_alloc): the n-th allocation of configure_path from now fails
static hipError_t plan_malloc(void **p, size_t bytes) {
    if (g_fail_alloc_in > 0 && --g_fail_alloc_in == 0) return hipErrorOutOfMemory;
    return hipMalloc(p, bytes);
}
static int configure_path(m1v_encoder *e) {
    const int dense_T = e->forced_T;
    const Geometry &g = e->g;
    const int bps = g.n_mbrows * 6;
    struct {
        bool tiles;
        int tile_cols, tile_rows, tiles_per_frame, tile_ring;
        uint32_t luma_region, chroma_region;
        int dense_T, runs_per_frame;
        uint32_t run_cap, slot_bytes, arena_slots;
        size_t arena_off;
        int image_words;
    } plan = {};
    // Tiles for every 3-channel picture: 1.3-4x faster than the run kernel where that cannot use its aligned 24-byte row loads
    // (widths that are not a multiple of 8, buffers off a 4-byte boundary), 1 % faster at 4K (the order of the tile rows keeps
    // the chroma re-reads in L2, tile_row_order_for), and 0.5-1 % faster per step on aligned 1080p in a sustained run
    // (profiles/r03_ab_history.txt).  The run kernel serves 4-channel input and the m1v_debug_set_* hooks.
    plan.tiles = g.C == 3 && e->forced_path != 0 && e->forced_mode < 0 && !(e->forced_path < 0 && dense_T > 0);
    size_t need, meta = 0, segb = 0;
    int segs = 0; // segments per strip
    if (plan.tiles) {
        plan.tile_cols = (g.n_strips + kTileStrips - 1) / kTileStrips;
        plan.tile_rows = (g.n_mbrows + kTileMbRows - 1) / kTileMbRows;
        plan.tiles_per_frame = plan.tile_cols * plan.tile_rows;
        plan.tile_ring = M1V_TILE_RING;
        const uint32_t stage = (uint32_t)(kWave * (e->narrow ? kStageStride8 : kStageStride16) * 4);
        plan.luma_region = plan.chroma_region = (std::max<uint32_t>((uint32_t)plan.tile_ring * kTileSlot, stage) + 15u) & ~15u;
        // worst case of a tile: 8 word-aligned segments of 24 blocks of <= 886 + 2 bits, 8 slice headers, slack
        plan.run_cap = (uint32_t)(((((size_t)kTileThreads * (kMaxBlockBits + 2) + kTileStrips * (38 + 32) + 64 + 7) / 8) + 32 + 15) & ~(size_t)15);
        // LDS image of the tile's bits (192 blocks: ~115 words at quality 12 on noise), scaled with the quantiser like the
        // run kernels' (512 words per 256 blocks at quality <= 25) + the segments' word alignment and slice headers
        plan.image_words = e->lds_words > 0 ? e->lds_words : (e->qf <= 25 ? 400 : (e->qf <= 50 ? 784 : (e->qf <= 76 ? 1552 : 3088)));
        plan.image_words = (plan.image_words + 3) & ~3; // cleared 16 bytes per lane
        plan.slot_bytes = (uint32_t)((((size_t)plan.image_words * 4 + 127) & ~(size_t)127) | 128);
        const size_t runs = (size_t)e->max_frames * plan.tiles_per_frame;
        plan.arena_slots = (uint32_t)(e->reserve_worst ? runs : (runs / 256 > 32 ? runs / 256 : (runs < 32 ? runs : 32)));
        plan.arena_off = runs * plan.slot_bytes;
        need = plan.arena_off + (size_t)plan.arena_slots * plan.run_cap;
        if ((need >> 2) >= (1ull << 32)) return fail(M1V_E_ARG, "scratch beyond 16 GiB: lower max_frames%s");
        segs = plan.tile_rows;
    } else if (e->dense) {
        int T = dense_T > 0 ? dense_T : (bps >= 256 ? 256 : (bps / kWave) * kWave);
        if (T < kWave || T > 384 || T % kWave || T > bps) return fail(M1V_E_ARG, "bad dense run length%s");
        plan.dense_T = T;
        int nb = g.n_strips * bps;
        plan.runs_per_frame = (nb + T - 1) / T;
        // worst case of a run: two word-aligned segments of at most T blocks of <= 886 + 2 bits, two slice headers, slack
        plan.run_cap = (uint32_t)(((((size_t)T * (kMaxBlockBits + 2) + 2 * 38 + 3 * 32 + 7) / 8) + 32 + 15) & ~(size_t)15);
        // LDS image of the run's bits: zeroing it costs time, outgrowing it the slow global-atomics path.  A run of 256
        // blocks needs ~150 words at quality 12 on noise; scale the default with the quantiser (finer quantisers emit
        // more bits per block).  The compact scratch slot of a run is exactly that image.
        plan.image_words = e->lds_words > 0 ? e->lds_words : (e->qf <= 25 ? 512 : (e->qf <= 50 ? 1024 : (e->qf <= 76 ? 2048 : 4096)));
        // + 128: an odd number of 128-byte lines, so that the slots (of which only the first third is written at quality
        // 12) do not all start on the same few memory channels (a power-of-two stride measured 3 % slower)
        plan.slot_bytes = (uint32_t)((((size_t)plan.image_words * 4 + 127) & ~(size_t)127) | 128);
        // Scratch: one compact slot per run + an overflow arena of worst-case slots for the runs whose image outgrows LDS
        // (handed out by an atomic counter).  By default the arena holds 1/256 of the runs (quality 12 noise needs none);
        // m1v_reserve_scratch(enc, 1) sizes it for all of them — what every run had in round 1, 47x the payload.
        const size_t runs = (size_t)e->max_frames * plan.runs_per_frame;
        plan.arena_slots = (uint32_t)(e->reserve_worst ? runs : (runs / 256 > 32 ? runs / 256 : (runs < 32 ? runs : 32)));
        plan.arena_off = runs * plan.slot_bytes;
        need = plan.arena_off + (size_t)plan.arena_slots * plan.run_cap;
        if ((need >> 2) >= (1ull << 32)) return fail(M1V_E_ARG, "scratch beyond 16 GiB: lower max_frames%s");
        meta = runs * 4 * sizeof(uint32_t);
        segs = (bps + T - 1) / T + 1; // the most runs whose blocks one strip can hold
    } else {
        need = (size_t)e->max_frames * g.n_strips * g.strip_cap;
        if ((need >> 2) >= (1ull << 32)) return fail(M1V_E_ARG, "scratch beyond 16 GiB: lower max_frames%s");
        segs = 1; // a strip is one piece
    }
    // k_assemble (m1v_assemble.h): strips per workgroup so that a group's bytes fit its 14-KiB LDS image (eight workgroups per CU) in one pass on noise at
    // this quality (19 bits per block at quality 12; anything larger takes more passes), a power of two (the eight
    // strips of a tile column share their scratch lines); lanes per segment (four words each) ~ the words a segment holds.
    int asm_group, asm_lanes_log2;
    {
        const int qscale = e->qf <= 25 ? 1 : (e->qf <= 50 ? 2 : (e->qf <= 76 ? 4 : 8));
        const size_t bpb = 22u * (size_t)qscale;
        const size_t strip_est = (38 + (size_t)g.n_mbrows * (2 + 6 * bpb)) / 8 + 1;
        asm_group = 1;
        while (asm_group < kAsmMaxGroup && (size_t)(2 * asm_group) * strip_est * 5 / 4 <= kAsmImageBytes && 2 * asm_group <= g.n_strips) asm_group *= 2;
        const size_t seg_blocks = plan.tiles ? (size_t)kTileSegBlocks : (e->dense ? (size_t)plan.dense_T : (size_t)bps);
        const size_t seg_words = seg_blocks * bpb / 32;
        asm_lanes_log2 = 1; // four words per lane: enough lanes for 1.6 x the expected words, at most one DPP row
        while (asm_lanes_log2 < 4 && ((size_t)4 << asm_lanes_log2) * 5 < seg_words * 8) asm_lanes_log2++;
    }
    const int sets = e->pipelined ? 2 : 1;
    segb = (size_t)e->max_frames * g.n_strips * segs * sizeof(uint2);
    // ---- allocate EVERYTHING the new plan needs first (every device allocation through plan_malloc: the fault-injection hook
    //      reaches all of them); nothing of the encoder is touched until all of it is there ----
    struct Fresh {
        uint8_t *scratch;
        void *meta, *seg;
        bool new_scratch, new_meta, new_seg, new_fixed;
        m1v_encoder::Counters ctr[2];
        hipEvent_t enc_done, gather_done;
    } fresh[2] = {};
    uint32_t *fresh_order = nullptr;
    const bool new_order = plan.tiles && e->tile_order_rows != plan.tile_rows;
    const size_t nslots = (size_t)e->max_frames * g.n_strips;
    bool ok = true;
    for (int i = 0; i < sets && ok; i++) {
        const m1v_encoder::Batch &bt = e->batch[i];
        Fresh &f = fresh[i];
        f.new_scratch = need != e->scratch_bytes || !bt.scratch;
        f.new_meta = meta != 0 && (meta != e->meta_bytes || !bt.run_meta);
        f.new_seg = segb != 0 && (segb != e->seg_bytes || !bt.seg);
        f.new_fixed = !bt.enc_done;
        if (f.new_scratch) ok = plan_malloc((void **)&f.scratch, need) == hipSuccess;
        if (ok && f.new_meta) ok = plan_malloc(&f.meta, meta) == hipSuccess;
        if (ok && f.new_seg) ok = plan_malloc(&f.seg, segb) == hipSuccess;
        if (ok && f.new_fixed) {
            for (m1v_encoder::Counters &c : f.ctr) {
                ok = ok && plan_malloc((void **)&c.strip_ctr, nslots * 8) == hipSuccess && hipMemset(c.strip_ctr, 0, nslots * 8) == hipSuccess;
                ok = ok && plan_malloc((void **)&c.frame_bytes, (size_t)e->max_frames * 8) == hipSuccess &&
                     hipMemset(c.frame_bytes, 0, (size_t)e->max_frames * 8) == hipSuccess;
                ok = ok && plan_malloc((void **)&c.words, 4 * sizeof(uint32_t)) == hipSuccess && hipMemset(c.words, 0, 4 * sizeof(uint32_t)) == hipSuccess;
                c.dirty_frames = 0;
            }
            ok = ok && hipEventCreateWithFlags(&f.gather_done, hipEventDisableTiming) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&f.enc_done, hipEventDisableTiming) == hipSuccess;
        }
    }
    if (ok && new_order) {
        std::vector<uint32_t> order;
        tile_row_order_for(plan.tile_rows, order);
        ok = plan_malloc((void **)&fresh_order, order.size() * sizeof(uint32_t)) == hipSuccess &&
             hipMemcpy(fresh_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) {
        for (Fresh &f : fresh) {
            (void)hipFree(f.scratch);
            (void)hipFree(f.meta);
            (void)hipFree(f.seg);
            for (m1v_encoder::Counters &c : f.ctr) {
                (void)hipFree(c.strip_ctr);
                (void)hipFree(c.frame_bytes);
                (void)hipFree(c.words);
            }
            if (f.enc_done) (void)hipEventDestroy(f.enc_done);
            if (f.gather_done) (void)hipEventDestroy(f.gather_done);
        }
        (void)hipFree(fresh_order);
        (void)hipGetLastError();
        return fail(M1V_E_HIP, "allocation failed (the encoder keeps its previous configuration)%s");
    }
    // ---- commit ----
    for (int i = 0; i < sets; i++) {
        m1v_encoder::Batch &bt = e->batch[i];
        Fresh &f = fresh[i];
        if (f.new_scratch) {
            (void)hipFree(bt.scratch);
            bt.scratch = f.scratch;
        }
        if (f.new_meta) {
            (void)hipFree(bt.run_meta);
            bt.run_meta = (uint32_t *)f.meta;
        }
        if (f.new_seg) {
            (void)hipFree(bt.seg);
            bt.seg = (uint2 *)f.seg;
        }
        if (f.new_fixed) {
            bt.ctr[0] = f.ctr[0];
            bt.ctr[1] = f.ctr[1];
            bt.turn = 0;
            bt.enc_done = f.enc_done;
            bt.gather_done = f.gather_done;
        }
    }
    if (new_order) {
        (void)hipFree(e->d_tile_order);
        e->d_tile_order = fresh_order;
        e->tile_order_rows = plan.tile_rows;
    }
    e->tiles = plan.tiles;
    e->tile_cols = plan.tile_cols; e->tile_rows = plan.tile_rows; e->tiles_per_frame = plan.tiles_per_frame;
    e->tile_ring = plan.tile_ring;
    e->luma_region = plan.luma_region; e->chroma_region = plan.chroma_region;
    e->dense_T = plan.dense_T; e->runs_per_frame = plan.runs_per_frame;
    e->run_cap = plan.run_cap; e->image_words = plan.image_words; e->slot_bytes = plan.slot_bytes;
    e->arena_slots = plan.arena_slots; e->arena_off = plan.arena_off;
    e->scratch_bytes = need;
    e->meta_bytes = meta ? meta : e->meta_bytes; // (a path without run metadata keeps the other path's array and its size)
    e->seg_bytes = segb ? segb : e->seg_bytes;
    e->segs = segs;
    e->asm_group = asm_group;
    e->asm_lanes_log2 = asm_lanes_log2;
    e->asm_img_words = kAsmImageBytes / 4;
    return M1V_OK;
}

int m1v_create(m1v_encoder **out, int device, int width, int height, int channels,
               int quality_factor, int mode, int max_frames) {
    if (!out) return fail(M1V_E_ARG, "null out%s");
    *out = nullptr;
    if (width <= 0 || height <= 0 || channels < 3 || channels > 4 || max_frames <= 0)
        return fail(M1V_E_ARG, "bad geometry%s");
    if (mode != M1V_MODE_STRICT && mode != M1V_MODE_FULL) return fail(M1V_E_ARG, "bad mode%s");
    int xe = mode == M1V_MODE_FULL ? (width & ~15) : 96;
    int ye = mode == M1V_MODE_FULL ? (height & ~15) : 144;
    if (xe > width || ye > height)
        return fail(M1V_E_ARG, "picture smaller than the 96x144 region the reference encodes%s");
    if (xe == 0 || ye == 0) return fail(M1V_E_ARG, "picture smaller than one macroblock%s");
    if ((unsigned long long)width * height * channels >= (1ull << 32))
        return fail(M1V_E_ARG, "a frame of 4 GiB or more (byte offsets inside a frame are 32-bit)%s");
    int n = m1v_device_count();
    if (n <= 0) return fail(M1V_E_NODEVICE, "no HIP device%s");
    if (device < 0 || device >= n) return fail(M1V_E_ARG, "device index out of range%s");
    HIP_TRY(hipSetDevice(device));

    m1v_encoder *e = new m1v_encoder();
    e->device = device;
    e->qf = quality_factor;
    e->mode = mode;
    e->max_frames = max_frames;
    Geometry &g = e->g;
    g.W = width; g.H = height; g.C = channels;
    g.n_strips = xe / 16; g.n_mbrows = ye / 16;
    g.half_w = width / 2;
    g.frame_bytes = (unsigned long long)width * height * channels;
    unsigned long long strip_bits = 38ull + (unsigned long long)g.n_mbrows * (2 + 6 * kMaxBlockBits);
    g.strip_cap = (uint32_t)((((strip_bits + 7) / 8) + 16 + 15) & ~15ull);
    int bps = g.n_mbrows * 6;
    e->threads = kWave;               // strip-per-workgroup kernel: only used when bps < 64
    e->lds_words = 0;                 // 0 = default of the kernel in use
    e->dense = bps >= kWave;
    e->dense_T = 0;
    e->scratch_bytes = 0;
    e->reserve_worst = false;
    e->pipelined = false;
    e->calls = 0;
    e->side = nullptr;
    memset(e->batch, 0, sizeof e->batch);
    memset(&e->hp, 0, sizeof e->hp);
    e->fast_ok = channels == 3 && (width % 8) == 0;
    e->forced_mode = -1;
    e->forced_path = -1;
    e->forced_T = 0;
    e->tiles = false;
    e->tile_cols = e->tile_rows = e->tiles_per_frame = e->tile_ring = 0;
    e->luma_region = e->chroma_region = 0;
    e->meta_bytes = e->seg_bytes = 0;
    e->segs = 0;
    e->d_tile_order = nullptr;
    e->tile_order_rows = 0;
    e->runs_per_frame = 0;
    e->run_cap = e->slot_bytes = e->arena_slots = 0;
    e->arena_off = 0;
    e->image_words = 0;
    e->prof = false;
    e->ev_used = 0;
    e->d_stamps = nullptr;

    Tables *t = new Tables();
    int q[64];
    scaled_matrix(quality_factor, q);
    for (int k = 0; k < 64; k++) t->rq[k] = (float)((1.0 / q[k]) * (1.0 + 1.0 / 1048576.0));
    for (int u = 0; u < 8; u++)
        for (int i = 0; i < 8; i++) t->rq_t[i * 8 + u] = t->rq[u * 8 + i];
    // One byte per staged level is exact iff no AC level can reach +-128.  |AC coefficient| of the
    // reference's FDCT on u8 pixels is at most 1022 (127.5 * 8 + the +2 rounding bias, reached at (0,4), (4,0),
    // (4,4); tests/test_host_tables.py::test_fdct_output_range), so 128 * (smallest AC divisor) >= 1024
    // suffices: quality factors <= 76.
    int min_ac = q[1];
    for (int k = 1; k < 64; k++) min_ac = q[k] < min_ac ? q[k] : min_ac;
    e->narrow = min_ac >= 8;
    build_vlc_table(t->vlc);
    for (int h = 0; h < 256; h++) build_frame_header(t->hdr[h], width, height, h);

    hipError_t err = hipMalloc(&e->d_tab, sizeof(Tables));
    if (err == hipSuccess) err = hipMemcpy(e->d_tab, t, sizeof(Tables), hipMemcpyHostToDevice);
    delete t;
#if defined(M1V_STAMPS) || defined(M1V_TILE_STAMPS) || defined(M1V_ASM_STAMPS)
    if (err == hipSuccess) err = hipMalloc(&e->d_stamps, (32 + 8 * 65536) * 8); // [32] phase sums, then a timeline of 8 stamps per workgroup
    if (err == hipSuccess) err = hipMemset(e->d_stamps, 0, (32 + 8 * 65536) * 8);
#endif
    if (err == hipSuccess) err = configure_path(e) == M1V_OK ? hipSuccess : hipErrorOutOfMemory;
    const void *kernels[] = {(const void *)&k_encode_dense<1, true>, (const void *)&k_encode_dense<1, false>,
                             (const void *)&k_encode_dense<2, true>, (const void *)&k_encode_dense<2, false>,
                             (const void *)&k_encode_dense<3, true>, (const void *)&k_encode_dense<3, false>,
                             (const void *)&k_encode_dense<0, true>, (const void *)&k_encode_dense<0, false>,
                             (const void *)&k_encode_strips<true>,       (const void *)&k_encode_strips<false>,
                             (const void *)&k_encode_tiles<true, M1V_TILE_RING>, (const void *)&k_encode_tiles<false, M1V_TILE_RING>};
    for (const void *kf : kernels)
        if (err == hipSuccess) err = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) {
        fail(M1V_E_HIP, "allocation failed: %s", hipGetErrorString(err));
        m1v_destroy(e);
        return M1V_E_HIP;
    }
    *out = e;
    return M1V_OK;
}

void m1v_destroy(m1v_encoder *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    for (hipEvent_t ev : e->ev) (void)hipEventDestroy(ev);
    (void)hipFree(e->d_tab);
    for (m1v_encoder::Batch &bt : e->batch) {
        (void)hipFree(bt.scratch);
        (void)hipFree(bt.run_meta);
        (void)hipFree(bt.seg);
        for (m1v_encoder::Counters &c : bt.ctr) {
            (void)hipFree(c.strip_ctr);
            (void)hipFree(c.frame_bytes);
            (void)hipFree(c.words);
        }
        if (bt.enc_done) (void)hipEventDestroy(bt.enc_done);
        if (bt.gather_done) (void)hipEventDestroy(bt.gather_done);
    }
    if (e->side) (void)hipStreamDestroy(e->side);
    (void)hipFree(e->hp.d_in);
    (void)hipFree(e->hp.d_out);
    (void)hipFree(e->hp.d_planes);
    (void)hipFree(e->hp.d_meta);
    if (e->hp.copy_in) (void)hipStreamDestroy(e->hp.copy_in);
    if (e->hp.work) (void)hipStreamDestroy(e->hp.work);
    for (hipEvent_t ev : e->hp.uploaded)
        if (ev) (void)hipEventDestroy(ev);
    (void)hipFree(e->d_stamps);
    (void)hipFree(e->d_tile_order);
    delete e;
}

int m1v_strips(const m1v_encoder *e) { return e ? e->g.n_strips : 0; }
int m1v_mb_rows(const m1v_encoder *e) { return e ? e->g.n_mbrows : 0; }
size_t m1v_frame_bytes_in(const m1v_encoder *e) { return e ? (size_t)e->g.frame_bytes : 0; }

size_t m1v_frame_bound_for(int width, int height, int mode) {
    const int xe = mode == M1V_MODE_FULL ? (width & ~15) : 96, ye = mode == M1V_MODE_FULL ? (height & ~15) : 144;
    if (width <= 0 || height <= 0 || xe <= 0 || ye <= 0 || xe > width || ye > height) return 0;
    const size_t strip_bits = 38 + (size_t)(ye / 16) * (2 + 6 * kMaxBlockBits);
    return 44 + (size_t)(xe / 16) * ((strip_bits + 7) / 8) + 4;
}

size_t m1v_frame_bound(const m1v_encoder *e) {
    if (!e) return 0;
    size_t strip_bits = 38 + (size_t)e->g.n_mbrows * (2 + 6 * kMaxBlockBits);
    return 44 + (size_t)e->g.n_strips * ((strip_bits + 7) / 8) + 4;
}

int m1v_debug_set_lds_words(m1v_encoder *e, int words) {
    if (!e) return fail(M1V_E_ARG, "null encoder%s");
    const int before = e->lds_words;
    e->lds_words = words > 0 ? (words < 4 ? 4 : words) : 0;
    if (!e->dense && !e->tiles) return M1V_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const int rc = configure_path(e); // (a small forced image sends many runs to the overflow arena: M1V_STATUS_SCRATCH)
    if (rc != M1V_OK) e->lds_words = before;
    return rc;
}

int m1v_reserve_scratch(m1v_encoder *e, int worst_case) {
    if (!e) return fail(M1V_E_ARG, "null encoder%s");
    if (!e->dense && !e->tiles) return M1V_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const bool before = e->reserve_worst;
    e->reserve_worst = worst_case != 0;
    const int rc = configure_path(e);
    if (rc != M1V_OK) e->reserve_worst = before; // the previous arena is still in place
    return rc;
}

size_t m1v_scratch_bytes(const m1v_encoder *e) { return e ? e->scratch_bytes * (e->pipelined ? 2 : 1) : 0; }

int m1v_set_pipelined(m1v_encoder *e, int enable) {
    if (!e) return fail(M1V_E_ARG, "null encoder%s");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    for (m1v_encoder::Batch &bt : e->batch) bt.gather_pending = false;
    const bool before = e->pipelined;
    e->pipelined = enable != 0;
    e->calls = 0;
    if (e->pipelined && !e->side) {
        // highest priority: the few memory-bound workgroups of layout + gather should take the first slots the (much longer,
        // arithmetic-bound) encode kernel of the next batch frees, not queue behind its whole grid
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, greatest));
    }
    const int rc = configure_path(e);
    if (rc != M1V_OK) e->pipelined = before; // the second set of buffers could not be allocated: stay as we were
    return rc;
}

int m1v_flush(m1v_encoder *e, void *stream) {
    if (!e) return fail(M1V_E_ARG, "null encoder%s");
    HIP_TRY(hipSetDevice(e->device));
    // The mark stays set: a later m1v_encode_device on ANOTHER stream must still wait for this set's gather before its
    // encode kernel overwrites the scratch (waiting for an event that has completed costs nothing).
    for (m1v_encoder::Batch &bt : e->batch)
        if (bt.gather_pending) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, bt.gather_done, 0));
    return M1V_OK;
}

int m1v_debug_set_input_mode(m1v_encoder *e, int mode) {
    if (!e) return fail(M1V_E_ARG, "null encoder%s");
    if (mode != -1 && mode != 0 && mode != 2) return fail(M1V_E_ARG, "input mode must be -1 (auto), 0 (byte loads) or 2 (funnel)%s");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const int before = e->forced_mode;
    e->forced_mode = mode; // an input mode is a property of the run kernels: forcing one selects them
    const int rc = configure_path(e);
    if (rc != M1V_OK) e->forced_mode = before;
    return rc;
}

int m1v_debug_set_dense_threads(m1v_encoder *e, int threads) {
    if (!e) return fail(M1V_E_ARG, "null encoder%s");
    if (!e->dense) return M1V_OK;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const int before = e->forced_T;
    e->forced_T = threads > 0 ? threads : 0; // a run length is a property of the run kernels: forcing one selects them
    const int rc = configure_path(e);
    if (rc != M1V_OK) e->forced_T = before;
    return rc;
}

int m1v_debug_set_path(m1v_encoder *e, int path) {
    if (!e) return fail(M1V_E_ARG, "null encoder%s");
    if (path < -1 || path > 1) return fail(M1V_E_ARG, "path must be -1 (by geometry), 0 (runs) or 1 (tiles)%s");
    if (path == 1 && e->g.C != 3) return fail(M1V_E_ARG, "the tile kernel takes 3-channel pictures%s");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    const int before = e->forced_path;
    e->forced_path = path;
    const int rc = configure_path(e);
    if (rc != M1V_OK) e->forced_path = before;
    return rc;
}

int m1v_path_in_use(const m1v_encoder *e) { return e ? (e->tiles ? 1 : 0) : -1; }

void m1v_debug_fail_alloc(int nth) {
    // fault injection for the tests: inert unless the process was started with EC504_DEBUG_HOOKS=1
    const char *on = getenv("EC504_DEBUG_HOOKS");
    g_fail_alloc_in = (on && on[0] == '1' && nth > 0) ? nth : 0;
}

#if defined(M1V_STAMPS) || defined(M1V_TILE_STAMPS) || defined(M1V_ASM_STAMPS)
// diagnostic builds only: read and clear the per-phase cycle sums
int m1v_debug_read_stamps(m1v_encoder *e, unsigned long long out[32]) {
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, e->d_stamps, 32 * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(e->d_stamps, 0, 32 * 8));
    return M1V_OK;
}
// diagnostic builds only: the per-workgroup timeline of the last k_assemble launch (8 stamps each)
int m1v_debug_read_timeline(m1v_encoder *e, unsigned long long *out, int workgroups) {
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, e->d_stamps + 32, (size_t)(workgroups > 65536 ? 65536 : workgroups) * 64, hipMemcpyDeviceToHost));
    return M1V_OK;
}
#endif

int m1v_profile_enable(m1v_encoder *e, int enable) {
    if (!e) return fail(M1V_E_ARG, "null encoder%s");
    e->prof = enable != 0;
    e->ev_used = 0;
    return M1V_OK;
}

int m1v_profile_read(m1v_encoder *e, int *launches, double *total_ms) {
    if (!e) return fail(M1V_E_ARG, "null encoder%s");
    HIP_TRY(hipSetDevice(e->device));
    double sum = 0;
    int n = 0;
    for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(e->ev[i + 1]));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]));
        sum += ms;
        n++;
    }
    e->ev_used = 0;
    if (launches) *launches = n;
    if (total_ms) *total_ms = sum;
    return M1V_OK;
}

int m1v_profile_read_times(m1v_encoder *e, float *ms, int cap, int *launches) {
    if (!e || (cap > 0 && !ms)) return fail(M1V_E_ARG, "bad argument%s");
    HIP_TRY(hipSetDevice(e->device));
    int n = 0;
    for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
        HIP_TRY(hipEventSynchronize(e->ev[i + 1]));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, e->ev[i], e->ev[i + 1]));
        if (n < cap) ms[n] = t;
        n++;
    }
    e->ev_used = 0;
    if (launches) *launches = n;
    return M1V_OK;
}

static int profile_event(m1v_encoder *e, hipStream_t st) {
    if (e->ev_used == e->ev.size()) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        e->ev.push_back(ev);
    }
    HIP_TRY(hipEventRecord(e->ev[e->ev_used++], st));
    return M1V_OK;
}

static bool fast_path(const m1v_encoder *e, const uint8_t *d_rgb) {
    return e->fast_ok && ((uintptr_t)d_rgb & 3) == 0;
}

int m1v_encode_device(m1v_encoder *e, const uint8_t *d_rgb, int n_frames, int first_frame_index,
                      uint8_t *d_out, size_t out_cap, uint64_t *d_frame_sizes, uint64_t *d_total,
                      uint32_t *d_status, void *stream) {
    if (!e || !d_rgb || !d_out) return fail(M1V_E_ARG, "null pointer%s");
    if (n_frames < 0 || n_frames > e->max_frames) return fail(M1V_E_ARG, "n_frames exceeds max_frames%s");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(e->device));
    m1v_encoder::Batch &bt = e->batch[e->pipelined ? (e->calls++ & 1u) : 0];
    hipStream_t gs = e->pipelined ? e->side : st;   // stream of the layout + gather kernels
    if (e->pipelined && bt.gather_pending)           // this set's previous gather must have drained its scratch
        HIP_TRY(hipStreamWaitEvent(st, bt.gather_done, 0));
    if (n_frames == 0) {
        if (d_total) HIP_TRY(hipMemsetAsync(d_total, 0, 8, st));
        if (d_status) HIP_TRY(hipMemsetAsync(d_status, 0, 4, st));
        return M1V_OK;
    }
    const Geometry &g = e->g;
    const bool fast = fast_path(e, d_rgb);
    if (!bt.scratch || !bt.seg || (!e->tiles && e->dense && !bt.run_meta))
        return fail(M1V_E_HIP, "the encoder has no scratch (an earlier allocation failed)%s");
    // the counters this batch adds to, and the set the next batch of this Batch will use (k_assemble clears it)
    m1v_encoder::Counters &cur = bt.ctr[bt.turn & 1u], &nxt = bt.ctr[(bt.turn + 1u) & 1u];
    bt.turn++;
    if (e->tiles) {
        TileArgs a;
        a.g = g;
        a.rgb = d_rgb;
        a.tab = e->d_tab;
        a.scratch = bt.scratch;
        a.seg = bt.seg;
        a.strip_ctr = cur.strip_ctr;
        a.frame_bytes = cur.frame_bytes;
        a.arena_next = cur.words + 2;
        a.slot_bytes = e->slot_bytes;
        a.arena_slots = e->arena_slots;
        a.arena_off = e->arena_off;
        a.status = cur.words;
        a.n_frames = n_frames;
        a.tile_cols = e->tile_cols;
        a.tile_rows = e->tile_rows;
        a.tiles_per_frame = e->tiles_per_frame;
        {
            const unsigned long long units = (unsigned long long)n_frames * (unsigned long long)e->tiles_per_frame;
            a.div_group = div_magic(8u * (uint32_t)e->tiles_per_frame, units);
            a.div_frame = div_magic((uint32_t)e->tiles_per_frame, units);
            a.div_cols = div_magic((uint32_t)e->tile_cols, (unsigned long long)e->tiles_per_frame);
        }
        a.tile_row_order = e->d_tile_order;
        a.lds_words = e->image_words;
        a.run_cap = e->run_cap;
        a.luma_region = e->luma_region;
        a.chroma_region = e->chroma_region;
        a.stamps = e->d_stamps;
        const size_t lds = (size_t)kTileFixedWords * 4 + 2 * (size_t)a.luma_region + a.chroma_region + (size_t)a.lds_words * 4;
        if (lds > 160 * 1024) return fail(M1V_E_ARG, "LDS budget exceeded%s");
        dim3 grid((unsigned)((size_t)n_frames * e->tiles_per_frame)), block((unsigned)kTileThreads);
        if (e->prof && profile_event(e, st) != M1V_OK) return M1V_E_HIP;
        if (e->narrow)
            hipLaunchKernelGGL((k_encode_tiles<true, M1V_TILE_RING>), grid, block, lds, st, a);
        else
            hipLaunchKernelGGL((k_encode_tiles<false, M1V_TILE_RING>), grid, block, lds, st, a);
        if (e->prof && profile_event(e, st) != M1V_OK) return M1V_E_HIP;
        HIP_TRY(hipGetLastError());
        if (e->pipelined) {
            HIP_TRY(hipEventRecord(bt.enc_done, st));
            HIP_TRY(hipStreamWaitEvent(gs, bt.enc_done, 0));
        }
    } else if (e->dense) {
        DenseArgs a;
        a.g = g;
        a.rgb = d_rgb;
        a.tab = e->d_tab;
        a.scratch = bt.scratch;
        a.run_meta = bt.run_meta;
        a.status = cur.words;
        a.n_frames = n_frames;
        a.threads = e->dense_T;
        a.runs_per_frame = e->runs_per_frame;
        // LDS image of the run's bits: zeroing it is ~3 % of the kernel per KiB-word, outgrowing it costs the
        // slow global-atomics path.  A run of 256 blocks needs ~230 words at quality 12 on noise; scale the
        // default with the quantiser (finer quantisers emit more bits per block).
        a.lds_words = e->image_words;
        a.run_cap = e->run_cap;
        a.slot_bytes = e->slot_bytes;
        a.arena_slots = e->arena_slots;
        a.arena_off = e->arena_off;
        a.arena_next = cur.words + 2;
        a.stamps = e->d_stamps;
        const int stride = e->narrow ? kStageStride8 : kStageStride16;
        a.zero_iters = (a.lds_words + e->dense_T - 1) / e->dense_T;
        size_t lds = (size_t)((e->dense_T / kWave) * kVlcWords + 32 + stride * e->dense_T + a.zero_iters * e->dense_T) * 4;
        if (lds > 160 * 1024) return fail(M1V_E_ARG, "LDS budget exceeded%s");
        dim3 grid((unsigned)((size_t)n_frames * e->runs_per_frame)), block((unsigned)e->dense_T);
        if (e->prof && profile_event(e, st) != M1V_OK) return M1V_E_HIP;
        // input mode of the pixel loads (see load_block_rows): 1 = aligned rows, 2 = any row offset in an aligned
        // buffer (3 channels), 0 = byte loads
        const bool aligned4 = ((uintptr_t)d_rgb & 3) == 0;
        int mode = fast ? 1 : (aligned4 && g.C == 3 ? 2 : (aligned4 && g.C == 4 ? 3 : 0));
        if (e->forced_mode == 0 || (e->forced_mode == 2 && mode == 1)) mode = e->forced_mode; // only modes that are valid here
        if (mode == 1 && e->narrow)
            hipLaunchKernelGGL((k_encode_dense<1, true>), grid, block, lds, st, a);
        else if (mode == 1)
            hipLaunchKernelGGL((k_encode_dense<1, false>), grid, block, lds, st, a);
        else if (mode == 2 && e->narrow)
            hipLaunchKernelGGL((k_encode_dense<2, true>), grid, block, lds, st, a);
        else if (mode == 2)
            hipLaunchKernelGGL((k_encode_dense<2, false>), grid, block, lds, st, a);
        else if (mode == 3 && e->narrow)
            hipLaunchKernelGGL((k_encode_dense<3, true>), grid, block, lds, st, a);
        else if (mode == 3)
            hipLaunchKernelGGL((k_encode_dense<3, false>), grid, block, lds, st, a);
        else if (e->narrow)
            hipLaunchKernelGGL((k_encode_dense<0, true>), grid, block, lds, st, a);
        else
            hipLaunchKernelGGL((k_encode_dense<0, false>), grid, block, lds, st, a);
        if (e->prof && profile_event(e, st) != M1V_OK) return M1V_E_HIP;
        HIP_TRY(hipGetLastError());
        if (e->pipelined) {
            HIP_TRY(hipEventRecord(bt.enc_done, st));
            HIP_TRY(hipStreamWaitEvent(gs, bt.enc_done, 0));
        }

        DenseGeom d;
        d.n_frames = n_frames;
        d.n_strips = g.n_strips;
        d.bps = g.n_mbrows * 6;
        d.T = e->dense_T;
        d.runs_per_frame = e->runs_per_frame;
        hipLaunchKernelGGL(k_dense_frame_layout, dim3(n_frames), dim3(256), 0, gs, d, e->segs, bt.run_meta, bt.seg, cur.strip_ctr,
                           cur.frame_bytes);
        HIP_TRY(hipGetLastError());
    } else {
        EncodeArgs a;
        a.g = g;
        a.rgb = d_rgb;
        a.tab = e->d_tab;
        a.scratch = bt.scratch;
        a.seg = bt.seg;
        a.strip_ctr = cur.strip_ctr;
        a.frame_bytes = cur.frame_bytes;
        a.status = cur.words;
        a.n_frames = n_frames;
        a.threads = e->threads;
        a.lds_words = e->lds_words > 0 ? e->lds_words : kDefaultLdsWords;
        a.stamps = e->d_stamps;
        size_t lds = (size_t)(kVlcWords + 32 + kStageStride16 * e->threads + a.lds_words) * 4;
        dim3 grid((unsigned)((size_t)n_frames * g.n_strips)), block((unsigned)e->threads);
        if (e->prof && profile_event(e, st) != M1V_OK) return M1V_E_HIP;
        if (fast)
            hipLaunchKernelGGL(k_encode_strips<true>, grid, block, lds, st, a);
        else
            hipLaunchKernelGGL(k_encode_strips<false>, grid, block, lds, st, a);
        if (e->prof && profile_event(e, st) != M1V_OK) return M1V_E_HIP;
        HIP_TRY(hipGetLastError());
        if (e->pipelined) {
            HIP_TRY(hipEventRecord(bt.enc_done, st));
            HIP_TRY(hipStreamWaitEvent(gs, bt.enc_done, 0));
        }
    }
    // ---- frame offsets, strip offsets, concatenation, headers, sizes, status: one launch (m1v_assemble.h) ----
    {
        AssembleArgs ga;
        ga.n_frames = n_frames;
        ga.n_strips = g.n_strips;
        ga.segs = e->segs;
        ga.group = e->asm_group;
        ga.lanes_log2 = e->asm_lanes_log2;
        ga.img_words = e->asm_img_words;
        ga.div_segs = div_magic((uint32_t)e->segs, (unsigned long long)e->asm_group * (unsigned long long)e->segs);
        ga.scratch = bt.scratch;
        ga.seg = bt.seg;
        ga.strip_ctr = cur.strip_ctr;
        ga.frame_bytes = cur.frame_bytes;
        ga.enc_words = cur.words;
        ga.next_strip_ctr = nxt.strip_ctr;
        ga.next_frame_bytes = nxt.frame_bytes;
        ga.next_words = nxt.words;
        if (nxt.dirty_frames > n_frames) { // a longer batch than this one used that set last: clear what this launch does not reach
            HIP_TRY(hipMemsetAsync(nxt.strip_ctr, 0, (size_t)nxt.dirty_frames * g.n_strips * 8, gs));
            HIP_TRY(hipMemsetAsync(nxt.frame_bytes, 0, (size_t)nxt.dirty_frames * 8, gs));
            nxt.dirty_frames = 0;
        }
        ga.next_frames = nxt.dirty_frames;
        nxt.dirty_frames = 0;
        cur.dirty_frames = n_frames;
        ga.tab = e->d_tab;
        ga.out = d_out;
        ga.out_cap = out_cap;
        ga.out_sizes = (unsigned long long *)d_frame_sizes;
        ga.out_total = (unsigned long long *)d_total;
        ga.out_status = d_status ? d_status : cur.words + 1;
        ga.first_index = first_frame_index;
        ga.stamps = e->d_stamps;
        const dim3 grid((unsigned)((g.n_strips + e->asm_group - 1) / e->asm_group), (unsigned)n_frames);
        const size_t lds = (size_t)(e->asm_img_words + kAsmFixedWords) * sizeof(uint32_t);
        if (e->scratch_bytes >= (1ull << 32))
            hipLaunchKernelGGL(k_assemble<true>, grid, dim3(kAsmThreads), lds, gs, ga);
        else
            hipLaunchKernelGGL(k_assemble<false>, grid, dim3(kAsmThreads), lds, gs, ga);
        HIP_TRY(hipGetLastError());
    }
    if (e->pipelined) {
        HIP_TRY(hipEventRecord(bt.gather_done, gs));
        bt.gather_pending = true;
    }
    return M1V_OK;
}

// ---- overlapped delivery to the host (include/mpeg1_hip.h) ----------------------------------------------------------
struct m1v_delivery {
    m1v_encoder *e;
    int device; // (the encoder may be destroyed before the delivery object is)
    size_t cap;
    int max_frames;
    uint8_t *d_out[2], *h_out[2];
    unsigned long long *d_meta[2], *h_meta[2];   // [0] total bytes, [1] status word (low 32 bits)
    unsigned long long *d_sizes[2], *h_sizes[2];
    hipStream_t side;
    hipEvent_t encoded[2], counted[2], delivered[2];
    bool in_flight[2];                            // delivered[b] has been recorded and not yet been waited for by an encode
    struct {
        const uint8_t *rgb;
        int n, first;
    } args[2];
    int pending;                                  // slot whose batch is encoded (or encoding) and not yet on its way, or -1
    unsigned step_no;
};

static int delivery_start(m1v_delivery *d, int b) { // the copy of slot b's batch, behind its encode
    m1v_encoder *e = d->e;
    HIP_TRY(hipStreamWaitEvent(d->side, d->encoded[b], 0));
    HIP_TRY(hipMemcpyAsync(d->h_meta[b], d->d_meta[b], 16, hipMemcpyDeviceToHost, d->side));
    HIP_TRY(hipEventRecord(d->counted[b], d->side));
    HIP_TRY(hipEventSynchronize(d->counted[b])); // the step's only host wait: the next encode is already queued
    unsigned long long total = d->h_meta[b][0];
    uint32_t status = (uint32_t)d->h_meta[b][1];
    if (status == M1V_STATUS_SCRATCH) { // recoverable: the worst case reserved (waits for the device), the same frames again
        int rc = m1v_reserve_scratch(e, 1);
        if (rc != M1V_OK) return rc;
        rc = m1v_encode_device(e, d->args[b].rgb, d->args[b].n, d->args[b].first, d->d_out[b], d->cap, (uint64_t *)d->d_sizes[b],
                               (uint64_t *)d->d_meta[b], reinterpret_cast<uint32_t *>(d->d_meta[b] + 1), d->side);
        if (rc != M1V_OK) return rc;
        if (e->pipelined) HIP_TRY(hipStreamWaitEvent(d->side, e->batch[(e->calls - 1u) & 1u].gather_done, 0));
        HIP_TRY(hipMemcpyAsync(d->h_meta[b], d->d_meta[b], 16, hipMemcpyDeviceToHost, d->side));
        HIP_TRY(hipStreamSynchronize(d->side));
        total = d->h_meta[b][0];
        status = (uint32_t)d->h_meta[b][1];
    }
    if (status & M1V_STATUS_UNENCODABLE) return fail(M1V_E_UNENCODABLE, "a level of 256 or more: the reference cannot code this batch%s");
    if (status & M1V_STATUS_NOSPACE) return fail(M1V_E_NOSPACE, "the delivery's output buffers are too small for this batch%s");
    if (status) return fail(M1V_E_SCRATCH, "the batch ran out of scratch twice%s");
    if (total > d->cap) return fail(M1V_E_NOSPACE, "the delivery's output buffers are too small for this batch%s");
    HIP_TRY(hipMemcpyAsync(d->h_out[b], d->d_out[b], total, hipMemcpyDeviceToHost, d->side));
    HIP_TRY(hipMemcpyAsync(d->h_sizes[b], d->d_sizes[b], (size_t)d->args[b].n * 8, hipMemcpyDeviceToHost, d->side));
    HIP_TRY(hipEventRecord(d->delivered[b], d->side));
    d->in_flight[b] = true;
    return b;
}

int m1v_delivery_create(m1v_encoder *e, size_t out_cap, m1v_delivery **out) {
    if (!e || !out) return fail(M1V_E_ARG, "null pointer%s");
    *out = nullptr;
    HIP_TRY(hipSetDevice(e->device));
    m1v_delivery *d = new m1v_delivery();
    memset(d, 0, sizeof *d);
    d->e = e;
    d->device = e->device;
    d->max_frames = e->max_frames;
    d->cap = out_cap ? out_cap : (size_t)e->max_frames * m1v_frame_bound(e);
    d->pending = -1;
    hipError_t err = hipStreamCreateWithFlags(&d->side, hipStreamNonBlocking);
    for (int b = 0; b < 2; b++) {
        if (err == hipSuccess) err = hipMalloc(&d->d_out[b], d->cap);
        if (err == hipSuccess) err = hipMalloc(&d->d_meta[b], 16);
        if (err == hipSuccess) err = hipMemset(d->d_meta[b], 0, 16);
        if (err == hipSuccess) err = hipMalloc(&d->d_sizes[b], (size_t)e->max_frames * 8);
        if (err == hipSuccess) err = hipHostMalloc(&d->h_out[b], d->cap, hipHostMallocDefault);
        if (err == hipSuccess) err = hipHostMalloc(&d->h_meta[b], 16, hipHostMallocDefault);
        if (err == hipSuccess) err = hipHostMalloc(&d->h_sizes[b], (size_t)e->max_frames * 8, hipHostMallocDefault);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&d->encoded[b], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&d->counted[b], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&d->delivered[b], hipEventDisableTiming);
    }
    if (err != hipSuccess) {
        fail(M1V_E_HIP, "allocation failed: %s", hipGetErrorString(err));
        m1v_delivery_destroy(d);
        return M1V_E_HIP;
    }
    *out = d;
    return M1V_OK;
}

void m1v_delivery_destroy(m1v_delivery *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->side) (void)hipStreamSynchronize(d->side);
    for (int b = 0; b < 2; b++) {
        (void)hipFree(d->d_out[b]);
        (void)hipFree(d->d_meta[b]);
        (void)hipFree(d->d_sizes[b]);
        (void)hipHostFree(d->h_out[b]);
        (void)hipHostFree(d->h_meta[b]);
        (void)hipHostFree(d->h_sizes[b]);
        if (d->encoded[b]) (void)hipEventDestroy(d->encoded[b]);
        if (d->counted[b]) (void)hipEventDestroy(d->counted[b]);
        if (d->delivered[b]) (void)hipEventDestroy(d->delivered[b]);
    }
    if (d->side) (void)hipStreamDestroy(d->side);
    delete d;
}

int m1v_delivery_step(m1v_delivery *d, const uint8_t *d_rgb, int n_frames, int first_frame_index, void *stream) {
    if (!d || !d_rgb) return fail(M1V_E_ARG, "null pointer%s");
    if (n_frames <= 0 || n_frames > d->max_frames) return fail(M1V_E_ARG, "n_frames exceeds max_frames%s");
    m1v_encoder *e = d->e;
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(e->device));
    const int b = (int)(d->step_no++ & 1u);
    if (d->in_flight[b]) { // slot b's previous batch has left for the host before its buffers are written again
        HIP_TRY(hipStreamWaitEvent(st, d->delivered[b], 0));
        d->in_flight[b] = false;
    }
    const int rc = m1v_encode_device(e, d_rgb, n_frames, first_frame_index, d->d_out[b], d->cap, (uint64_t *)d->d_sizes[b],
                                     (uint64_t *)d->d_meta[b], reinterpret_cast<uint32_t *>(d->d_meta[b] + 1), st);
    if (rc != M1V_OK) return rc;
    if (e->pipelined) HIP_TRY(m1v_flush(e, st) == M1V_OK ? hipSuccess : hipErrorUnknown);
    HIP_TRY(hipEventRecord(d->encoded[b], st));
    d->args[b].rgb = d_rgb;
    d->args[b].n = n_frames;
    d->args[b].first = first_frame_index;
    const int before = d->pending;
    d->pending = b;
    return before >= 0 ? delivery_start(d, before) : (int)M1V_DELIVERY_NONE;
}

int m1v_delivery_flush(m1v_delivery *d) {
    if (!d) return fail(M1V_E_ARG, "null pointer%s");
    HIP_TRY(hipSetDevice(d->e->device));
    const int before = d->pending;
    d->pending = -1;
    return before >= 0 ? delivery_start(d, before) : (int)M1V_DELIVERY_NONE;
}

uint64_t m1v_delivery_bytes(const m1v_delivery *d, int slot) { return d && slot >= 0 && slot <= 1 ? d->h_meta[slot][0] : 0; }

int m1v_delivery_wait(m1v_delivery *d, int slot, const uint8_t **host, uint64_t *bytes, const uint64_t **frame_sizes) {
    if (!d || slot < 0 || slot > 1) return fail(M1V_E_ARG, "bad slot%s");
    HIP_TRY(hipSetDevice(d->device));
    HIP_TRY(hipEventSynchronize(d->delivered[slot]));
    if (host) *host = d->h_out[slot];
    if (bytes) *bytes = d->h_meta[slot][0];
    if (frame_sizes) *frame_sizes = (const uint64_t *)d->h_sizes[slot];
    return M1V_OK;
}

/* Pinned host memory for callers of the host-buffer entry points: H2D/D2H copies from it run at the PCIe
 * rate instead of being staged by the runtime. */
void *m1v_alloc_host(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void *m1v_alloc_device(size_t bytes) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void m1v_free_device(void *p) { (void)hipFree(p); }
void m1v_free_host(void *p) {
    if (p) (void)hipHostFree(p);
}

long m1v_encode_planes_host(m1v_encoder *e, const uint8_t *rgb, int n_frames, int first_frame_index,
                            uint8_t *out, size_t out_cap, uint64_t *frame_sizes, uint8_t *planes) {
    if (!e || !rgb || !out) return fail(M1V_E_ARG, "null pointer%s");
    if (n_frames < 0 || n_frames > e->max_frames) return fail(M1V_E_ARG, "n_frames exceeds max_frames%s");
    if (n_frames == 0) return 0;
    HIP_TRY(hipSetDevice(e->device));
    const size_t frame_in = (size_t)e->g.frame_bytes, frame_planes = (size_t)e->g.W * e->g.H * 3;
    size_t in_bytes = frame_in * n_frames;
    size_t bound = m1v_frame_bound(e) * (size_t)n_frames;
    size_t dcap = out_cap < bound ? out_cap : bound;
    m1v_encoder::HostPath &hp = e->hp;
    HIP_TRY(ensure_device(&hp.d_in, &hp.in_cap, in_bytes));
    HIP_TRY(ensure_device(&hp.d_out, &hp.out_cap, dcap));
    HIP_TRY(ensure_device(&hp.d_meta, &hp.meta_cap, (size_t)(n_frames + 2) * 8)); // [n] sizes, total, status
    if (planes) HIP_TRY(ensure_device(&hp.d_planes, &hp.planes_cap, frame_planes * n_frames));
    if (!hp.copy_in) { // all or nothing: a half-built set must not survive into the next call
        hipError_t err = hipStreamCreateWithFlags(&hp.copy_in, hipStreamNonBlocking);
        if (err == hipSuccess) err = hipStreamCreateWithFlags(&hp.work, hipStreamNonBlocking);
        for (hipEvent_t &ev : hp.uploaded)
            if (err == hipSuccess) err = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (err != hipSuccess) {
            if (hp.copy_in) (void)hipStreamDestroy(hp.copy_in);
            if (hp.work) (void)hipStreamDestroy(hp.work);
            for (hipEvent_t &ev : hp.uploaded) {
                if (ev) (void)hipEventDestroy(ev);
                ev = nullptr;
            }
            hp.copy_in = hp.work = nullptr;
            return fail(M1V_E_HIP, "stream creation failed: %s", hipGetErrorString(err));
        }
    }
    // From here on copies to and from the caller's buffers are in flight: every error return first waits for both
    // streams, so that the caller may free (or reuse) rgb / planes / out as soon as this function has returned.
    auto drained = [&](int rc) {
        (void)hipStreamSynchronize(hp.copy_in);
        (void)hipStreamSynchronize(hp.work);
        return rc;
    };
#define HIP_TRY_DRAIN(expr)                                                                        \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return drained(fail(M1V_E_HIP, #expr ": %s", hipGetErrorString(e_))); \
    } while (0)
    // copy_in: H2D half A, H2D half B.   work: [planes A -> host] while B uploads, [planes B -> host], encode all.
    const int half[3] = {0, planes && n_frames > 1 ? n_frames / 2 : n_frames, n_frames};
    for (int h = 0; h < 2; h++) {
        int f0 = half[h], nf = half[h + 1] - half[h];
        if (nf == 0) continue;
        HIP_TRY_DRAIN(hipMemcpyAsync(hp.d_in + frame_in * f0, rgb + frame_in * f0, frame_in * nf, hipMemcpyHostToDevice, hp.copy_in));
        HIP_TRY_DRAIN(hipEventRecord(hp.uploaded[h], hp.copy_in));
        HIP_TRY_DRAIN(hipStreamWaitEvent(hp.work, hp.uploaded[h], 0));
        if (planes) {
            int rc = m1v_convert_device(e, hp.d_in + frame_in * f0, nf, hp.d_planes + frame_planes * f0, hp.work);
            if (rc != M1V_OK) return drained(rc);
            HIP_TRY_DRAIN(hipMemcpyAsync(planes + frame_planes * f0, hp.d_planes + frame_planes * f0, frame_planes * nf,
                                   hipMemcpyDeviceToHost, hp.work));
        }
    }
    std::vector<unsigned long long> meta((size_t)n_frames + 2);
    uint32_t status = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        int r = m1v_encode_device(e, hp.d_in, n_frames, first_frame_index, hp.d_out, dcap, (uint64_t *)hp.d_meta,
                                  (uint64_t *)(hp.d_meta + n_frames), (uint32_t *)(hp.d_meta + n_frames + 1), hp.work);
        if (r != M1V_OK) return drained(r);
        if (m1v_flush(e, hp.work) != M1V_OK) return drained(M1V_E_HIP);
        HIP_TRY_DRAIN(hipMemcpyAsync(meta.data(), hp.d_meta, meta.size() * 8, hipMemcpyDeviceToHost, hp.work));
        HIP_TRY_DRAIN(hipStreamSynchronize(hp.work));
        status = (uint32_t)meta[(size_t)n_frames + 1];
        if (!(status & M1V_STATUS_SCRATCH) || attempt == 1) break;
        // more runs outgrew their compact scratch slot than the overflow arena holds: reserve the worst case, encode again
        int rr = m1v_reserve_scratch(e, 1);
        if (rr != M1V_OK) return drained(rr);
    }
    if (status & M1V_STATUS_SCRATCH) return drained(fail(M1V_E_SCRATCH, "scratch exhausted%s"));
    unsigned long long total = meta[n_frames];
    if (status & M1V_STATUS_UNENCODABLE)
        return fail(M1V_E_UNENCODABLE, "an AC level has |level| >= 256 (the reference crashes here)%s");
    if ((status & M1V_STATUS_NOSPACE) || total > out_cap) return fail(M1V_E_NOSPACE, "output buffer too small%s");
    HIP_TRY_DRAIN(hipMemcpy(out, hp.d_out, total, hipMemcpyDeviceToHost));
    if (frame_sizes)
        for (int f = 0; f < n_frames; f++) frame_sizes[f] = meta[f];
    return (long)total;
#undef HIP_TRY_DRAIN
}

long m1v_encode_host(m1v_encoder *e, const uint8_t *rgb, int n_frames, int first_frame_index,
                     uint8_t *out, size_t out_cap, uint64_t *frame_sizes) {
    return m1v_encode_planes_host(e, rgb, n_frames, first_frame_index, out, out_cap, frame_sizes, nullptr);
}

int m1v_coefficients_device(m1v_encoder *e, const uint8_t *d_rgb, int n_frames, int16_t *d_coeffs,
                            void *stream) {
    if (!e || !d_rgb || !d_coeffs || n_frames < 0) return fail(M1V_E_ARG, "bad argument%s");
    if (n_frames == 0) return M1V_OK;
    HIP_TRY(hipSetDevice(e->device));
    CoefArgs a;
    a.g = e->g;
    a.rgb = d_rgb;
    a.tab = e->d_tab;
    a.out = d_coeffs;
    a.n_frames = n_frames;
    if (e->g.C == 3 && e->forced_mode < 0 && e->forced_path != 0) { // tiles (any width, any alignment); the run-shaped kernel serves 4 channels
        CoefTileArgs t;
        t.g = e->g;
        t.rgb = d_rgb;
        t.tab = e->d_tab;
        t.out = d_coeffs;
        t.n_frames = n_frames;
        t.tile_cols = (e->g.n_strips + kTileStrips - 1) / kTileStrips;
        t.tile_rows = (e->g.n_mbrows + kTileMbRows - 1) / kTileMbRows;
        t.tiles_per_frame = t.tile_cols * t.tile_rows;
        t.region = (std::max<uint32_t>((uint32_t)M1V_TILE_RING * kTileSlot, (uint32_t)(kWave * kCoefStride * 4)) + 15u) & ~15u;
        hipLaunchKernelGGL((k_coefficient_tiles<M1V_TILE_RING>), dim3((unsigned)((size_t)n_frames * t.tiles_per_frame)),
                           dim3(kTileThreads), 3 * (size_t)t.region, (hipStream_t)stream, t);
        HIP_TRY(hipGetLastError());
        return M1V_OK;
    }
    int bps = e->g.n_mbrows * 6;
    dim3 grid((bps + 255) / 256, e->g.n_strips, n_frames);
    if (fast_path(e, d_rgb))
        hipLaunchKernelGGL(k_coefficients<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_coefficients<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return M1V_OK;
}

int m1v_convert_device(m1v_encoder *e, const uint8_t *d_rgb, int n_frames, uint8_t *d_planes,
                       void *stream) {
    if (!e || !d_rgb || !d_planes || n_frames < 0) return fail(M1V_E_ARG, "bad argument%s");
    if (n_frames == 0) return M1V_OK;
    HIP_TRY(hipSetDevice(e->device));
    unsigned long long npx = (unsigned long long)e->g.W * e->g.H;
    unsigned long long total = npx * n_frames;
    if (npx % 4 == 0 && (((uintptr_t)d_rgb | (uintptr_t)d_planes) & 3) == 0) { // four pixels per lane, dword loads and stores
        total /= 4;
        unsigned blocks = (unsigned)((total + 255) / 256 > 131072 ? 131072 : (total + 255) / 256);
        if (e->g.C == 3)
            hipLaunchKernelGGL(k_convert4<3>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)d_rgb,
                               npx / 4, n_frames, (uint32_t *)d_planes);
        else
            hipLaunchKernelGGL(k_convert4<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)d_rgb,
                               npx / 4, n_frames, (uint32_t *)d_planes);
        HIP_TRY(hipGetLastError());
        return M1V_OK;
    }
    unsigned blocks = (unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(k_convert, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_rgb, e->g.C, npx,
                       n_frames, d_planes);
    HIP_TRY(hipGetLastError());
    return M1V_OK;
}

int m1v_convert_host(m1v_encoder *e, const uint8_t *rgb, int n_frames, uint8_t *planes) {
    if (!e || !rgb || !planes || n_frames < 0) return fail(M1V_E_ARG, "bad argument%s");
    if (n_frames == 0) return M1V_OK;
    HIP_TRY(hipSetDevice(e->device));
    size_t in_bytes = (size_t)e->g.frame_bytes * n_frames;
    size_t out_bytes = (size_t)e->g.W * e->g.H * 3 * n_frames;
    m1v_encoder::HostPath &hp = e->hp;
    HIP_TRY(ensure_device(&hp.d_in, &hp.in_cap, in_bytes));
    HIP_TRY(ensure_device(&hp.d_planes, &hp.planes_cap, out_bytes));
    HIP_TRY(hipMemcpy(hp.d_in, rgb, in_bytes, hipMemcpyHostToDevice));
    int rc = m1v_convert_device(e, hp.d_in, n_frames, hp.d_planes, nullptr);
    if (rc != M1V_OK) return rc;
    HIP_TRY(hipMemcpy(planes, hp.d_planes, out_bytes, hipMemcpyDeviceToHost));
    return M1V_OK;
}

int m1v_subsample_device(m1v_encoder *e, const uint8_t *d_cb, const uint8_t *d_cr, uint8_t *d_cb_sub,
                         uint8_t *d_cr_sub, void *stream) {
    if (!e || !d_cb || !d_cr || !d_cb_sub || !d_cr_sub) return fail(M1V_E_ARG, "bad argument%s");
    if ((e->g.W | e->g.H) & 1) return fail(M1V_E_ARG, "odd dimensions: the reference reads out of bounds%s");
    HIP_TRY(hipSetDevice(e->device));
    int n = (e->g.W / 2) * (e->g.H / 2);
    hipLaunchKernelGGL(k_subsample, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_cb, d_cr,
                       e->g.W, e->g.H, d_cb_sub, d_cr_sub);
    HIP_TRY(hipGetLastError());
    return M1V_OK;
}

int m1v_synth_device(uint8_t *d_rgb, size_t bytes_per_frame, int n_frames, uint64_t seed,
                     uint64_t first_frame_index, void *stream) {
    if (!d_rgb || n_frames < 0) return fail(M1V_E_ARG, "bad argument%s");
    if (n_frames == 0 || bytes_per_frame == 0) return M1V_OK;
    unsigned long long total = ((bytes_per_frame + 7) / 8) * (unsigned long long)n_frames;
    unsigned blocks = (unsigned)((total + 255) / 256 > 262144 ? 262144 : (total + 255) / 256);
    hipLaunchKernelGGL(k_synth, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_rgb,
                       (unsigned long long)bytes_per_frame, n_frames, (unsigned long long)seed,
                       (unsigned long long)first_frame_index);
    HIP_TRY(hipGetLastError());
    return M1V_OK;
}

} // extern "C"
