// ec504_imageencoder_amd/csrc/m1v_assemble.h — the ONE kernel behind an encode kernel (gfx950).  Not a standalone header: it is
// included by m1v_kernels.hip inside its anonymous namespace.
//
// What it replaces in the reference: the sequential tail of the frame loop — strip zero-padding (encoder.h:442-443),
// bitvector_fwrite of the picture's bits (:445), the packet-length back-patch (:448-454), the four trailer bytes
// (:456-458) and the ftell arithmetic that places frame i behind frame i-1 (:198) — and in this library, up to round 3, three
// launches (strip layout, frame offsets, a gather of one wave per strip at 0.195 of the HBM rate).
//
// Input, all produced by the encode kernel of the same batch (any of the three):
//   seg[frame][q][strip] = (bits, where)   what a strip is concatenated from: `bits` bits that start at word `where` of scratch
//                                          (segment-major: the eight entries a tile writes are 64 contiguous bytes)
//   strip_ctr[frame][strip]                low 40 bits: bits of the strip (the tile kernel adds them up with one returning atomic
//                                          per segment; its upper bits count the arrivals)
//   frame_bytes[frame]                     sum over the frame's strips of ceil(bits / 8): added by whoever completed a strip
// so that NO value flows from workgroup to workgroup here: every workgroup derives its own place from those arrays.
//
// Mapping.  One workgroup (256 lanes = 4 waves) per (frame, GROUP of consecutive strips).
//   1. Prologue, one job per wave, all at once: wave 0 sums frame_bytes[0 .. frame) (+ 48 per frame: PKT SEQ GOP PIC + trailer) =
//      where the frame starts; wave 1 sums the bytes of the frame's strips in front of the group = where the group starts, and
//      the group's own bytes; wave 2 scans the bit counts of the group's segments (a strip's zero padding rides on its last
//      segment) = every segment's destination bit inside the group's bytes; all waves clear the LDS image.  One barrier.
//   2. SOURCE words scatter into the image, which is laid out against the 16-byte grid of the OUTPUT address: 2^k lanes per
//      segment, four consecutive source words per lane (one 16-byte load; the word in front of them comes from the neighbouring
//      lane by DPP), one funnel shift (v_alignbit_b32) per destination word and one ds_or_b32 (neighbouring segments share
//      their boundary word).  All loads of a trip are issued before the first is used.  One barrier.
//   3. The image leaves as aligned 16-byte stores; only the group's first and last partial units are written byte by byte
//      (their other bytes belong to the neighbouring groups, the frame header or the trailer).
// A group whose bytes outgrow the image takes several passes over its segments, a group of more than 256 segments several
// chunks (any picture, any quality: correct; the host sizes groups so that noise at the encoder's quality needs one of each).
//
// What bounds it (tools/asm_stamps.py: a timeline of every workgroup; tools/pmc_asm.sh): INSTRUCTION ISSUE.  The first forms of
// this kernel spent 1,300-2,500 instructions per wave — every wave repeating the reductions and the scan, 64-bit vector
// addressing, one lane per source word — and ran 37-46 us for 300 x 1080p at any occupancy (23 M wave instructions over 1,024
// SIMDs).  Hence the one-job-per-wave prologue, scalar bases with 32-bit offsets, and four words per lane.
constexpr unsigned long long kCtrBitsMask = (1ull << 40) - 1; // strip_ctr: bits of the strip; above: tile rows that have arrived
constexpr int kCtrCountShift = 40;
#ifndef M1V_ASM_THREADS
#define M1V_ASM_THREADS 256
#endif
constexpr int kAsmThreads = M1V_ASM_THREADS; // 256, 128 or 64: the prologue's four jobs are dealt to the waves there are
constexpr int kAsmWaves = kAsmThreads / 64;
#ifndef M1V_ASM_CHUNK
#define M1V_ASM_CHUNK 256
#endif
constexpr int kAsmChunk = M1V_ASM_CHUNK;     // segments placed at a time (one wave scans them in blocks of 64)
constexpr int kAsmScanBlocks = kAsmChunk / 64;
constexpr int kAsmMaxGroup = 16;
#ifndef M1V_ASM_IMAGE_BYTES
#define M1V_ASM_IMAGE_BYTES 14336
#endif
constexpr int kAsmImageBytes = M1V_ASM_IMAGE_BYTES; // LDS image of a group's output bytes: with the placement table 18.3 KB = eight workgroups per CU
// LDS words behind the image: [0..1] frame offset (u64), [2] bytes in front of the group, [3] bytes of the frame's strips,
// [4] bytes of the group; from word 16: placement of <= 256 segments (16 bytes each)
constexpr int kAsmPlace = 16, kAsmFixedWords = kAsmPlace + 4 * kAsmChunk;

struct AssembleArgs {
    int n_frames, n_strips, segs; // segs = segments per strip
    int group;                    // strips per workgroup (<= kAsmMaxGroup)
    int lanes_log2;               // 2^k lanes (of four words each) share a segment, k <= 4
    int img_words;                // LDS image capacity in words (multiple of 4)
    DivMagic div_segs;            // division of a segment index by segs
    const uint8_t *scratch;
    const uint2 *seg;
    const unsigned long long *strip_ctr, *frame_bytes;
    const uint32_t *enc_words;    // [0] status bits of the encode kernel
    // the counters the NEXT batch's encode kernel adds to: cleared here for frames < next_frames (the host clears the rest)
    unsigned long long *next_strip_ctr, *next_frame_bytes;
    uint32_t *next_words;         // [0] status, [2] arena counter
    int next_frames;
    const Tables *tab;
    uint8_t *out;
    unsigned long long out_cap;
    unsigned long long *out_sizes, *out_total; // may be null
    uint32_t *out_status;
    int first_index;
    unsigned long long *stamps; // diagnostic builds only (-DM1V_ASM_STAMPS, tools/asm_stamps.py)
};

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_inclusive(v), kWave - 1); }
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}

// the top min(max(nb, 0), 32) bits set
__device__ __forceinline__ uint32_t asm_top_bits(int nb) { return (uint32_t)(0xffffffff00000000ull >> (uint32_t)min(max(nb, 0), 32)); }

struct __attribute__((aligned(4))) AsmWords4 {
    uint32_t w[4];
};

#ifndef M1V_ASM_U
#define M1V_ASM_U 5 // segments per lane and trip: 8 lanes per segment x 5 trips of 32 = 160 segments in flight (1080p: 8 strips x 17, 4K: 4 x 34)
#endif

// Diagnostic build only: cycles thread 0 of every workgroup spends in each phase, added to AssembleArgs::stamps at the end.
#ifdef M1V_ASM_STAMPS
#define ASTAMP(ph)                                                                                 \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        astamp_[ph] = __builtin_amdgcn_s_memrealtime();                                            \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define ASTAMP_INIT()                                                                              \
    unsigned long long astamp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                      \
    const unsigned long long astamp_t_ = __builtin_amdgcn_s_memrealtime();                         \
    asm volatile("s_waitcnt lgkmcnt(0)")
#define ASTAMP_FLUSH()                                                                             \
    do {                                                                                           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
        ASTAMP(6);                                                                                 \
        if (threadIdx.x == 0) {                                                                    \
            unsigned long long *tl_ = a.stamps + 32 + 8 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x); \
            tl_[0] = astamp_t_;                                                                    \
            for (int ph_ = 0; ph_ < 7; ph_++) tl_[1 + ph_] = astamp_[ph_];                         \
        }                                                                                          \
    } while (0)
#else
#define ASTAMP(ph) do { } while (0)
#define ASTAMP_INIT() do { } while (0)
#define ASTAMP_FLUSH() do { } while (0)
#endif

// WIDE: the scratch is 4 GiB or more (byte offsets of source words need 64 bits)
template <bool WIDE>
__global__ __launch_bounds__(kAsmThreads) void k_assemble(AssembleArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t asm_lds[];
    ASTAMP_INIT();
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y, s0 = (int)blockIdx.x * a.group, ns = min(a.group, a.n_strips - s0), T = a.segs;
    uint32_t *img = asm_lds, *fixed = asm_lds + a.img_words;
    uint4 *place = reinterpret_cast<uint4 *>(fixed + kAsmPlace); // per segment: destination bit (from the group's first byte), bits, source word, -
    const unsigned long long *ctr_f = a.strip_ctr + (size_t)f * a.n_strips;
    const int nseg = ns * T;
    const uint2 *seg_f = a.seg + (size_t)f * T * a.n_strips + s0; // segment q = j * T + t of the group: seg_f[t * n_strips + j]
    const uint32_t cap_bytes = (uint32_t)a.img_words * 4u;

    // Destination bits of segments [c0, c0 + 256) -> place[]: one wave, four blocks of 64, loads first.  A strip's padding (zero
    // bits up to a byte, encoder.h:442-443) rides on its last segment, so the scan runs through strip boundaries.
    auto scan_chunk = [&](int c0, uint32_t &carry) {
        uint2 sg[kAsmScanBlocks];
        unsigned long long ctr[kAsmScanBlocks];
        bool last[kAsmScanBlocks];
        const int blocks = min(kAsmScanBlocks, (nseg - c0 + kWave - 1) / kWave); // uniform
#pragma unroll
        for (int k = 0; k < kAsmScanBlocks; k++) {
            if (k >= blocks) break;
            const int q = min(c0 + 64 * k + lane, nseg - 1), j = (int)udiv((uint32_t)q, a.div_segs), t = q - j * T;
            last[k] = t == T - 1;
            sg[k] = seg_f[(size_t)t * a.n_strips + j];
            ctr[k] = ctr_f[s0 + j];
        }
#pragma unroll
        for (int k = 0; k < kAsmScanBlocks; k++) {
            if (k >= blocks) break;
            const int q = c0 + 64 * k + lane;
            const uint32_t bits = q < nseg ? sg[k].x : 0u;
            const uint32_t v = bits + (q < nseg && last[k] ? (0u - (uint32_t)ctr[k]) & 7u : 0u);
            const uint32_t incl = wave_scan_inclusive(v);
            place[64 * k + lane] = make_uint4(carry + incl - v, bits, sg[k].y, 0u);
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
        }
    };
    auto clear_image = [&](uint32_t bytes) {
        uint4 *img4 = reinterpret_cast<uint4 *>(img);
        for (uint32_t k = tid; k < ((bytes + 15u) >> 4); k += kAsmThreads) img4[k] = make_uint4(0u, 0u, 0u, 0u);
    };

    const int lg = a.lanes_log2, L = 1 << lg, sub = tid & (L - 1), slot = tid >> lg, per_trip = kAsmThreads >> lg, w0 = 4 * sub;
    constexpr int U = M1V_ASM_U;
    // four source words from word x of the scratch: a scalar base + a 32-bit byte offset unless the scratch is 4 GiB or more
    auto src_words = [&](uint32_t x) -> AsmWords4 {
        if (WIDE) return *reinterpret_cast<const AsmWords4 *>(a.scratch + ((size_t)x << 2));
        return *reinterpret_cast<const AsmWords4 *>(a.scratch + (x << 2));
    };

    // ---- 1. prologue: one job per wave (every load unconditional: clamped index, masked afterwards).  (Requesting the first
    //      trip's source words here, in front of the barrier — their addresses need the segment table only, not the scan — was
    //      tried: the workgroup's chain stays segment entry -> source word -> image -> store, 7.8 us against 7.6.) ----
    uint32_t carry = 0; // wave 2: destination bit of the next chunk's first segment
    if (wave == 0 % kAsmWaves) {
        unsigned long long acc = 0; // where the frame starts: 48 bytes of headers and trailer + its strips, for every frame in front
        for (int i0 = 0; i0 < f; i0 += 8 * kWave) { // eight loads in flight per lane: one trip up to 512 frames
            unsigned long long v[8];
#pragma unroll
            for (int k = 0; k < 8; k++) v[k] = a.frame_bytes[min(i0 + kWave * k + lane, a.n_frames - 1)];
#pragma unroll
            for (int k = 0; k < 8; k++) acc += i0 + kWave * k + lane < f ? 48ull + v[k] : 0ull;
        }
        // (a lane's sum stays far below 2^56: two 32-bit reductions)
        const uint32_t lo = wave_sum_u32((uint32_t)acc & 0xffffffu), hi = wave_sum_u32((uint32_t)(acc >> 24));
        if (lane == 0) *reinterpret_cast<unsigned long long *>(fixed) = ((unsigned long long)hi << 24) + lo;
    }
    if (wave == 1 % kAsmWaves) {
        uint32_t before = 0, payload = 0; // bytes of the frame's strips: below 4 GiB (m1v_create bounds the frame)
        for (int t0 = 0; t0 < a.n_strips; t0 += 2 * kWave) {
            const unsigned long long c0 = ctr_f[min(t0 + lane, a.n_strips - 1)], c1 = ctr_f[min(t0 + kWave + lane, a.n_strips - 1)];
            const uint32_t b0 = t0 + lane < a.n_strips ? (uint32_t)(((c0 & kCtrBitsMask) + 7ull) >> 3) : 0u;
            const uint32_t b1 = t0 + kWave + lane < a.n_strips ? (uint32_t)(((c1 & kCtrBitsMask) + 7ull) >> 3) : 0u;
            payload += b0 + b1;
            before += (t0 + lane < s0 ? b0 : 0u) + (t0 + kWave + lane < s0 ? b1 : 0u);
        }
        const unsigned long long own = ctr_f[s0 + min(lane, ns - 1)];
        const uint32_t own_bytes = lane < ns ? (uint32_t)(((own & kCtrBitsMask) + 7ull) >> 3) : 0u;
        const uint32_t group_incl = row_scan_inclusive(own_bytes); // lanes 0..15 hold the group (<= kAsmMaxGroup strips)
        before = wave_sum_u32(before);
        payload = wave_sum_u32(payload);
        if (lane == 0) {
            fixed[2] = before;
            fixed[3] = payload;
        }
        if (lane == 15) fixed[4] = group_incl;
        if (f < a.next_frames && lane < ns) a.next_strip_ctr[(size_t)f * a.n_strips + s0 + lane] = 0ull;
    }
    if (wave == 2 % kAsmWaves) {
        scan_chunk(0, carry);
    }
    clear_image(cap_bytes);
    __syncthreads();
    const unsigned long long fo = uniform_u64(*reinterpret_cast<const unsigned long long *>(fixed));
    const uint32_t B0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)fixed[2]);
    const unsigned long long fs = 48ull + (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)fixed[3]);
    const uint32_t group_bytes = (uint32_t)__builtin_amdgcn_readfirstlane((int)fixed[4]);
    const bool fits = fo + fs <= a.out_cap;
    ASTAMP(0);

    // ---- housekeeping (wave 3 of the frame's first group): header, trailer, sizes, total, status; the next batch's counters ----
    if (blockIdx.x == 0 && wave == 3 % kAsmWaves) {
        if (fits) frame_header_and_trailer(a.tab, a.out, fo, fs, a.first_index + f, lane);
        if (lane == 0) {
            if (f < a.next_frames) a.next_frame_bytes[f] = 0ull;
            if (a.out_sizes) a.out_sizes[f] = fs;
            if (f == a.n_frames - 1) {
                // frames are laid out one behind the other: the batch fits iff its last frame does
                if (a.out_total) *a.out_total = fo + fs;
                *a.out_status = a.enc_words[0] | (fits ? 0u : (uint32_t)M1V_STATUS_NOSPACE);
                a.next_words[0] = 0u;
                a.next_words[2] = 0u;
            }
        }
    }
    if (!fits) return;
    ASTAMP(1);

    const uintptr_t A0 = reinterpret_cast<uintptr_t>(a.out) + fo + 44ull + B0, a_lo = A0 & ~(uintptr_t)15;
    const uint32_t lead = (uint32_t)(A0 - a_lo), img_end = lead + group_bytes; // the group's bytes are image bytes [lead, img_end)
    // Destination words w .. w + 3 of a segment of `nw` source words from its words w - 1 (`p`: swapped, zero if there is none),
    // w .. w + 3: each is the funnel shift of two neighbours by the segment's phase.  Words past the segment's end count as
    // zero, so that destination word `nw` comes out as the tail the last source word leaves; the padding bits of a segment's
    // last word ARE zero (every encode kernel builds its segments from word boundaries of a cleared image).  CHECK: the pass
    // does not hold every word the lanes may touch.  Returns the lane's last source word (swapped, masked).
    auto scatter4 = [&](auto check, const AsmWords4 &cur, uint32_t p, int nw, int w, uint32_t sh, int k0, int capw) -> uint32_t {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t c = w + k < nw ? __builtin_bswap32(cur.w[k]) : 0u;
            const uint32_t o = __builtin_amdgcn_alignbit(p, c, sh);
            if (!decltype(check)::value)
                atomicOr(&img[k0 + k], o);
            else if (o != 0u && (unsigned)(k0 + k) < (unsigned)capw)
                atomicOr(&img[k0 + k], o);
            p = c;
        }
        return p;
    };
    // a lane reaches at most 4 * L + 3 words behind the start of its segment's destination: one pass without range checks
    // if that still lies inside the image
    const bool unchecked = img_end + 4u * (4u * (uint32_t)L + 4u) <= cap_bytes;

    // ---- 2. four destination words of U segments per lane and trip (e0 = the trip's first segment in the chunk) ----
    // A segment's placement, as the lane needs it: source words; phase; image word of the lane's first destination word; where
    // its source starts.  Read from LDS in front of the loads and again behind them rather than carried in registers across them.
    auto placement = [&](int e0, int cnt, int word0, int u, int &nw, uint32_t &sh, int &k0, uint32_t &from) {
        const int e = e0 + u * per_trip + slot;
        const uint4 pl = place[min(e, cnt - 1)];
        const uint32_t d = pl.x + 8u * lead;
        nw = e < cnt ? (int)((pl.y + 31u) >> 5) : 0;
        sh = d & 31u;
        k0 = (int)(d >> 5) - word0 + w0;
        from = pl.z;
    };
    auto load_trip = [&](AsmWords4 (&cur)[U], int e0, int cnt, int word0) { // all loads of the trip, nothing waited for
#pragma unroll
        for (int u = 0; u < U; u++) {
            int nw, k0;
            uint32_t sh, from;
            placement(e0, cnt, word0, u, nw, sh, k0, from);
            // lanes behind the segment's end read its first words (an empty segment's: the first words of the scratch)
            cur[u] = src_words(from + (w0 < nw ? (uint32_t)w0 : 0u));
        }
    };
    auto place_trip = [&](const AsmWords4 (&cur)[U], int e0, int cnt, int word0, int capw) {
        bool more = false;
#pragma unroll
        for (int u = 0; u < U; u++) {
            int nw, k0;
            uint32_t sh, from;
            placement(e0, cnt, word0, u, nw, sh, k0, from);
            more |= nw >= 4 * L;
            // the word in front of the lane's four: the neighbouring lane's last, swapped and masked there (row_shr:1)
            const uint32_t last = w0 + 3 < nw ? __builtin_bswap32(cur[u].w[3]) : 0u;
            const uint32_t p = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)last, 0x111, 0xf, 0xf, true);
            if (unchecked)
                scatter4(std::false_type(), cur[u], sub ? p : 0u, nw, w0, sh, k0, capw);
            else
                scatter4(std::true_type(), cur[u], sub ? p : 0u, nw, w0, sh, k0, capw);
        }
        // segments of more than 4 * L - 1 words (rare at the quality the host sized L for): the rest of their words
        if (__builtin_amdgcn_ballot_w64(more)) {
#pragma unroll 1
            for (int u = 0; u < U; u++) {
                int nw, k0;
                uint32_t sh, from;
                placement(e0, cnt, word0, u, nw, sh, k0, from);
                for (int w = w0 + 4 * L; w <= nw; w += 4 * L) {
                    const AsmWords4 c = src_words(from + (uint32_t)(w < nw ? w : 0));
                    const uint32_t p = __builtin_bswap32(src_words(from + (uint32_t)(w - 1)).w[0]);
                    scatter4(std::true_type(), c, p, nw, w, sh, k0 - w0 + w, capw);
                }
            }
        }
    };

    ASTAMP(2);
    for (uint32_t pass0 = 0; pass0 < img_end; pass0 += cap_bytes) { // image bytes [pass0, pass0 + cap_bytes) of this pass
        const uint32_t pass_bytes = min(cap_bytes, img_end - pass0);
        const int word0 = (int)(pass0 >> 2), capw = (int)((pass_bytes + 3u) >> 2); // the pass's words of the image
        if (pass0 != 0) {
            clear_image(pass_bytes);
            carry = 0;
        }
        for (int c0 = 0; c0 < nseg; c0 += kAsmChunk) {
            if (pass0 != 0 || c0 != 0) { // (the first chunk of the first pass was placed in the prologue)
                if (wave == 2 % kAsmWaves) scan_chunk(c0, carry);
                __syncthreads();
            }
            const int cnt = min(kAsmChunk, nseg - c0);
            for (int e0 = 0; e0 < cnt; e0 += per_trip * U) {
                AsmWords4 cur[U];
                load_trip(cur, e0, cnt, word0);
                place_trip(cur, e0, cnt, word0, capw);
            }
            ASTAMP(3);
            __syncthreads(); // the chunk's placement has been used; behind the last chunk: the image is complete
            ASTAMP(4);
        }

        // ---- 3. the image leaves: aligned 16-byte units, the group's two partial units byte by byte ----
        const uint4 *img4 = reinterpret_cast<const uint4 *>(img);
        uint8_t *const out_lo = reinterpret_cast<uint8_t *>(a_lo) + pass0; // uniform
        for (uint32_t k = tid; 16u * k < pass_bytes; k += kAsmThreads) {
            const uint32_t ib = pass0 + 16u * k; // image byte of the unit
            const uint4 v = img4[k];
            if (ib >= lead && ib + 16u <= img_end) {
                *reinterpret_cast<uint4 *>(out_lo + 16u * k) =
                    make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w));
            } else {
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t b = 0; b < 16u; b++)
                    if (ib + b >= lead && ib + b < img_end) out_lo[16u * k + b] = (uint8_t)(w4[b >> 2] >> (24u - 8u * (b & 3u)));
            }
        }
        ASTAMP(5);
        if (pass0 + cap_bytes < img_end) __syncthreads(); // the next pass clears the image
    }
    ASTAMP_FLUSH();
}
