// ec504_imageencoder_amd/csrc/m1v_assemble.h — the ONE kernel behind an encode kernel (gfx950).  Not a standalone header: it is
// included by m1v_kernels.hip inside its anonymous namespace.
//
// What it replaces in the reference: the sequential tail of the frame loop — strip zero-padding (encoder.h:442-443),
// bitvector_fwrite of the picture's bits (:445), the packet-length back-patch (:448-454), the four trailer bytes
// (:456-458) and the ftell arithmetic that places frame i behind frame i-1 (:198) — and in this library, up to round 3, three
// launches (strip layout, frame offsets, a gather of one wave per strip at 0.195 of the HBM rate).
//
// Input, all produced by the encode kernel of the same batch (any of the three):
//   seg[frame][strip][q] = (bits, where)   what a strip is concatenated from: `bits` bits that start at word `where` of scratch
//   strip_ctr[frame][strip]                low 40 bits: bits of the strip (the tile kernel adds them up with one returning atomic
//                                          per segment; its upper bits count the arrivals)
//   frame_bytes[frame]                     sum over the frame's strips of ceil(bits / 8): added by whoever completed a strip
// so that NO value flows from workgroup to workgroup here: every workgroup derives its own place from those arrays.
//
// Mapping.  One workgroup (256 lanes) per (frame, GROUP of consecutive strips).  It
//   1. sums frame_bytes[0 .. frame) (+ 48 per frame: PKT SEQ GOP PIC + trailer) = where its frame starts, and the byte counts of
//      the frame's strips in front of its group = where its bytes start (two block reductions, all loads issued together);
//   2. scans the bit counts of its group's segments — the strip padding rides on each strip's last segment — which gives every
//      segment its destination bit in an LDS image of the group's output bytes; the image is laid out against the 16-byte
//      grid of the OUTPUT address, so that
//   3. SOURCE words scatter into it: 2^k lanes per segment read consecutive words (coalesced, several independent loads per
//      lane in flight: one memory latency for the whole group) and OR them in at their destination phase (two ds_or_b32);
//   4. the image leaves as aligned 16-byte stores; only the group's first and last partial units are written byte by byte
//      (their other bytes belong to the neighbouring groups, the frame header or the trailer).
// A group whose bytes outgrow the image takes several passes over its segments (any picture, any quality: correct; the host
// sizes groups so that noise at the encoder's quality needs one).
constexpr unsigned long long kCtrBitsMask = (1ull << 40) - 1; // strip_ctr: bits of the strip; above: tile rows that have arrived
constexpr int kCtrCountShift = 40;
constexpr int kAsmThreads = 256;
constexpr int kAsmMaxGroup = 16;
// LDS words behind the image: 12 x u64 reduction slots, 2 x 16 + 4 scan words, placement of <= 256 segments, per-strip bytes and padding
constexpr int kAsmRed = 0, kAsmScan = 24, kAsmDst = 60, kAsmBits = kAsmDst + kAsmThreads, kAsmSrc = kAsmBits + kAsmThreads,
              kAsmStripBytes = kAsmSrc + kAsmThreads, kAsmStripPad = kAsmStripBytes + kAsmMaxGroup, kAsmFixedWords = kAsmStripPad + kAsmMaxGroup;

struct AssembleArgs {
    int n_frames, n_strips, segs; // segs = segments per strip
    int group;                    // strips per workgroup (<= kAsmMaxGroup)
    int lanes_log2;               // 2^k lanes share a segment
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
};

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

// OR the top `nb` bits of `word` (MSB first) into the image at bit `rel` (relative to the pass; may lie outside it)
__device__ __forceinline__ void asm_place(uint32_t *img, int capw, int rel, uint32_t word, uint32_t nb) {
    const uint32_t val = nb >= 32u ? word : (word & ~(0xffffffffu >> nb));
    const int wi = rel >> 5; // arithmetic: floor for the words that start in front of the pass
    const uint32_t sh = (uint32_t)rel & 31u;
    const uint32_t hi = val >> sh, lo = sh ? val << (32u - sh) : 0u;
    if (hi && (unsigned)wi < (unsigned)capw) atomicOr(&img[wi], hi);
    if (lo && (unsigned)(wi + 1) < (unsigned)capw) atomicOr(&img[wi + 1], lo);
}

__global__ __launch_bounds__(kAsmThreads) void k_assemble(AssembleArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t asm_lds[];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.y, s0 = (int)blockIdx.x * a.group, ns = min(a.group, a.n_strips - s0), T = a.segs;
    uint32_t *img = asm_lds, *fixed = asm_lds + a.img_words;
    unsigned long long *red = reinterpret_cast<unsigned long long *>(fixed + kAsmRed);
    uint32_t *ws = fixed + kAsmScan, *sd = fixed + kAsmDst, *sb = fixed + kAsmBits, *sw = fixed + kAsmSrc, *sbytes = fixed + kAsmStripBytes,
             *spad = fixed + kAsmStripPad;

    // ---- 1. where the frame starts, where the group starts inside it ----
    const unsigned long long *ctr_f = a.strip_ctr + (size_t)f * a.n_strips;
    unsigned long long in_front = 0, before = 0, payload = 0;
    for (int i = tid; i < f; i += kAsmThreads) in_front += 48ull + a.frame_bytes[i];
    for (int s = tid; s < a.n_strips; s += kAsmThreads) {
        const unsigned long long b = ((ctr_f[s] & kCtrBitsMask) + 7ull) >> 3; // zero bits pad a strip to a byte, encoder.h:442-443
        payload += b;
        before += s < s0 ? b : 0ull;
    }
    if (tid < ns) {
        const unsigned long long bits = ctr_f[s0 + tid] & kCtrBitsMask, bytes = (bits + 7ull) >> 3;
        sbytes[tid] = (uint32_t)bytes;
        spad[tid] = (uint32_t)(8ull * bytes - bits);
    }
    in_front = wave_sum_u64(in_front);
    before = wave_sum_u64(before);
    payload = wave_sum_u64(payload);
    if (lane == 0) {
        red[wave * 3 + 0] = in_front;
        red[wave * 3 + 1] = before;
        red[wave * 3 + 2] = payload;
    }
    __syncthreads();
    const unsigned long long fo = red[0] + red[3] + red[6] + red[9], B0 = red[1] + red[4] + red[7] + red[10],
                             fs = 48ull + red[2] + red[5] + red[8] + red[11];
    const bool fits = fo + fs <= a.out_cap;

    // ---- housekeeping: sizes, total, status; the next batch's counters ----
    if (f < a.next_frames) {
        if (tid < ns) a.next_strip_ctr[(size_t)f * a.n_strips + s0 + tid] = 0ull;
        if (blockIdx.x == 0 && tid == 0) a.next_frame_bytes[f] = 0ull;
    }
    if (blockIdx.x == 0) {
        if (fits) frame_header_and_trailer(a.tab, a.out, fo, fs, a.first_index + f, tid);
        if (tid == 0) {
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

    uint32_t group_bytes = 0;
    for (int j = 0; j < ns; j++) group_bytes += sbytes[j];
    const uintptr_t A0 = reinterpret_cast<uintptr_t>(a.out) + fo + 44ull + B0, a_lo = A0 & ~(uintptr_t)15;
    const uint32_t lead = (uint32_t)(A0 - a_lo), img_end = lead + group_bytes; // the group's bytes are image bytes [lead, img_end)
    const uint32_t cap_bytes = (uint32_t)a.img_words * 4u;
    const int nseg = ns * T;
    const uint2 *seg_g = a.seg + ((size_t)f * a.n_strips + s0) * T;
    const uint32_t *src32 = reinterpret_cast<const uint32_t *>(a.scratch);
    const int lg = a.lanes_log2, L = 1 << lg, sub = tid & (L - 1), slot = tid >> lg, per_trip = kAsmThreads >> lg;
    constexpr int U = 4; // independent source loads per lane in flight

    int parity = 0;
    for (uint32_t pass0 = 0; pass0 < img_end; pass0 += cap_bytes) { // image bytes [pass0, pass0 + cap_bytes) of this pass
        const uint32_t pass_bytes = min(cap_bytes, img_end - pass0);
        {
            uint4 *img4 = reinterpret_cast<uint4 *>(img);
            for (uint32_t k = tid; k < ((pass_bytes + 15u) >> 4); k += kAsmThreads) img4[k] = make_uint4(0u, 0u, 0u, 0u);
        }
        const uint32_t bit0 = pass0 * 8u, bit1 = bit0 + pass_bytes * 8u;
        uint32_t carry = lead * 8u; // destination bit (image space) of the chunk's first segment
        for (int c0 = 0; c0 < nseg; c0 += kAsmThreads) {
            // ---- 2. destination bit of every segment: a scan of the bit counts, the strip's padding on its last segment ----
            const int q = c0 + tid;
            uint2 sg = make_uint2(0u, 0u);
            uint32_t v = 0;
            if (q < nseg) {
                sg = seg_g[q];
                const int j = (int)udiv((uint32_t)q, a.div_segs), t = q - j * T;
                v = sg.x + (t == T - 1 ? spad[j] : 0u);
            }
            uint32_t tot;
            const uint32_t excl = block_scan_exclusive_1b(v, ws, parity, kAsmThreads, tot);
            parity ^= 1;
            sd[tid] = carry + excl;
            sb[tid] = sg.x;
            sw[tid] = sg.y;
            carry += tot;
            const int any_long = __syncthreads_or(sg.x > 32u * (uint32_t)L); // (also: the image is cleared, the placement is in LDS)
            const int cnt = min(kAsmThreads, nseg - c0);

            // ---- 3. source words into the image: word `sub` of U segments per trip, all loads first ----
            for (int e0 = 0; e0 < cnt; e0 += per_trip * U) {
                uint32_t word[U], nb[U];
                int rel[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int e = e0 + u * per_trip + slot;
                    const bool ok = e < cnt;
                    const int ee = ok ? e : 0;
                    const uint32_t d = sd[ee] + 32u * (uint32_t)sub, b = ok ? sb[ee] : 0u;
                    const bool have = 32u * (uint32_t)sub < b && d + 32u > bit0 && d < bit1;
                    nb[u] = have ? min(32u, b - 32u * (uint32_t)sub) : 0u;
                    rel[u] = (int)(d - bit0);
                    word[u] = have ? src32[(size_t)sw[ee] + (uint32_t)sub] : 0u;
                }
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (nb[u]) asm_place(img, a.img_words, rel[u], __builtin_bswap32(word[u]), nb[u]);
            }
            if (any_long) { // segments of more than 32 * L bits: the rest of their words
                for (int e = slot; e < cnt; e += per_trip) {
                    const uint32_t b = sb[e], d0 = sd[e];
                    const size_t from = sw[e];
                    for (uint32_t w = (uint32_t)(sub + L); 32u * w < b; w += (uint32_t)L) {
                        const uint32_t d = d0 + 32u * w;
                        if (d + 32u > bit0 && d < bit1)
                            asm_place(img, a.img_words, (int)(d - bit0), __builtin_bswap32(src32[from + w]), min(32u, b - 32u * w));
                    }
                }
            }
            __syncthreads(); // the chunk's placement has been used; behind the last chunk: the image is complete
        }

        // ---- 4. the image leaves: aligned 16-byte units, the group's two partial units byte by byte ----
        const uint4 *img4 = reinterpret_cast<const uint4 *>(img);
        for (uint32_t k = tid; 16u * k < pass_bytes; k += kAsmThreads) {
            const uint32_t ib = pass0 + 16u * k; // image byte of the unit
            const uint4 v = img4[k];
            uint8_t *o = reinterpret_cast<uint8_t *>(a_lo + ib);
            if (ib >= lead && ib + 16u <= img_end) {
                *reinterpret_cast<uint4 *>(o) = make_uint4(__builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w));
            } else {
                const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t b = 0; b < 16u; b++)
                    if (ib + b >= lead && ib + b < img_end) o[b] = (uint8_t)(w4[b >> 2] >> (24u - 8u * (b & 3u)));
            }
        }
        if (pass0 + cap_bytes < img_end) __syncthreads(); // the next pass clears the image
    }
}
