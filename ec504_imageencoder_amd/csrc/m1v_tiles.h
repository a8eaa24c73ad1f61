// ec504_imageencoder_amd/csrc/m1v_tiles.h — the TILE form of the hot kernel (gfx950).  Not a standalone header: it is
// included by m1v_kernels.hip inside its anonymous namespace, behind the pieces it shares with the run kernels
// (colour conversion, fp32 FDCT, staging layout, VLC walk, bit packing helpers).
//
// Why tiles.  A strip is a 16-pixel wide COLUMN of macroblocks (encoder.h:238 iterates x outermost), i.e. 48 bytes of
// every 128-byte line of the picture; the run kernels (one lane per block down a strip, 24-byte row loads per lane)
// touch ~37 lines with every wave load and fetch every line 2.7 times into some L1 (profiles/r02_memory_path_pmc.txt:
// 4.8 x L1 fills per pixel byte).  Here a workgroup owns a TILE of 8 adjacent strips x 4 macroblock rows
// (128 x 64 pixels: rows of 384 bytes = three whole lines) and brings it in with LDS-DMA
// (global_load_lds_dwordx4: 1 KiB of whole lines per wave instruction, no vector registers), through a small ring of
// row-steps per wave; lanes then take their 24-byte block rows out of LDS.
//
//   workgroup = 3 waves = 192 lanes = the tile's 192 blocks (image_processing.c:138-150 extract_8x8_block order does
//   not matter before the bits are placed):
//     wave 0, 1   luma: macroblock rows 2w, 2w+1 of the tile.  lane = [mb row:1][block row (Y0Y1 | Y2Y3):1][strip:3][Y left|right:1]
//                 so row-step i of the wave (row i of each of its 64 blocks) is 4 picture rows x 384 contiguous bytes
//                 = 1536 bytes, and lane L's 24 bytes sit at L * 24: conflict-free ds_read_b64 x 3.
//     wave 2      chroma: Cb (lanes 0..31) and Cr (32..63) of the tile's 32 macroblocks; both read the SAME 24 bytes
//                 (encoder.h:347-348: the full-resolution plane addressed with stride W/2), which arrive once:
//                 row-step = 4 macroblock rows x 192 bytes.
//   Row-steps travel in a ring of R slots per wave (wave-private: ordered by the wave's own vmcnt, no barrier); the
//   staged levels of the wave's blocks later reuse the ring's bytes.
//
// Bits.  A tile holds 8 strip SEGMENTS (24 consecutive blocks of the strip's stream each).  Lanes write their block's
// bit count to LDS in emission order; after ONE barrier every wave scans all 192 counts itself (no second barrier),
// segments start on word boundaries of the tile's LDS image, and the tile records (bits, where) per segment;
// k_assemble (m1v_assemble.h) concatenates a strip's segments (encoder.h:442-445).

constexpr int kTileStrips = 8, kTileMbRows = 4;
constexpr int kTileThreads = kTileStrips * kTileMbRows * 6; // 192
constexpr int kTileSegBlocks = kTileMbRows * 6;             // 24 blocks of one strip
constexpr int kTileSlot = 2048;                             // bytes of one ring slot = one row-step of a wave (two 1-KiB LDS-DMA instructions)
// LDS words in front of the per-wave regions: one VLC table per wave, bit counts, prefix sums (+ total), segment table, spare
constexpr int kTileVlc = 0, kTileCnt = 3 * kVlcWords, kTileG = kTileCnt + 192, kTileSegTab = kTileG + 196, kTileMisc = kTileSegTab + 16,
              kTileFixedWords = kTileMisc + 4; // 984 words

struct TileArgs {
    Geometry g;
    const uint8_t *rgb;
    const Tables *tab;
    uint8_t *scratch;       // [frame][tile][slot_bytes] compact slots, then the overflow arena (as the run kernels)
    uint2 *seg;             // [frame][tile row][strip]: bits of the segment, where it starts (4-byte words from `scratch`)
    unsigned long long *strip_ctr;   // [frame][strip]: every tile adds (1 << 40 | its segment's bits) with ONE returning atomic: the tile
                                     // that sees tile_rows - 1 arrivals in front of it completes the strip and knows its bits
    unsigned long long *frame_bytes; // [frame]: that tile adds the strip's bytes (both zero before the batch: k_assemble of the batch
                                     // before clears them)
    uint32_t *arena_next;
    uint32_t slot_bytes, arena_slots;
    unsigned long long arena_off;
    uint32_t *status;
    int n_frames;
    int tile_cols, tile_rows, tiles_per_frame;
    DivMagic div_group, div_frame, div_cols; // 8 * tiles_per_frame, tiles_per_frame, tile_cols (udiv)
    const uint32_t *tile_row_order; // [tile_rows] (32-bit: a scalar load): which tile row the k-th group of tile_cols workgroups of a frame takes
    int lds_words;          // capacity of the LDS image of the tile's bits
    uint32_t run_cap;       // bytes of one arena slot: the worst case of a tile
    uint32_t luma_region, chroma_region; // LDS bytes of a wave's ring / staging region
    unsigned long long *stamps; // diagnostic builds only: [0..11] luma waves, [12..23] the chroma wave
};

// One LDS-DMA instruction: 16 bytes per active lane from (sbase + voff) to LDS (ldsdst + 16 * lane).  M0 carries the
// destination and is the compiler's: saved and restored inside the statement (cdna_hip_programming.md, inline asm).
__device__ __forceinline__ void dma16(uint32_t voff, uint32_t ldsdst, const uint8_t *sbase) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(ldsdst), "s"(sbase));
}
__device__ __forceinline__ void dma4(uint32_t voff, uint32_t ldsdst, const void *sbase) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(ldsdst), "s"(sbase));
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
}
// workgroup barrier that orders LDS only: the wave's LDS operations are retired, global operations in flight (the
// LDS-DMA row-steps) stay in flight across it (__syncthreads() would wait for them)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 24 bytes at an 8-byte aligned LDS address; loads and their wait in ONE statement (the compiler's waitcnt pass does
// not see LDS operations inside asm)
__device__ __forceinline__ Row24 ring_read24(uint32_t addr) {
    unsigned long long r0, r1, r2;
    asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:8\n\tds_read_b64 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                 : "v"(addr));
    Row24 v;
    v.d[0] = (uint32_t)r0; v.d[1] = (uint32_t)(r0 >> 32);
    v.d[2] = (uint32_t)r1; v.d[3] = (uint32_t)(r1 >> 32);
    v.d[4] = (uint32_t)r2; v.d[5] = (uint32_t)(r2 >> 32);
    return v;
}

// column pass + quantise + stage in LDS (as block_to_stage's second half); returns the DC level
template <bool STAGE8, int KEEP>
__device__ __forceinline__ int columns_to_stage(const RowStore<KEEP> &rows, const M1V_CONST_AS float *rq_t, uint32_t &lds_addr) {
    int dc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float c[8];
        m1vf::fdct_col_f<float>(rows.get(0, i), rows.get(1, i), rows.get(2, i), rows.get(3, i), rows.get(4, i), rows.get(5, i),
                                rows.get(6, i), rows.get(7, i), c, i == 0 ? RowStore<KEEP>::kBias0 : 0.0f);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = quant(c[u], rq_t[i * 8 + u]);
            const int p = scan_pos(u * 8 + i);
            if (p == 0) dc = q;
            if (STAGE8)
                asm("ds_write_b8 %0, %1 offset:%2" : "+v"(lds_addr) : "v"(q), "n"(stage_byte8(p)));
            else
                asm("ds_write_b16 %0, %1 offset:%2" : "+v"(lds_addr) : "v"(q), "n"(stage_byte16(p)));
        }
    }
    return dc;
}

// Row-pass outputs stay unpacked (RowStore<8>: 64 registers, 96 VGPRs = 5 waves per SIMD).  Packing all of them as f16 pairs
// (66 VGPRs, 6 waves per SIMD) won 2 % while the launches came in bursts of a few and the chip boosted, and lost 0.5-2.5 % in a
// sustained run, where the package sits at its 1400 W limit and the 96 extra conversions per block cost more than the sixth
// wave hides (tools/sustained.py, profiles/r03_ab_history.txt); that form was removed in round 4.
#ifndef M1V_TILE_KEEP
#define M1V_TILE_KEEP 8
#endif
// the 16-bit staging of qualities above 76 needs a few registers more in the column stage: three column pairs packed there
// (85 VGPRs; 4 unpacked columns already spill 40 bytes per lane)
#ifndef M1V_TILE_KEEP_WIDE
#define M1V_TILE_KEEP_WIDE 2
#endif
// Diagnostic build only (-DM1V_TILE_STAMPS, tools/tile_stamps.py): cycles a wave spends in each phase, kept in scalar
// registers and added to TileArgs::stamps once at the end (a global atomic inside the row loop would join the vmcnt queue).
#ifdef M1V_TILE_STAMPS
#define TSTAMP(ph)                                                                                 \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                              \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                      \
        tstamp_[ph] += now_ - tstamp_t_;                                                           \
        tstamp_t_ = now_;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } while (0)
#define TSTAMP_INIT()                                                                              \
    unsigned long long tstamp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                          \
    unsigned long long tstamp_t_ = __builtin_amdgcn_s_memtime();                                   \
    asm volatile("s_waitcnt lgkmcnt(0)")
#define TSTAMP_FLUSH()                                                                             \
    do {                                                                                           \
        if (lane == 0)                                                                             \
            for (int ph_ = 0; ph_ < 12; ph_++) atomicAdd(&a.stamps[ph_ + (chroma ? 12 : 0)], tstamp_[ph_]); \
    } while (0)
#else
#define TSTAMP(ph) do { } while (0)
#define TSTAMP_INIT() do { } while (0)
#define TSTAMP_FLUSH() do { } while (0)
#endif

#ifndef M1V_TILE_LEAN
#define M1V_TILE_LEAN false
#endif
#ifndef M1V_TILE_WAVES_PER_EU
#define M1V_TILE_WAVES_PER_EU 5
#endif

// The front half of a tile wave (shared by k_encode_tiles and k_coefficient_tiles): brings the wave's 64 blocks in through
// its LDS ring and leaves the 64 row-pass outputs of the lane's block in `rows`.  `ring` = LDS byte address of the wave's
// region; wave 0, 1 = luma, wave 2 = chroma (see the head of this file); `first` / `meanwhile`: see below.
template <int R, int KEEP, typename First, typename Meanwhile>
__device__ __forceinline__ void tile_pixel_rows(const Geometry &g, const uint8_t *fbase, uint32_t ring, int wave, int lane, int s0,
                                                int m0, int strips_here, int comp, First first, Meanwhile meanwhile,
                                                RowStore<KEEP> &rows) {
    const bool chroma = wave == 2;
    // ---- the lane's share of the wave's DMA.  One row-step = TWO 1-KiB LDS-DMA instructions into a 2-KiB ring slot, the
    //      same for the luma waves and the chroma wave (no branch, no EXEC mask, one vmcnt count):
    //        luma    units 0..63 | units 64..95 (lanes 32..63 repeat them into the slot's unused last 512 bytes)
    //        chroma  macroblock rows 0, 1 (2 x 192 bytes, lanes 0..23; the others repeat) | rows 2, 3, at +1024
    //      Pieces outside the picture region (last tile column / row) re-read bytes of the last strip / macroblock row;
    //      the lanes that own those blocks are not `valid`. ----
    uint32_t pitch;          // bytes from row i to row i + 1 of a block
    uint32_t voff_a, voff_b; // this lane's 16 bytes of row 0 (first / second instruction): byte offset from the frame base
    uint32_t lane_row;       // LDS address of the lane's 24 bytes inside slot 0
    // The four row offsets a wave's lanes choose from are products of UNIFORM values (scalar multiplies); a lane picks its own by
    // compare and select: per-lane 32-bit multiplies and the division of the lane index by 24 are quarter-rate instructions.
    const uint32_t third = (uint32_t)(lane >= 24) + (uint32_t)(lane >= 48); // lane / 24
    auto uniform = [](uint32_t v) { // an opaque scalar (the compiler removes a readfirstlane of a value it knows to be uniform)
        asm volatile("" : "+s"(v));
        return v;
    };
    if (!chroma) {
        pitch = (uint32_t)g.W * 3u;
        const uint32_t vw = (uint32_t)strips_here * 48u;
        // 16-byte unit L of the 1536-byte row-step: piece = L / 24 = picture row of the step, `within` inside its 384 bytes
        auto row_off = [&](uint32_t piece) { // uniform
            const uint32_t mb = (uint32_t)min(m0 + 2 * wave + (int)(piece >> 1), g.n_mbrows - 1);
            return ((mb * 16u + (piece & 1u) * 8u) * (uint32_t)g.W + (uint32_t)s0 * 16u) * 3u;
        };
        // (opaque: left visible, the compiler folds the selects back into per-lane multiplies)
        const uint32_t r0 = uniform(row_off(0)), r1 = uniform(row_off(1)), r2 = uniform(row_off(2)), r3 = uniform(row_off(3));
        // first instruction: units 0..63 -> pieces 0, 1, 2
        voff_a = (third == 0 ? r0 : (third == 1 ? r1 : r2)) + min(((uint32_t)lane - (third << 4) - (third << 3)) * 16u, vw - 16u);
        // second instruction: units 64 + (lane & 31) = 64..95 -> piece 2 (units 64..71) or 3
        const uint32_t l5 = (uint32_t)lane & 31u;
        voff_b = l5 < 8u ? r2 + min((16u + l5) * 16u, vw - 16u) : r3 + min((l5 - 8u) * 16u, vw - 16u);
        lane_row = ring + ((uint32_t)lane << 4) + ((uint32_t)lane << 3);
    } else {
        pitch = (uint32_t)g.half_w * 3u;
        const uint32_t vw = (uint32_t)strips_here * 24u;
        // (an odd number of strips ends in the middle of a 16-byte unit: that unit is still fetched whole — up to 8 bytes
        //  past the tile's last strip, still inside the first quarter of the frame, where all chroma sources lie)
        const uint32_t L = (uint32_t)lane - (third << 4) - (third << 3), piece = (uint32_t)(L >= 12u), within = min((L - (piece << 3) - (piece << 2)) * 16u, ((vw + 15u) & ~15u) - 16u);
        auto row_off = [&](uint32_t mbrow) { // uniform
            const uint32_t mb = (uint32_t)min(m0 + (int)mbrow, g.n_mbrows - 1);
            return ((mb * 8u) * (uint32_t)g.half_w + (uint32_t)s0 * 8u) * 3u;
        };
        const uint32_t r0 = uniform(row_off(0)), r1 = uniform(row_off(1)), r2 = uniform(row_off(2)), r3 = uniform(row_off(3));
        voff_a = (piece ? r1 : r0) + within;
        voff_b = (piece ? r3 : r2) + within;
        const uint32_t mrow = ((uint32_t)lane >> 3) & 3u, l3 = (uint32_t)lane & 7u;
        lane_row = ring + (mrow >> 1) * 1024u + (mrow & 1u) * 192u + (l3 << 4) + (l3 << 3);
    }
    constexpr uint32_t kSlot = 2048;
    auto issue_row = [&](int r) { // row-step r -> slot r % R.  M0 (the LDS destination) is set once: the second instruction's
                                  // offset:1024 moves its LDS address AND its source address, so its base is 1024 lower
        const uint8_t *sb = fbase + (size_t)r * pitch;
        const uint32_t dst = ring + (uint32_t)(r % R) * kSlot;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %4\n\tglobal_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff_a), "v"(voff_b), "s"(dst), "s"(sb), "s"(sb - 1024));
    };

    // ---- everything this wave needs from memory is requested up front: what the caller adds (older in the vmcnt queue than
    //      the rows), then R row-steps; the caller's other prologue work runs while they travel ----
    first();
#pragma unroll
    for (int r = 0; r < R; r++) issue_row(r);
    meanwhile();

    // ---- rows out of the ring as they land, the freed slot refilled with row i + R ----
    (void)comp;
    const CompCoefF kf = comp_coef_wave(!chroma, lane);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int newest = (i - 1 + R < 7) ? (i - 1 + R) : 7; // newest row-step requested so far
        const int behind = newest - i;                         // row-steps that may still be in flight: two instructions each
        if (behind == 0) wait_vm<0>(); else if (behind == 1) wait_vm<2>(); else if (behind == 2) wait_vm<4>();
        else if (behind == 3) wait_vm<6>(); else if (behind == 4) wait_vm<8>(); else if (behind == 5) wait_vm<10>();
        else if (behind == 6) wait_vm<12>(); else wait_vm<14>();
        const Row24 v = ring_read24(lane_row + (uint32_t)(i % R) * kSlot);
        if (i + R < 8) issue_row(i + R);
        float px[8];
        convert_row<3, M1V_TILE_LEAN>(v, kf, px);
        float ro[8];
        m1vf::fdct_row_f<float, true>(px, ro); // the wave rounds down: pixel_stage_rounds_down()
        rows.put(i, ro);
    }
}


template <bool STAGE8, int R>
__global__ __launch_bounds__(kTileThreads) __attribute__((amdgpu_waves_per_eu(M1V_TILE_WAVES_PER_EU, M1V_TILE_WAVES_PER_EU)))
void k_encode_tiles(TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const Geometry &g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool chroma = wave == 2; // wave-uniform
    constexpr int kStride = STAGE8 ? kStageStride8 : kStageStride16;

    uint32_t *vlc = lds + kTileVlc + wave * kVlcWords, *cnt = lds + kTileCnt, *G = lds + kTileG, *segtab = lds + kTileSegTab, *misc = lds + kTileMisc;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds;
    const uint32_t region_off = (uint32_t)kTileFixedWords * 4u + (uint32_t)wave * a.luma_region; // bytes from lds
    uint32_t *image = lds + kTileFixedWords + (2u * a.luma_region + a.chroma_region) / 4u;

    int frame, tile;
    frame_unit_of(blockIdx.x, a.n_frames, a.div_group, a.div_frame, frame, tile);
    // Tile rows are taken in an order that keeps the chroma quirk's re-reads in L2 (tile_row_order_for): the tile's position
    // in that order only decides WHEN it runs; everything it writes is indexed by the tile row itself.
    const int tk = (int)udiv((uint32_t)tile, a.div_cols), tc = tile - tk * a.tile_cols;
    const int tr = (int)a.tile_row_order[tk];
    tile = tr * a.tile_cols + tc;
    int s0 = tc * kTileStrips, m0 = tr * kTileMbRows;
    const uint8_t *fbase = pixel_stage_rounds_down(a.rgb + (unsigned long long)frame * g.frame_bytes, s0, m0);
    const unsigned long long tile_index = (unsigned long long)frame * a.tiles_per_frame + tile;

    // ---- which block a lane owns: strip j and macroblock row m inside the tile, block inside the macroblock (Y0 Y1 Y2 Y3 Cb Cr) ----
    auto owner = [&](int ln, int &j, int &m, int &blk) {
        if (!chroma) {
            m = 2 * wave + (ln >> 5);
            blk = ((ln >> 4) & 1) * 2 + (ln & 1);
            j = (ln >> 1) & 7;
        } else {
            m = (ln >> 3) & 3;
            blk = 4 + (ln >> 5);
            j = ln & 7;
        }
    };
    const int strips_here = min(kTileStrips, g.n_strips - s0); // >= 1
    int comp;
    {
        int j_, m_, blk_;
        owner(lane, j_, m_, blk_);
        comp = blk_ < 4 ? 0 : blk_ - 3;
    }

    // ---- pixel stage (tile_pixel_rows): the wave's own copy of the VLC table is requested first (a wave reads only its own
    //      copy, so the waves of a tile do not meet before the bit counts are exchanged), the image is cleared while the
    //      first rows travel (it is first touched in pass 2, behind the barrier of the bit counts) ----
    TSTAMP_INIT();
    const uint32_t ring = lds0 + region_off; // LDS byte address of this wave's region
    constexpr int kKeep = STAGE8 ? M1V_TILE_KEEP : M1V_TILE_KEEP_WIDE;
    RowStore<kKeep> rows;
    tile_pixel_rows<R, kKeep>(
        g, fbase, ring, wave, lane, s0, m0, strips_here, comp,
        [&]() {
#pragma unroll
            for (int q = 0; q < kVlcWords / kWave; q++)
                dma4((uint32_t)lane * 4u, lds0 + (uint32_t)(kTileVlc + wave * kVlcWords + q * kWave) * 4u, a.tab->vlc + q * kWave);
        },
        [&]() {
            uint4 *image4 = reinterpret_cast<uint4 *>(image); // 16-byte aligned, a.lds_words % 4 == 0 (configure_path)
            for (int k = tid; k < (a.lds_words >> 2); k += kTileThreads) image4[k] = make_uint4(0u, 0u, 0u, 0u);
        },
        rows);
    const M1V_CONST_AS float *rq_t = reinterpret_cast<const M1V_CONST_AS float *>(reinterpret_cast<uintptr_t>(a.tab->rq_t));
    // The lane's place in the tile, derived again behind the pixel stage (from an opaque copy of the lane id: five values
    // less to carry through the stage, whose register budget decides the waves per SIMD)
    int j, m, blk;
    {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        owner(ln, j, m, blk);
    }
    const bool valid = j < strips_here && m0 + m < g.n_mbrows;
    const int e = j * kTileSegBlocks + m * 6 + blk; // position in the tile's emission order (strip, macroblock, block)
    // every row-step has landed and has been read: the ring's bytes now hold the staged levels of the wave's blocks
    uint32_t *blkp = lds + region_off / 4u + lane * kStride;
    uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)blkp;
    TSTAMP(2);
    const int dc = columns_to_stage<STAGE8, kKeep>(rows, rq_t, lds_addr);
    const unsigned long long nz = (stage_nonzero_mask<STAGE8>(blkp, lds_addr) & ~1ull) | (dc != 0 ? 1ull : 0ull);
    TSTAMP(3);

    // ---- entropy pass 1 (private: own staged levels, shared read-only VLC table) ----
    auto fetch = [&](int p) -> int { return fetch_level<STAGE8>(blkp, p); };
    uint32_t hdr = 0, bad = 0;
    int hlen = 0;
    BlockBits bb = {0, 0};
    dc_header(dc, blk < 4, blk, vlc, hdr, hlen);
    const unsigned long long emit = emit_set(nz);
    // one coefficient per trip: two per trip (half the dependent LDS round trips) measured no faster in bursts and 1 % slower in
    // a sustained run — the lanes with a single coefficient left do the second one's work for nothing (r03_ab_history.txt)
    block_bits_pass1<STAGE8>(hdr, hlen, dc != 0, emit, vlc, fetch, bb.acc, bb.tot, bad);
    if (!valid) {
        bb.tot = 0;
        bad = 0;
    }
    cnt[e] = (uint32_t)bb.tot;
    TSTAMP(4);
    lds_barrier();
    TSTAMP(5);

    // ---- every wave scans the 192 counts (emission order) itself: no second barrier ----
    const uint32_t c0 = cnt[lane], c1 = cnt[64 + lane], c2 = cnt[128 + lane];
    const uint32_t i0 = wave_scan_inclusive(c0), i1 = wave_scan_inclusive(c1), i2 = wave_scan_inclusive(c2);
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)i0, 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)i1, 63);
    G[lane] = i0 - c0;                 // all three waves store the same values: whichever lands last, a wave reads what
    G[64 + lane] = t0 + i1 - c1;       // it wrote itself
    G[128 + lane] = t0 + t1 + i2 - c2;
    if (lane == 63) G[192] = t0 + t1 + i2;
    // segment table: lanes 0..7 = the tile's strips
    const uint32_t slice_bits = tr == 0 ? 38u : 0u; // the strip starts in this tile: slice header in front (mpeg1_blk.c:12-16)
    const uint32_t gs = G[min(lane, 8) * kTileSegBlocks];
    const uint32_t gs_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)gs, 0x101, 0xf, 0xf, true); // row_shl:1
    const uint32_t seg_bits = lane < strips_here ? slice_bits + (gs_next - gs) : 0u;
    const uint32_t seg_words = lane < 8 ? (seg_bits + 31u) >> 5 : 0u;
    const uint32_t seg_incl = row_scan_inclusive(seg_words);
    const uint32_t end_words = (uint32_t)__builtin_amdgcn_readlane((int)seg_incl, 7);
    if (lane < 8) {
        segtab[2 * lane] = gs;
        segtab[2 * lane + 1] = seg_incl - seg_words;
    }
    const uint32_t my_gs = segtab[2 * j], my_base = segtab[2 * j + 1];
    const uint32_t off = my_base * 32u + slice_bits + (G[e] - my_gs);
    TSTAMP(6);

    auto walk = [&](auto &sink) { walk_codes<STAGE8>(hdr, hlen, dc != 0, emit, vlc, fetch, sink); };
    // The strip's bit total (wave 0, lanes < strips_here): one returning atomic per segment; the tile whose add finds every other
    // tile row of the strip already counted knows the strip's bits and adds its bytes (zero bits pad a strip to a byte,
    // encoder.h:442-443) to the frame's total.  Only the values the atomics return travel between tiles: no fence.
    // (uniform 64-bit bases + 32-bit lane offsets: per-lane 64-bit index products are quarter-rate multiplies)
    unsigned long long *const strip_ctr_s0 = a.strip_ctr + ((unsigned long long)frame * (unsigned)g.n_strips + (unsigned)s0);
    auto strip_arrives = [&](uint32_t bits) -> unsigned long long {
#ifdef M1V_TILE_NOCOMPLETE // timing build (wrong sizes): what the returning atomic and the completion cost
        __hip_atomic_fetch_add(strip_ctr_s0 + lane, (1ull << kCtrCountShift) | (unsigned long long)bits,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0ull;
#endif
        return atomicAdd(strip_ctr_s0 + lane, (1ull << kCtrCountShift) | (unsigned long long)bits);
    };
    auto strip_completes = [&](unsigned long long before, uint32_t bits) {
        if ((uint32_t)(before >> kCtrCountShift) == (uint32_t)a.tile_rows - 1u)
            atomicAdd(&a.frame_bytes[frame], ((before & kCtrBitsMask) + bits + 7ull) >> 3);
    };
    // seg[frame][tile row][strip]: the tile's eight entries are 64 contiguous bytes (strip-major they were eight 32-byte
    // sectors 8 * tile_rows bytes apart: 256 bytes of write traffic for 64)
    uint2 *seg_out = a.seg + (((unsigned long long)frame * (unsigned)a.tile_rows + (unsigned)tr) * (unsigned)g.n_strips + (unsigned)s0) + lane; // lanes < strips_here
    auto slice_headers = [&](uint32_t *img, bool swapped) { // wave 0, lanes < strips_here
        if (tr == 0) {
            const uint32_t h0 = slice_word0(s0 + lane), h1 = kSliceWord1, w = seg_incl - seg_words;
            atomicOr(&img[w], swapped ? __builtin_bswap32(h0) : h0);
            atomicOr(&img[w + 1], swapped ? __builtin_bswap32(h1) : h1);
        }
    };

    // ---- image too large for LDS (rare): global atomics in a worst-case slot of the overflow arena ----
    if (end_words + 2 > (uint32_t)a.lds_words) {
        if (tid == 0) misc[0] = atomicAdd(a.arena_next, 1u);
        __syncthreads();
        const uint32_t got = misc[0];
        if (got >= a.arena_slots) { // arena exhausted: the caller re-encodes after m1v_reserve_scratch
            if (tid == 0) atomicOr(a.status, (uint32_t)M1V_STATUS_SCRATCH);
            if (wave == 0 && lane < strips_here) {
                *seg_out = make_uint2(0u, 0u);
                strip_completes(strip_arrives(0u), 0u); // sizes stay consistent; the batch is flagged and encoded again
            }
            return;
        }
        const unsigned long long where = a.arena_off + (unsigned long long)got * a.run_cap;
        uint32_t *big = reinterpret_cast<uint32_t *>(a.scratch + where);
        for (uint32_t i = tid; i < (a.run_cap >> 2); i += kTileThreads) big[i] = 0;
        __syncthreads();
        if (wave == 0 && lane < strips_here) {
            slice_headers(big, true);
            *seg_out = make_uint2(seg_bits, (uint32_t)(where >> 2) + (seg_incl - seg_words));
            strip_completes(strip_arrives(seg_bits), seg_bits);
        }
        if (valid) put_block<true>(big, off, bb, walk);
        if (bad) atomicOr(a.status, (uint32_t)M1V_STATUS_UNENCODABLE);
        return;
    }

    // ---- common path: OR the bits into the LDS image, store it once to the tile's compact slot ----
    unsigned long long arrived = 0; // requested here, looked at behind pass 2 and the store
    if (wave == 0 && lane < strips_here) {
        slice_headers(image, false);
        *seg_out = make_uint2(seg_bits, (uint32_t)((tile_index * a.slot_bytes) >> 2) + (seg_incl - seg_words));
        arrived = strip_arrives(seg_bits);
    }
    if (valid) put_block<false>(image, off, bb, walk);
    TSTAMP(7);
    lds_barrier();
    TSTAMP(8);
    // (in front of the stores: the answer has been back since pass 2, and waiting for it here does not wait for the stores)
    if (wave == 0 && lane < strips_here) strip_completes(arrived, seg_bits);
    uint32_t *slot32 = reinterpret_cast<uint32_t *>(a.scratch + tile_index * a.slot_bytes);
    for (uint32_t i = tid; i < end_words; i += kTileThreads) slot32[i] = __builtin_bswap32(image[i]);
    if (bad) atomicOr(a.status, (uint32_t)M1V_STATUS_UNENCODABLE);
    TSTAMP(9);
    TSTAMP_FLUSH();
}

// ---- BASELINE config 2 on tiles: FDCT + quantise + zigzag only (image_processing.c:192-381), int16 levels out ----------
// The same front half as k_encode_tiles; the 64 levels of a block are staged as int16 at their zigzag index (block stride 34
// words: 8-byte aligned, two lanes per bank), and the tile's eight strip segments (24 consecutive blocks = 3 KiB of the
// output each) leave as whole 128-byte lines: 8 bytes per lane and store instruction, consecutive lanes consecutive bytes.
struct CoefTileArgs {
    Geometry g;
    const uint8_t *rgb;
    const Tables *tab;
    int16_t *out; // [frame][strip][macroblock][Y0 Y1 Y2 Y3 Cb Cr][64]
    int n_frames, tile_cols, tile_rows, tiles_per_frame;
    uint32_t region; // LDS bytes of a wave's ring / staging region
};
constexpr int kCoefStride = 34; // words per staged block

template <int R>
__global__ __launch_bounds__(kTileThreads) __attribute__((amdgpu_waves_per_eu(M1V_TILE_WAVES_PER_EU, M1V_TILE_WAVES_PER_EU)))
void k_coefficient_tiles(CoefTileArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const Geometry &g = a.g;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)lds;
    int frame, tile;
    frame_strip_of(blockIdx.x, a.n_frames, a.tiles_per_frame, frame, tile);
    const int tr = tile / a.tile_cols, tc = tile - tr * a.tile_cols;
    int s0 = tc * kTileStrips, m0 = tr * kTileMbRows;
    const uint8_t *fbase = pixel_stage_rounds_down(a.rgb + (unsigned long long)frame * g.frame_bytes, s0, m0);
    const int strips_here = min(kTileStrips, g.n_strips - s0), mrows_here = min(kTileMbRows, g.n_mbrows - m0);
    const int comp = wave == 2 ? 1 + (lane >> 5) : 0;

    RowStore<M1V_TILE_KEEP> rows;
    tile_pixel_rows<R, M1V_TILE_KEEP>(g, fbase, lds0 + (uint32_t)wave * a.region, wave, lane, s0, m0, strips_here, comp, [] {}, [] {}, rows);

    // column pass + quantise; level of zigzag position p -> int16 p of the lane's staged block (the ring is drained: its bytes are reused)
    const M1V_CONST_AS float *rq_t = reinterpret_cast<const M1V_CONST_AS float *>(reinterpret_cast<uintptr_t>(a.tab->rq_t));
    uint32_t lds_addr = lds0 + (uint32_t)wave * a.region + (uint32_t)lane * (kCoefStride * 4u);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        float c[8];
        m1vf::fdct_col_f<float>(rows.get(0, i), rows.get(1, i), rows.get(2, i), rows.get(3, i), rows.get(4, i), rows.get(5, i),
                                rows.get(6, i), rows.get(7, i), c, i == 0 ? RowStore<M1V_TILE_KEEP>::kBias0 : 0.0f);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = quant(c[u], rq_t[i * 8 + u]);
            asm("ds_write_b16 %0, %1 offset:%2" : "+v"(lds_addr) : "v"(q), "n"(2 * scan_pos(u * 8 + i)));
        }
    }
    // every wave's blocks staged (the barrier takes the address register the stores are chained through)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::"v"(lds_addr) : "memory");

    // store: unit = 16 bytes (two 8-byte LDS reads: the staged blocks are 8-byte aligned); a strip segment = 24 blocks x 8
    // units, in emission order (macroblock row, then Y0 Y1 Y2 Y3 Cb Cr)
    const int bps = g.n_mbrows * 6;
    const unsigned char *lds_bytes = reinterpret_cast<const unsigned char *>(lds);
    const int units_valid = mrows_here * 6 * 8;
#pragma unroll
    for (int t = 0; t < kTileStrips * kTileSegBlocks * 8 / kTileThreads; t++) {
        const int u = tid + t * kTileThreads;
        const int j = u / (kTileSegBlocks * 8), within = u - j * (kTileSegBlocks * 8);
        if (j >= strips_here || within >= units_valid) continue;
        const int b = within >> 3, part = within & 7, m = b / 6, blk = b - m * 6;
        // who staged block (j, m, blk): see the lane order of the luma / chroma waves at the head of this file
        const int w_src = blk < 4 ? (m >> 1) : 2;
        const int l_src = blk < 4 ? (m & 1) * 32 + (blk >> 1) * 16 + j * 2 + (blk & 1) : (blk - 4) * 32 + m * 8 + j;
        const unsigned char *from = lds_bytes + (size_t)w_src * a.region + (size_t)l_src * (kCoefStride * 4) + part * 16;
        const uint2 lo = *reinterpret_cast<const uint2 *>(from), hi = *reinterpret_cast<const uint2 *>(from + 8);
        int16_t *o = a.out + (((size_t)frame * g.n_strips + (size_t)(s0 + j)) * bps + (size_t)m0 * 6) * 64;
        *reinterpret_cast<uint4 *>(reinterpret_cast<unsigned char *>(o) + (size_t)within * 16) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

