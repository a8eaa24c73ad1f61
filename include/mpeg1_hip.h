/* include/mpeg1_hip.h — C-ABI of the MI355X (gfx950) MPEG-1 I-frame hot path.
 *
 * This is the boundary a host program (C, JNI, cgo, ctypes ...) binds.  Plain pointers and sizes
 * only; no C++ or torch types.  Implemented by libencoder.so (ec504_imageencoder_amd/csrc/).
 *
 * What each entry point replaces in the reference (eburhansjah/ec504_ImageEncoder, paths under
 * /root/reference):
 *
 *   m1v_encode_device / m1v_encode_host
 *       the per-frame body of mpeg_encode_procedure, include/encoder.h:196-458:
 *       packet/sequence/GOP/picture headers (source/mpeg1_enc.c:47-129), convert_rgb_to_ycbcr
 *       (source/image_processing.c:68), the slice -> macroblock -> block loops (encoder.h:238-444:
 *       extract_8x8_block :138, fast_DCT :192, quantization :349, zigzag_scanning :373,
 *       run_length_encode :703, encode_block_header_i source/mpeg1_blk.c:67, VLC_encode
 *       image_processing.c:400, encode_blk_coeff source/vlc.c:315, bit_vector.c appends),
 *       the strip zero-padding (encoder.h:442), the 16-bit length back-patch (encoder.h:448-453)
 *       and the 4 trailing bytes (encoder.h:456-458).
 *   m1v_encode_planes_host     one frame-loop iteration per frame, complete: the frame record AND the planes that
 *                              write_to_bitstream (image_processing.c:753, called at encoder.h:461-465) stores
 *   m1v_set_pipelined / m1v_flush   no reference counterpart: overlap of one batch's gather with the next encode
 *   m1v_warm_up, m1v_alloc_host/_free_host   no reference counterpart: runtime start-up off the critical path, pinned buffers
 *   m1v_coefficients_device    fast_DCT + quantization + zigzag_scanning only (BASELINE config 2)
 *   m1v_convert_device/_host   convert_rgb_to_ycbcr, image_processing.c:68-110 (feeds the .bit files,
 *                              write_to_bitstream image_processing.c:753)
 *   m1v_subsample_device       subsampling_420, image_processing.c:114-133 (dead in the reference's
 *                              data flow; provided for completeness)
 *   m1v_file_prolog            mpeg1_file_header + mpeg1_sys_header, mpeg1_enc.c:7-44 / encoder.h:86-89
 *   m1v_synth_device           no reference counterpart: device-side synthetic frames for benchmarks
 *
 * The coarse, drop-in entry point mpeg_encode_procedure() is declared in include/encoder.h.
 *
 * Threads: an m1v_encoder is used by one thread at a time (different encoders, also on the same GPU, may be
 * used concurrently); m1v_last_error() is per thread.
 *
 * All *_device entry points are asynchronous on `stream` (a hipStream_t passed as void*, NULL =
 * the default stream) and take DEVICE pointers.  Return value: M1V_OK or a negative M1V_E_*.
 */
#ifndef MPEG1_HIP_H
#define MPEG1_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Region covered by the macroblock loops:
 *   STRICT  x in [0,96), y in [0,144): the literals at encoder.h:238 / :248 (byte-identical to the
 *           unmodified reference)
 *   FULL    x in [0, W & ~15), y in [0, H & ~15): the reference with those two literals restored */
enum { M1V_MODE_STRICT = 0, M1V_MODE_FULL = 1 };

enum {
    M1V_OK = 0,
    M1V_E_ARG = -1,         /* bad argument / geometry the reference would read out of bounds with  */
    M1V_E_UNENCODABLE = -2, /* an emitted AC level has |level| >= 256 (reference: vlc.c:349 -> NULL
                               -> segfault); output of the batch is undefined                       */
    M1V_E_NOSPACE = -3,     /* output buffer too small                                             */
    M1V_E_HIP = -4,         /* HIP runtime error, see m1v_last_error()                             */
    M1V_E_NODEVICE = -5,    /* no usable gfx950 device                                             */
    M1V_E_SCRATCH = -6      /* the batch needs more scratch than reserved: m1v_reserve_scratch(enc, 1)
                               and encode again (the host-buffer entry points do so themselves)       */
};

/* bits of the device status word (m1v_encode_device's d_status) */
enum { M1V_STATUS_UNENCODABLE = 1u, M1V_STATUS_NOSPACE = 2u, M1V_STATUS_SCRATCH = 4u };

typedef struct m1v_encoder m1v_encoder;

int m1v_device_count(void);
const char *m1v_last_error(void); /* thread-local, never NULL */

/* One encoder = one device, one picture geometry, one quality factor.  max_frames bounds the batch
 * a single m1v_encode_* call may carry (scratch is sized for it). */
int m1v_create(m1v_encoder **out, int device, int width, int height, int channels,
               int quality_factor, int mode, int max_frames);
void m1v_destroy(m1v_encoder *enc);

/* Scratch policy.  Every run of 256 blocks owns a compact slot (what the kernel's LDS image of its bits can hold: 2 KiB at
 * quality <= 25); runs that emit more build their bits in a worst-case sized slot (28 KiB) taken from an overflow arena.  By
 * default the arena holds 1/256 of the runs — about 4x the payload in total at quality 12, where round 1 reserved 47x.
 * A batch that exhausts it reports M1V_STATUS_SCRATCH in its status word (its output is then undefined):
 * m1v_reserve_scratch(enc, 1) sizes the arena for every run (no batch can exhaust that), 0 returns to the default.
 * Both reallocate: call them between batches. */
int m1v_reserve_scratch(m1v_encoder *enc, int worst_case);
size_t m1v_scratch_bytes(const m1v_encoder *enc); /* device bytes currently held as scratch */

/* geometry helpers */
int m1v_strips(const m1v_encoder *enc);          /* x_extent / 16 */
int m1v_mb_rows(const m1v_encoder *enc);         /* y_extent / 16 */
size_t m1v_frame_bound(const m1v_encoder *enc);  /* worst-case bytes of one frame record */
size_t m1v_frame_bound_for(int width, int height, int mode); /* the same without an encoder (0: geometry not encodable) */
size_t m1v_frame_bytes_in(const m1v_encoder *enc); /* width*height*channels */

/* PACK(12)+SYS(15), written once per file.  Returns 27. */
size_t m1v_file_prolog(uint8_t out[27]);

/* n_frames frames, contiguous in d_rgb (interleaved, width*height*channels bytes each), to
 * contiguous frame records in d_out.  first_frame_index is the global index of frame 0 of the batch
 * (it drives the `hour` fields, encoder.h:42,475-484).  d_rgb is only read, but a kernel may read up to the next
 * 4-byte boundary past a row's last byte (three bytes at most, inside the aligned word that holds that byte; pictures
 * whose width is not a multiple of 8 in a 4-byte aligned buffer): allocations are at least 4-byte granular, so this
 * never leaves the buffer's last word.
 *   d_frame_sizes  uint64[n_frames] bytes of each record (may be NULL)
 *   d_total        uint64[1] total bytes written (may be NULL)
 *   d_status       uint32[1] OR of M1V_STATUS_* (may be NULL; M1V_STATUS_SCRATCH: see m1v_reserve_scratch) */
int m1v_encode_device(m1v_encoder *enc, const uint8_t *d_rgb, int n_frames, int first_frame_index,
                      uint8_t *d_out, size_t out_cap, uint64_t *d_frame_sizes, uint64_t *d_total,
                      uint32_t *d_status, void *stream);

/* Pipelined mode (off by default).  When on, m1v_encode_device launches only the encode kernel on `stream`;
 * the layout scans and the gather into d_out run on an internal stream, so the NEXT batch's encode kernel
 * (launched on `stream`) overlaps them.  d_out / d_frame_sizes / d_total / d_status of a batch are complete
 * once work enqueued behind m1v_flush(enc, s) on stream s has started; callers double-buffer d_out.  The
 * internal scratch is double-buffered, so at most two batches are in flight. */
int m1v_set_pipelined(m1v_encoder *enc, int enable);
/* Makes `stream` wait for every gather still pending on the internal stream.  An encoder in pipelined mode is meant to
 * be driven from ONE stream; calls on another stream are still ordered behind the gather that last used their set
 * of internal buffers. */
int m1v_flush(m1v_encoder *enc, void *stream);

/* Overlapped delivery of the frame records to the HOST (the path ends in a host bitstream: the reference writes it with
 * bitvector_fwrite, include/encoder.h:445).  Input resident in device memory; the device-to-host copy of batch k runs on an
 * internal stream under the encode of batch k+1, through two device output buffers and two pinned host buffers owned by the
 * delivery object.  Nothing is allocated inside the loop; one host wait per step (for 16 bytes: the batch's byte count and
 * status word).  A batch that ran out of overflow scratch is encoded again with the worst case reserved; any other status
 * bit fails the step (nothing of an undefined batch is delivered).
 *
 *   m1v_delivery_create(enc, out_cap, &d)      out_cap = capacity of each output buffer in bytes (0: max_frames x frame bound)
 *   slot = m1v_delivery_step(d, d_rgb, n, first_index, stream)
 *                                              queues the encode of this batch on `stream`, then starts the copy of the batch
 *                                              BEFORE it; returns that batch's slot (0 or 1), M1V_DELIVERY_NONE on the first
 *                                              call, or a negative M1V_E_*.  d_rgb must stay valid until the batch's copy has started.
 *   slot = m1v_delivery_flush(d)               starts the copy of the last batch (M1V_DELIVERY_NONE if none is pending)
 *   m1v_delivery_wait(d, slot, &host, &bytes, &frame_sizes)
 *                                              blocks until that slot's copy has arrived; host / frame_sizes point into the
 *                                              slot's pinned buffers, valid until the second step after the one that returned it
 * The delivery object must not be stepped or flushed after its encoder has been destroyed (wait and destroy are fine). */
typedef struct m1v_delivery m1v_delivery;
enum { M1V_DELIVERY_NONE = 2 };
int m1v_delivery_create(m1v_encoder *enc, size_t out_cap, m1v_delivery **out);
void m1v_delivery_destroy(m1v_delivery *d);
int m1v_delivery_step(m1v_delivery *d, const uint8_t *d_rgb, int n_frames, int first_frame_index, void *stream);
int m1v_delivery_flush(m1v_delivery *d);
int m1v_delivery_wait(m1v_delivery *d, int slot, const uint8_t **host, uint64_t *bytes, const uint64_t **frame_sizes);
uint64_t m1v_delivery_bytes(const m1v_delivery *d, int slot); /* bytes of the batch in `slot`: known once its copy has been started */

/* Device memory for callers that do not link the HIP runtime themselves (a plain-C caller of m1v_encode_device /
 * m1v_delivery_step, tests/delivery_main.c): hipMalloc / hipFree on the current device.  NULL on failure. */
void *m1v_alloc_device(size_t bytes);
void m1v_free_device(void *p);

/* Starts the GPU runtime for `device` (context, code objects) so that a later m1v_create() does not pay for it.
 * Optional; meant to be called from another thread while the caller is still busy with host work. */
int m1v_warm_up(int device);

/* Host-buffer convenience (PCIe inclusive, synchronous): returns total bytes or negative M1V_E_*. */
long m1v_encode_host(m1v_encoder *enc, const uint8_t *rgb, int n_frames, int first_frame_index,
                     uint8_t *out, size_t out_cap, uint64_t *frame_sizes);

/* m1v_encode_host plus, when planes != NULL, the full-resolution Y/Cb/Cr planes of m1v_convert_host from the SAME
 * upload (what one frame-loop iteration of the reference produces: its frame record, encoder.h:196-458, and the
 * content of image_<k>.bit, encoder.h:461-465).  The upload runs in two halves so the download of the first
 * half's planes (PCIe is full duplex) overlaps the upload of the second. */
long m1v_encode_planes_host(m1v_encoder *enc, const uint8_t *rgb, int n_frames, int first_frame_index,
                            uint8_t *out, size_t out_cap, uint64_t *frame_sizes, uint8_t *planes);

/* Pinned host memory for the buffers handed to the *_host entry points (optional; any host pointer works, pinned
 * ones are copied at the PCIe rate). */
void *m1v_alloc_host(size_t bytes);
void m1v_free_host(void *p);

/* The 64 zigzag-ordered quantised levels of every visited block, int16, in emission order
 * [frame][strip][macroblock][Y0 Y1 Y2 Y3 Cb Cr][64]. */
int m1v_coefficients_device(m1v_encoder *enc, const uint8_t *d_rgb, int n_frames, int16_t *d_coeffs,
                            void *stream);

/* Full-resolution planes per frame: Y[w*h] Cb[w*h] Cr[w*h] (3*w*h bytes per frame). */
int m1v_convert_device(m1v_encoder *enc, const uint8_t *d_rgb, int n_frames, uint8_t *d_planes,
                       void *stream);
/* Host-buffer form of m1v_convert_device (synchronous; feeds the image_<k>.bit side files). */
int m1v_convert_host(m1v_encoder *enc, const uint8_t *rgb, int n_frames, uint8_t *planes);
/* 2x2 truncated mean of one Cb and one Cr plane (even width and height). */
int m1v_subsample_device(m1v_encoder *enc, const uint8_t *d_cb, const uint8_t *d_cr,
                         uint8_t *d_cb_sub, uint8_t *d_cr_sub, void *stream);

/* byte k of frame f = byte (k & 7) of splitmix64(seed + f*0x9E3779B97F4A7C15 + (k >> 3)), little-endian */
int m1v_synth_device(uint8_t *d_rgb, size_t bytes_per_frame, int n_frames, uint64_t seed,
                     uint64_t first_frame_index, void *stream);

/* Kernel timing by HIP events recorded on the launch stream around the dominant kernel
 * (k_encode_tiles; k_encode_dense / k_encode_strips on the run path).  enable!=0 starts collecting; m1v_profile_read synchronises the recorded events
 * and returns launches/total milliseconds since the last read. */
int m1v_profile_enable(m1v_encoder *enc, int enable);
int m1v_profile_read(m1v_encoder *enc, int *launches, double *total_ms);
/* The same, one duration per launch: ms[0 .. min(cap, *launches)) in launch order (for min / median / spread). */
int m1v_profile_read_times(m1v_encoder *enc, float *ms, int cap, int *launches);

/* Test hook: capacity in 32-bit words of the per-strip LDS bit buffer (0 = default).  A tiny value
 * forces the global-memory fallback path so that tests can cover it. */
int m1v_debug_set_lds_words(m1v_encoder *enc, int words);
/* Two encode kernels serve the path.  TILES (default for 3-channel pictures of any width and alignment): a workgroup
 * owns 8 adjacent strips x 4 macroblock rows and brings the pixels in as whole 128-byte lines by LDS-DMA.  RUNS (4-channel
 * pictures): a workgroup owns 256 consecutive blocks of the stream, every lane loads its own 24-byte block rows.  Both
 * produce the same bytes.  Test hook: -1 = by geometry, 0 = runs, 1 = tiles (3 channels only). */
int m1v_debug_set_path(m1v_encoder *enc, int path);
int m1v_path_in_use(const m1v_encoder *enc); /* 1 = tiles, 0 = runs */
/* Test hook: the nth device allocation made from now on by a reconfiguration (m1v_reserve_scratch, m1v_set_pipelined,
 * the m1v_debug_set_* hooks) fails as if the device were out of memory; 0 = off.  A failed reconfiguration returns
 * M1V_E_HIP and leaves the encoder exactly as it was.  Inert unless the process runs with EC504_DEBUG_HOOKS=1. */
void m1v_debug_fail_alloc(int nth);
/* Test hook: force how the RUN kernel loads its pixels (forcing a mode selects the run path): -1 = automatic (by width,
 * channel count and pointer alignment), 0 = byte loads (valid everywhere), 2 = 28-byte loads + funnel shift (3 channels,
 * 4-byte aligned buffer).  A mode that is not valid for the buffer at hand is ignored.  Lets tests compare the load paths
 * on one buffer. */
int m1v_debug_set_input_mode(m1v_encoder *enc, int mode);
/* Tuning/test hook: blocks per workgroup of the RUN kernel (multiple of 64, 64..384, not more than the blocks of one
 * strip; 0 = default).  Forcing a run length selects the run path unless m1v_debug_set_path says tiles. */
int m1v_debug_set_dense_threads(m1v_encoder *enc, int threads);

#ifdef __cplusplus
}
#endif
#endif
