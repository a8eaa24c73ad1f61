/* tests/host_driver_standin.c — TEST INFRASTRUCTURE, never shipped or linked into libencoder.so.
 *
 * A CPU stand-in for the handful of m1v_* entry points that csrc/encoder_host.c calls, implemented with
 * the oracle (oracle/mpeg1_oracle.c).  tests/test_host_driver_cpu.py links it with encoder_host.c under
 * -fsanitize=thread / address so that the host driver's own logic — directory scan order, parallel
 * decode, batch staging, global frame indices across batches, .bit write-behind, error returns — runs
 * in this GPU-less container against the real reference binary's output.  It says nothing about the
 * HIP kernels; those are checked on the GPU box through the real library.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mpeg1_hip.h"
#include "mpeg1_oracle.h"

struct m1v_encoder {
    int W, H, C, qf, mode, max_frames;
};

static const char *g_err = "";

const char *m1v_last_error(void) { return g_err; }
int m1v_warm_up(int device) { (void)device; return M1V_OK; }
size_t m1v_file_prolog(uint8_t out[27]) { return orc_file_prolog(out); }

int m1v_create(m1v_encoder **out, int device, int w, int h, int channels, int qf, int mode, int max_frames) {
    (void)device;
    int xe, ye;
    orc_region(mode == M1V_MODE_FULL ? ORC_MODE_FULL : ORC_MODE_STRICT, w, h, &xe, &ye);
    if (xe > w || ye > h || channels < 3) {
        g_err = "picture smaller than the encoded region";
        return M1V_E_ARG;
    }
    m1v_encoder *e = (m1v_encoder *)calloc(1, sizeof *e);
    e->W = w, e->H = h, e->C = channels, e->qf = qf, e->max_frames = max_frames;
    e->mode = mode == M1V_MODE_FULL ? ORC_MODE_FULL : ORC_MODE_STRICT;
    *out = e;
    return M1V_OK;
}
void m1v_destroy(m1v_encoder *e) { free(e); }
size_t m1v_frame_bytes_in(const m1v_encoder *e) { return (size_t)e->W * e->H * e->C; }
size_t m1v_frame_bound(const m1v_encoder *e) { return orc_frame_bound(e->W, e->H, e->mode); }
size_t m1v_frame_bound_for(int width, int height, int mode) { return orc_frame_bound(width, height, mode); }
void *m1v_alloc_host(size_t bytes) { return malloc(bytes ? bytes : 1); }
void m1v_free_host(void *p) { free(p); }

long m1v_encode_planes_host(m1v_encoder *e, const uint8_t *rgb, int n, int first, uint8_t *out, size_t cap,
                            uint64_t *sizes, uint8_t *planes) {
    if (n > e->max_frames) {
        g_err = "n_frames exceeds max_frames";
        return M1V_E_ARG;
    }
    long r = orc_encode_frames(rgb, n, e->W, e->H, e->C, first, e->qf, e->mode, 1, out, cap, sizes);
    if (r == ORC_E_UNENCODABLE) return M1V_E_UNENCODABLE;
    if (r == ORC_E_NOSPACE) return M1V_E_NOSPACE;
    if (r < 0) return M1V_E_ARG;
    size_t npx = (size_t)e->W * e->H;
    for (int f = 0; planes && f < n; f++) {
        uint8_t *p = planes + 3 * npx * (size_t)f;
        orc_convert_rgb_to_ycbcr(rgb + npx * e->C * (size_t)f, e->C, npx, p, p + npx, p + 2 * npx);
    }
    return r;
}
