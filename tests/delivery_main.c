/* tests/delivery_main.c — a main.c-style C caller of the overlapped delivery (include/mpeg1_hip.h, m1v_delivery_*): synthetic
 * frames resident on the device, `batches` batches of `n` frames, every batch's frame records delivered to pinned host memory
 * under the next batch's encode and appended to a file the way the reference's driver appends its bit vector
 * (include/encoder.h:445 under /root/reference).  Plain C against libencoder.so; the test compares the file with the oracle.
 *   usage: delivery_main W H n batches out.mpeg */
#include <stdio.h>
#include <stdlib.h>
#include "mpeg1_hip.h"

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    const int W = atoi(argv[1]), H = atoi(argv[2]), n = atoi(argv[3]), batches = atoi(argv[4]);
    FILE *fp = fopen(argv[5], "wb");
    if (!fp) return 1;
    uint8_t prolog[27];
    fwrite(prolog, 1, m1v_file_prolog(prolog), fp);
    m1v_encoder *enc = NULL;
    m1v_delivery *d = NULL;
    if (m1v_create(&enc, 0, W, H, 3, 12, M1V_MODE_FULL, n) != M1V_OK || m1v_delivery_create(enc, 0, &d) != M1V_OK) {
        fprintf(stderr, "setup: %s\n", m1v_last_error());
        return 1;
    }
    /* two input buffers: batch k's pixels must stay until its copy has STARTED, i.e. until step k+1 has returned */
    uint8_t *d_rgb[2] = {NULL, NULL};
    const size_t frame = m1v_frame_bytes_in(enc);
    for (int b = 0; b < 2; b++)
        if (!(d_rgb[b] = (uint8_t *)m1v_alloc_device(frame * (size_t)n))) return 1;
    int rc = 0;
    for (int k = 0; k <= batches && rc == 0; k++) {
        int slot;
        if (k < batches) {
            if (m1v_synth_device(d_rgb[k & 1], frame, n, 504, (uint64_t)k * (uint64_t)n, NULL) != M1V_OK) rc = 1;
            slot = m1v_delivery_step(d, d_rgb[k & 1], n, k * n, NULL);
        } else {
            slot = m1v_delivery_flush(d);
        }
        if (slot < 0) {
            fprintf(stderr, "step %d: %s\n", k, m1v_last_error());
            rc = 1;
        } else if (slot != M1V_DELIVERY_NONE) {
            const uint8_t *p;
            const uint64_t *sizes;
            uint64_t bytes, sum = 0;
            if (m1v_delivery_wait(d, slot, &p, &bytes, &sizes) != M1V_OK) rc = 1;
            for (int i = 0; i < n; i++) sum += sizes[i];
            if (sum != bytes) rc = 3;
            fwrite(p, 1, (size_t)bytes, fp);
        }
    }
    m1v_delivery_destroy(d);
    for (int b = 0; b < 2; b++) m1v_free_device(d_rgb[b]);
    m1v_destroy(enc);
    fclose(fp);
    return rc;
}
