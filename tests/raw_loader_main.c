/* tests/raw_loader_main.c — TEST INFRASTRUCTURE: a caller of mpeg_encode_procedure_region that registers a
 * loader for files holding raw pixels (int32 W, H, C, then W*H*C bytes) instead of a JPEG decoder, so tests
 * can feed the host driver exact pixels (pictures JPEG would smooth away).
 *   raw_loader_main images bits video quality strict|full */
#include <string.h>
#define EC504_NO_STB
#include "encoder.h"

static unsigned char *load_raw(char const *path, int *w, int *h, int *c, int desired) {
    (void)desired;
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    int hdr[3];
    unsigned char *px = NULL;
    if (fread(hdr, sizeof hdr, 1, f) == 1 && hdr[0] > 0 && hdr[1] > 0 && hdr[2] > 0) {
        size_t n = (size_t)hdr[0] * hdr[1] * hdr[2];
        px = (unsigned char *)malloc(n);
        if (px && fread(px, 1, n, f) != n) {
            free(px);
            px = NULL;
        }
        *w = hdr[0], *h = hdr[1], *c = hdr[2];
    }
    fclose(f);
    return px;
}

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    encoder_set_image_loader(load_raw, free);
    return mpeg_encode_procedure_region(argv[1], argv[2], argv[3], atoi(argv[4]), strcmp(argv[5], "full") == 0);
}
