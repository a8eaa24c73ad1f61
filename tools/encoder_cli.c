/* tools/encoder_cli.c — command-line caller of mpeg_encode_procedure (the role main.c:15-17 plays in
 * the reference, which hard-codes its four arguments).  With no arguments it behaves like the
 * reference's CLI: images/ -> bitstreams/awesome_video.mpeg at quality 12.
 *
 *   encoder [images_folder [bitstream_folder [video_path [quality [strict|full]]]]]
 *
 * EC504_CLI_REPEAT=<n> (tests, timing): the same call n times in this process; call k > 1 writes <video_path>.<k> and
 * reuses the GPU encoders and pinned buffers the library keeps between calls (encoder_release_cache()).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "encoder.h"

int main(int argc, char **argv) {
    const char *images = argc > 1 ? argv[1] : "images/";
    const char *bits = argc > 2 ? argv[2] : "bitstreams";
    const char *video = argc > 3 ? argv[3] : "bitstreams/awesome_video.mpeg";
    int quality = argc > 4 ? atoi(argv[4]) : 12;
    const char *rep = getenv("EC504_CLI_REPEAT");
    int repeat = rep ? atoi(rep) : 1, rc = 0;
    for (int k = 1; k <= (repeat < 1 ? 1 : repeat) && rc == 0; k++) {
        char path[512];
        if (k == 1)
            snprintf(path, sizeof path, "%s", video);
        else
            snprintf(path, sizeof path, "%s.%d", video, k);
        rc = argc > 5 ? mpeg_encode_procedure_region(images, bits, path, quality, strcmp(argv[5], "full") == 0)
                      : mpeg_encode_procedure(images, bits, path, quality);
    }
    encoder_release_cache();
    return rc;
}
