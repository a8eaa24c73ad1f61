/* tools/encoder_cli.c — command-line caller of mpeg_encode_procedure (the role main.c:15-17 plays in
 * the reference, which hard-codes its four arguments).  With no arguments it behaves like the
 * reference's CLI: images/ -> bitstreams/awesome_video.mpeg at quality 12.
 *
 *   encoder [images_folder [bitstream_folder [video_path [quality [strict|full]]]]]
 */
#include <string.h>
#include "encoder.h"

int main(int argc, char **argv) {
    const char *images = argc > 1 ? argv[1] : "images/";
    const char *bits = argc > 2 ? argv[2] : "bitstreams";
    const char *video = argc > 3 ? argv[3] : "bitstreams/awesome_video.mpeg";
    int quality = argc > 4 ? atoi(argv[4]) : 12;
    if (argc > 5)
        return mpeg_encode_procedure_region(images, bits, video, quality, strcmp(argv[5], "full") == 0);
    return mpeg_encode_procedure(images, bits, video, quality);
}
