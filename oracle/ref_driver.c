/* TEST INFRASTRUCTURE ONLY (oracle/): argv-forwarding main() around the *unmodified reference*
 * driver.  The reference defines mpeg_encode_procedure() inside its header
 * (/root/reference/include/encoder.h:20-498); this file is compiled against that header where it
 * lies, it contains no reference code.  Built only in the authoring container (oracle/Makefile,
 * target _ref); the resulting binary lives in oracle/_ref/ (git-ignored).
 *
 * usage: ref_encoder <images_folder> <bitstream_folder> <video_path> <quality_factor>
 */
#include "encoder.h"

int main(int argc, char **argv) {
    if (argc != 5) {
        fprintf(stderr, "usage: %s images_folder bitstream_folder video_path quality\n", argv[0]);
        return 64;
    }
    return mpeg_encode_procedure(argv[1], argv[2], argv[3], atoi(argv[4]));
}
