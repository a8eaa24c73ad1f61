/* TEST INFRASTRUCTURE ONLY (oracle/): dumps the pixels the reference driver would see.
 * Enumerates <images_folder> exactly like /root/reference/include/encoder.h:119-172 does
 * (raw readdir order, substring match on ".jpg"/".jpeg", "%s/%s" path join, stbi_load(...,0))
 * and writes, for every image that loads:
 *     <out_prefix>.order   one file name per line, in enumeration order
 *     <out_prefix>.rgb     int32 n, then per image: int32 w, int32 h, int32 channels, w*h*channels bytes
 * The JPEG decoder is the reference's vendored stb_image.h (v2.30), compiled where it lies.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dirent.h>
#define STB_IMAGE_IMPLEMENTATION
#include "stb_image.h"

int main(int argc, char **argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s images_folder out_prefix\n", argv[0]); return 64; }
    char path[1024];
    snprintf(path, sizeof path, "%s.order", argv[2]);
    FILE *fo = fopen(path, "w");
    snprintf(path, sizeof path, "%s.rgb", argv[2]);
    FILE *fr = fopen(path, "wb");
    if (!fo || !fr) { perror("open output"); return 1; }
    int n = 0;
    fwrite(&n, 4, 1, fr);
    DIR *dir = opendir(argv[1]);
    if (!dir) { perror("opendir"); return 1; }
    struct dirent *e;
    while ((e = readdir(dir)) != NULL) {
        if (strstr(e->d_name, ".jpg") == NULL && strstr(e->d_name, ".jpeg") == NULL) continue;
        char fp[256];
        snprintf(fp, sizeof fp, "%s/%s", argv[1], e->d_name);
        int w, h, c;
        unsigned char *d = stbi_load(fp, &w, &h, &c, 0);
        if (!d) continue;
        fprintf(fo, "%s\n", e->d_name);
        fwrite(&w, 4, 1, fr); fwrite(&h, 4, 1, fr); fwrite(&c, 4, 1, fr);
        fwrite(d, 1, (size_t)w * h * c, fr);
        stbi_image_free(d);
        n++;
    }
    closedir(dir);
    fseek(fr, 0, SEEK_SET);
    fwrite(&n, 4, 1, fr);
    fclose(fr); fclose(fo);
    return 0;
}
