/* include/encoder.h — drop-in replacement for the reference's include/encoder.h.
 *
 * In the reference this header DEFINES mpeg_encode_procedure() (include/encoder.h:20-498 under
 * /root/reference) so that it is compiled into every caller (main.c:10, encoder_jni.c:2).  Here the
 * header only DECLARES it; the implementation lives in libencoder.so (C host driver
 * ec504_imageencoder_amd/csrc/encoder_host.c on top of the HIP kernels behind include/mpeg1_hip.h).
 * main.c:16 and encoder_jni.c:14 compile unchanged against this header and link -lencoder.
 *
 * JPEG decoding stays where the reference put it: in the CALLER's translation unit, through the
 * single-header stb_image.h (reference include/stb_image.h, third-party, public domain / MIT), which
 * the reference's header pulls in under STB_IMAGE_IMPLEMENTATION (encoder.h:9-10).  When stb_image.h
 * is on the caller's include path this header does the same and registers stbi_load /
 * stbi_image_free with the library before main() runs, so the library sees exactly the pixels the
 * reference sees.  Without it, callers register any loader with encoder_set_image_loader().
 *
 * Return values of mpeg_encode_procedure (encoder.h:77-80,111-116,121-124,175-183,497):
 *    1  video_path cannot be opened for writing
 *    0  success; also 0 (and nothing encoded) when images_folder did not exist and was created
 *   -1  images_folder cannot be read, no image loaded, dimensions differ
 *       (and, instead of the reference's undefined behaviour: fewer than 3 channels, a picture smaller
 *       than the encoded region, a coefficient the reference's VLC cannot code, no GPU / no loader)
 */
#ifndef EC504_ENCODER_H
#define EC504_ENCODER_H

#include <stdio.h>
#include <stdlib.h>
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>
#include <math.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference include/jpeg_handler.h:6-11 */
#ifndef JPEG_HANDLER_H
#define JPEG_HANDLER_H
typedef struct {
    int width;
    int height;
    int channels;
    unsigned char *data;
} Image;
#endif

/* reference include/encoder.h:20 */
int mpeg_encode_procedure(const char *images_folder, const char *bitstream_folder,
                          const char *video_path, int quality_factor);

/* Same, with the macroblock region chosen explicitly: 0 = the 96x144 region of the unmodified
 * reference (encoder.h:238,248), 1 = every macroblock.  mpeg_encode_procedure() uses region 0 unless
 * the environment says EC504_ENCODE_REGION=full. */
int mpeg_encode_procedure_region(const char *images_folder, const char *bitstream_folder,
                                 const char *video_path, int quality_factor, int region);

/* Image loader used for every directory entry whose name contains ".jpg" or ".jpeg"
 * (encoder.h:141,162).  Same contract as stbi_load(path,&w,&h,&channels,0) / stbi_image_free. */
typedef unsigned char *(*encoder_image_load_fn)(char const *path, int *w, int *h, int *channels, int desired);
typedef void (*encoder_image_free_fn)(void *pixels);
void encoder_set_image_loader(encoder_image_load_fn load, encoder_image_free_fn release);

/* Host threads used for decoding (the loader is called from several threads at once: stbi_load is
 * re-entrant), for staging pixels into pinned memory and for writing the image_<k>.bit files behind
 * the GPU.  0 = one per online CPU (default), 1 = everything on the calling thread like the reference.
 * The environment variable EC504_HOST_THREADS overrides it. */
void encoder_set_host_threads(int n);

/* mpeg_encode_procedure keeps its GPU encoders and pinned buffers for the next call with the same picture geometry,
 * quality factor, region and device list (environment EC504_KEEP_ENCODER=0 turns that off).  This frees them. */
void encoder_release_cache(void);

#ifdef __cplusplus
}
#endif

/* JPEG decoder in the caller's TU, as in the reference (encoder.h:9-10). */
#if !defined(EC504_NO_STB) && defined(__has_include)
#if __has_include("stb_image.h")
#ifndef STB_IMAGE_IMPLEMENTATION
#define STB_IMAGE_IMPLEMENTATION
#endif
#include "stb_image.h"
__attribute__((constructor)) static void ec504_register_stb_loader(void) {
    encoder_set_image_loader(stbi_load, stbi_image_free);
}
#endif
#endif

#endif /* EC504_ENCODER_H */
