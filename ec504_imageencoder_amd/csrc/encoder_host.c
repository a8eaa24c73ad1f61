/* encoder_host.c — C host driver behind include/encoder.h: the folder-level control flow of the
 * reference's mpeg_encode_procedure (include/encoder.h:20-498 under /root/reference) around the HIP
 * hot path of include/mpeg1_hip.h.  Plain C; all device work goes through the m1v_* C-ABI.
 *
 * Kept from the reference (observable behaviour): the order of side effects (open video, write
 * PACK+SYS, create folders, scan, load, check), return codes, the ".jpg"/".jpeg" substring filter,
 * raw readdir order, the 256-byte path buffer, one frame record per image, image_<k>.bit side files.
 * Not kept: stdout chatter, leaks, loading through a fixed decoder (see encoder_set_image_loader).
 *
 * Environment:
 *   EC504_ENCODE_REGION=full|strict   region used by mpeg_encode_procedure()  (default strict)
 *   EC504_WRITE_BIT=0                 skip the image_<k>.bit side files         (default: write them)
 *   EC504_DEVICE=<n>                  GPU index                                (default 0)
 *   EC504_BATCH=<n>                   frames per device batch                  (default 64)
 */
#define _DEFAULT_SOURCE
#define EC504_NO_STB
#include "encoder.h"
#include "mpeg1_hip.h"

#include <errno.h>
#include <stdint.h>
#include <string.h>

static encoder_image_load_fn g_load = NULL;
static encoder_image_free_fn g_free = NULL;

void encoder_set_image_loader(encoder_image_load_fn load, encoder_image_free_fn release) {
    g_load = load;
    g_free = release;
}

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

typedef struct {
    Image *v;
    int n, cap;
} ImageList;

static void release_images(ImageList *l) {
    for (int i = 0; i < l->n; i++)
        if (l->v[i].data && g_free) g_free(l->v[i].data);
    free(l->v);
    l->v = NULL;
    l->n = l->cap = 0;
}

/* image_processing.c:753-787: int32 W, int32 H, then the Y, Cb, Cr planes (full resolution). */
static void write_bit_file(const char *folder, int k, const uint8_t *planes, int W, int H) {
    char path[256];
    snprintf(path, sizeof path, "%s/image_%d.bit", folder, k);
    FILE *f = fopen(path, "wb");
    if (!f) {
        printf("Error: Could not open bitstream file.\n");
        return;
    }
    int32_t w = W, h = H;
    fwrite(&w, sizeof w, 1, f);
    fwrite(&h, sizeof h, 1, f);
    fwrite(planes, 1, (size_t)W * H * 3, f);
    fclose(f);
}

int mpeg_encode_procedure_region(const char *images_folder, const char *bitstream_folder,
                                 const char *video_path, int quality_factor, int region) {
    FILE *fp = fopen(video_path, "wb"); /* encoder.h:75-80 */
    if (fp == NULL) {
        perror("Error opening mpeg file");
        return 1;
    }
    uint8_t prolog[27];
    m1v_file_prolog(prolog); /* encoder.h:86-89 */
    fwrite(prolog, 1, sizeof prolog, fp);

    struct stat st;
    if (stat(bitstream_folder, &st) == -1) { /* encoder.h:104-108 */
        mkdir(bitstream_folder, 0700);
        printf("Created directory for bitstreams: %s\n", bitstream_folder);
    }
    if (stat(images_folder, &st) == -1) { /* encoder.h:111-116 */
        mkdir(images_folder, 0700);
        printf("Created directory for images: %s\n", images_folder);
        printf("Please add your .jpg images in the '%s' folder and rerun the program.\n", images_folder);
        fclose(fp);
        return 0;
    }
    DIR *dir = opendir(images_folder); /* encoder.h:119-124 */
    if (!dir) {
        printf("Error: Could not open images directory.\n");
        fclose(fp);
        return -1;
    }
    if (!g_load) {
        printf("Error: no image loader registered (include encoder.h with stb_image.h on the include "
               "path, or call encoder_set_image_loader).\n");
        closedir(dir);
        fclose(fp);
        return -1;
    }

    ImageList imgs = {NULL, 0, 0};
    struct dirent *entry;
    char filepath[256];
    while ((entry = readdir(dir)) != NULL) { /* encoder.h:140-171 */
        if (strstr(entry->d_name, ".jpg") == NULL && strstr(entry->d_name, ".jpeg") == NULL) continue;
        if (imgs.n == imgs.cap) {
            int ncap = imgs.cap ? imgs.cap * 2 : 100;
            Image *nv = (Image *)realloc(imgs.v, (size_t)ncap * sizeof *nv);
            if (!nv) {
                printf("Error: Memory reallocation failed for images array.\n");
                closedir(dir);
                release_images(&imgs);
                fclose(fp);
                return -1;
            }
            imgs.v = nv;
            imgs.cap = ncap;
        }
        snprintf(filepath, sizeof filepath, "%s/%s", images_folder, entry->d_name);
        Image im;
        im.data = g_load(filepath, &im.width, &im.height, &im.channels, 0);
        if (!im.data) {
            printf("Error loading image %s\n", filepath);
            continue;
        }
        imgs.v[imgs.n++] = im;
    }
    closedir(dir);

    int rc = -1;
    m1v_encoder *enc = NULL;
    uint8_t *batch_in = NULL, *batch_out = NULL, *planes = NULL;
    uint64_t *sizes = NULL;

    if (imgs.n == 0) { /* check_dimensions, image_processing.c:48-66 */
        printf("No images found in directory.\n");
        goto done;
    }
    const int W = imgs.v[0].width, H = imgs.v[0].height, C = imgs.v[0].channels;
    for (int i = 1; i < imgs.n; i++) {
        if (imgs.v[i].width != W || imgs.v[i].height != H) {
            printf("Error: Image dimensions do not match\n");
            goto done;
        }
        if (imgs.v[i].channels != C) {
            printf("Error: Image channel counts do not match\n");
            goto done;
        }
    }
    if (C < 3) { /* image_processing.c:69-73 prints this and the reference then crashes */
        printf("Error: Image does not have correct color channels for RBG to YCbCr conversion.\n");
        goto done;
    }

    int batch = env_int("EC504_BATCH", 64);
    if (batch < 1) batch = 1;
    if (batch > imgs.n) batch = imgs.n;
    int mrc = m1v_create(&enc, env_int("EC504_DEVICE", 0), W, H, C, quality_factor,
                         region ? M1V_MODE_FULL : M1V_MODE_STRICT, batch);
    if (mrc != M1V_OK) {
        printf("Error: cannot set up the GPU encoder: %s\n", m1v_last_error());
        goto done;
    }
    const size_t frame_in = m1v_frame_bytes_in(enc), bound = m1v_frame_bound(enc);
    const int write_bit = env_int("EC504_WRITE_BIT", 1);
    /* pinned staging: the copies to and from the GPU then run at the PCIe rate */
    batch_in = (uint8_t *)m1v_alloc_host(frame_in * (size_t)batch);
    batch_out = (uint8_t *)m1v_alloc_host(bound * (size_t)batch);
    sizes = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)batch);
    if (write_bit) planes = (uint8_t *)m1v_alloc_host((size_t)W * H * 3 * (size_t)batch);
    if (!batch_in || !batch_out || !sizes || (write_bit && !planes)) {
        printf("Error: Memory allocation failed.\n");
        goto done;
    }

    for (int base = 0; base < imgs.n; base += batch) { /* frame loop, encoder.h:196-486 */
        int n = imgs.n - base < batch ? imgs.n - base : batch;
        for (int i = 0; i < n; i++) memcpy(batch_in + frame_in * (size_t)i, imgs.v[base + i].data, frame_in);
        long total = m1v_encode_host(enc, batch_in, n, base, batch_out, bound * (size_t)batch, sizes);
        if (total < 0) {
            printf("Error: GPU encode failed: %s\n", m1v_last_error());
            goto done;
        }
        fwrite(batch_out, 1, (size_t)total, fp);
        if (write_bit) { /* encoder.h:461-465 */
            if (m1v_convert_host(enc, batch_in, n, planes) != M1V_OK) {
                printf("Error: GPU colour conversion failed: %s\n", m1v_last_error());
                goto done;
            }
            for (int i = 0; i < n; i++)
                write_bit_file(bitstream_folder, base + i + 1, planes + (size_t)W * H * 3 * (size_t)i, W, H);
        }
    }
    printf("Image processing finished.\n");
    rc = 0;

done:
    m1v_free_host(batch_in);
    m1v_free_host(batch_out);
    free(sizes);
    m1v_free_host(planes);
    m1v_destroy(enc);
    release_images(&imgs);
    fclose(fp);
    return rc;
}

int mpeg_encode_procedure(const char *images_folder, const char *bitstream_folder,
                          const char *video_path, int quality_factor) {
    const char *r = getenv("EC504_ENCODE_REGION");
    int region = (r && (strcmp(r, "full") == 0 || strcmp(r, "FULL") == 0 || strcmp(r, "1") == 0)) ? 1 : 0;
    return mpeg_encode_procedure_region(images_folder, bitstream_folder, video_path, quality_factor, region);
}
