/* encoder_host.c — C host driver behind include/encoder.h: the folder-level control flow of the
 * reference's mpeg_encode_procedure (include/encoder.h:20-498 under /root/reference) around the HIP
 * hot path of include/mpeg1_hip.h.  Plain C; all device work goes through the m1v_* C-ABI.
 *
 * Kept from the reference (observable behaviour): the order of side effects (open video, write
 * PACK+SYS, create folders, scan, load, check), return codes, the ".jpg"/".jpeg" substring filter,
 * raw readdir order, the 256-byte path buffer, one frame record per image, image_<k>.bit side files.
 * Not kept: stdout chatter, leaks, loading through a fixed decoder (see encoder_set_image_loader),
 * doing the host work on one thread (see the worker pool below).
 *
 * Environment:
 *   EC504_ENCODE_REGION=full|strict   region used by mpeg_encode_procedure()  (default strict)
 *   EC504_WRITE_BIT=0                 skip the image_<k>.bit side files         (default: write them)
 *   EC504_DEVICE=<n>                  GPU index                                (default 0)
 *   EC504_BATCH=<n>                   frames per device batch                  (default 64)
 *   EC504_HOST_THREADS=<n>            decode / staging / .bit-writer threads   (default: online CPUs, <= 64;
 *                                     1 = everything on the calling thread, as the reference)
 */
#define _DEFAULT_SOURCE
#define EC504_NO_STB
#include "encoder.h"
#include "mpeg1_hip.h"

#include <errno.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>
#include <time.h>

static encoder_image_load_fn g_load = NULL;
static encoder_image_free_fn g_free = NULL;
static int g_threads = 0; /* 0 = one per online CPU */

void encoder_set_image_loader(encoder_image_load_fn load, encoder_image_free_fn release) {
    g_load = load;
    g_free = release;
}

void encoder_set_host_threads(int n) { g_threads = n < 0 ? 0 : n; }

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

typedef struct {
    Image *v;
    int n, cap;
} ImageList;

static void release_images(ImageList *l) {
    for (int i = 0; l->v && i < l->n; i++)
        if (l->v[i].data && g_free) g_free(l->v[i].data);
    free(l->v);
    l->v = NULL;
    l->n = l->cap = 0;
}

/* ---- host worker pool ---------------------------------------------------------
 * The reference decodes, converts and writes one image at a time on one thread (stb decode alone is
 * the largest share of its wall time after the debug printing).  Here three kinds of host work run on
 * a small pthread pool while the main thread drives the GPU: JPEG decode (one task per file),
 * staging of decoded pixels into pinned batch buffers, and the image_<k>.bit writes (write-behind).
 * A "group" is a parallel-for of n tasks; the main thread starts groups and joins them later, and
 * helps run tasks while it waits, so a pool of 1 thread is simply the serial program. */
typedef struct Group {
    void (*fn)(void *ctx, int i);
    void *ctx;
    int n, next, done;
    struct Group *link;
} Group;

typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t work, idle;
    Group *head, *tail;
    pthread_t *threads;
    int n_threads, quit;
} Pool;

/* with mu held: next runnable (group, index), or 0.  A group leaves the queue with its last index, so
 * once group_wait() has seen done == n nothing in the pool refers to it any more. */
static int pool_take(Pool *p, Group **g, int *i) {
    Group *h = p->head;
    if (!h) return 0;
    *g = h;
    *i = h->next++;
    if (h->next >= h->n) {
        p->head = h->link;
        if (!p->head) p->tail = NULL;
    }
    return 1;
}

static void pool_finish(Pool *p, Group *g) { /* with mu held */
    if (++g->done == g->n) pthread_cond_broadcast(&p->idle);
}

static void *pool_worker(void *arg) {
    Pool *p = (Pool *)arg;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        Group *g;
        int i;
        if (pool_take(p, &g, &i)) {
            pthread_mutex_unlock(&p->mu);
            g->fn(g->ctx, i);
            pthread_mutex_lock(&p->mu);
            pool_finish(p, g);
        } else if (p->quit) {
            break;
        } else {
            pthread_cond_wait(&p->work, &p->mu);
        }
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

static void pool_open(Pool *p, int threads) {
    memset(p, 0, sizeof *p);
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->work, NULL);
    pthread_cond_init(&p->idle, NULL);
    if (threads > 1) p->threads = (pthread_t *)calloc((size_t)threads - 1, sizeof(pthread_t));
    for (int t = 0; p->threads && t < threads - 1; t++) {
        if (pthread_create(&p->threads[p->n_threads], NULL, pool_worker, p) != 0) break;
        p->n_threads++;
    }
}

static void group_start(Pool *p, Group *g, void (*fn)(void *, int), void *ctx, int n) {
    g->fn = fn;
    g->ctx = ctx;
    g->n = n;
    g->next = g->done = 0;
    g->link = NULL;
    if (n <= 0) return;
    pthread_mutex_lock(&p->mu);
    if (p->tail) p->tail->link = g;
    else p->head = g;
    p->tail = g;
    pthread_cond_broadcast(&p->work);
    pthread_mutex_unlock(&p->mu);
}

/* Returns when every task of g has run — or, with a flag, as soon as *flag is set (under p->mu, with a
 * broadcast on p->idle); the caller runs queued tasks (of any group) meanwhile. */
static void group_wait_or(Pool *p, Group *g, const int *flag) {
    if (g->n <= 0) return;
    pthread_mutex_lock(&p->mu);
    while (g->done < g->n && !(flag && *flag)) {
        Group *h;
        int i;
        if (pool_take(p, &h, &i)) {
            pthread_mutex_unlock(&p->mu);
            h->fn(h->ctx, i);
            pthread_mutex_lock(&p->mu);
            pool_finish(p, h);
        } else {
            pthread_cond_wait(&p->idle, &p->mu);
        }
    }
    pthread_mutex_unlock(&p->mu);
}

static void group_wait(Pool *p, Group *g) { group_wait_or(p, g, NULL); }

static void pool_close(Pool *p) {
    pthread_mutex_lock(&p->mu);
    p->quit = 1;
    pthread_cond_broadcast(&p->work);
    pthread_mutex_unlock(&p->mu);
    for (int t = 0; t < p->n_threads; t++) pthread_join(p->threads[t], NULL);
    free(p->threads);
    pthread_cond_destroy(&p->idle);
    pthread_cond_destroy(&p->work);
    pthread_mutex_destroy(&p->mu);
}

/* ---- the three kinds of host task ------------------------------------------------------------- */
typedef struct {
    char **path;
    Image *img; /* img[i].data == NULL: the loader refused path[i] */
    Pool *pool;
    int have_first; /* set (under pool->mu) once any file has decoded; first = its geometry */
    Image first;
} DecodeJob;

static void decode_task(void *ctx, int i) { /* encoder.h:162 */
    DecodeJob *j = (DecodeJob *)ctx;
    Image *im = &j->img[i];
    im->data = g_load(j->path[i], &im->width, &im->height, &im->channels, 0);
    if (im->data) {
        pthread_mutex_lock(&j->pool->mu);
        if (!j->have_first) {
            j->first = *im;
            j->have_first = 1;
            pthread_cond_broadcast(&j->pool->idle);
        }
        pthread_mutex_unlock(&j->pool->mu);
    }
}

static void warm_task(void *ctx, int i) { /* start the GPU runtime while the first files decode */
    (void)ctx;
    (void)i;
    (void)m1v_device_count();
}

typedef struct {
    Image *img;    /* the batch's first image */
    uint8_t *dst;  /* pinned batch buffer */
    size_t frame_in;
} StageJob;

static void stage_task(void *ctx, int i) {
    StageJob *j = (StageJob *)ctx;
    memcpy(j->dst + j->frame_in * (size_t)i, j->img[i].data, j->frame_in);
    if (g_free) g_free(j->img[i].data); /* the decoded copy is not needed again */
    j->img[i].data = NULL;
}

typedef struct {
    const char *folder;
    const uint8_t *planes;
    int first_k, W, H;
} BitJob;

/* image_processing.c:753-787: int32 W, int32 H, then the Y, Cb, Cr planes (full resolution). */
static void bit_task(void *ctx, int i) {
    BitJob *j = (BitJob *)ctx;
    char path[256];
    snprintf(path, sizeof path, "%s/image_%d.bit", j->folder, j->first_k + i);
    FILE *f = fopen(path, "wb");
    if (!f) {
        printf("Error: Could not open bitstream file.\n");
        return;
    }
    int32_t w = j->W, h = j->H;
    fwrite(&w, sizeof w, 1, f);
    fwrite(&h, sizeof h, 1, f);
    fwrite(j->planes + (size_t)j->W * j->H * 3 * (size_t)i, 1, (size_t)j->W * j->H * 3, f);
    fclose(f);
}

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static int host_threads(void) {
    int n = env_int("EC504_HOST_THREADS", g_threads);
    if (n <= 0) {
        long cpus = sysconf(_SC_NPROCESSORS_ONLN);
        n = cpus < 1 ? 1 : (int)cpus;
    }
    return n > 64 ? 64 : n;
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

    /* encoder.h:140-171 in two steps: the directory scan fixes the frame order (raw readdir order, the
     * ".jpg"/".jpeg" substring filter, the 256-byte path buffer), then the files are decoded in parallel. */
    char **paths = NULL;
    int n_paths = 0, cap_paths = 0;
    struct dirent *entry;
    char filepath[256];
    while ((entry = readdir(dir)) != NULL) {
        if (strstr(entry->d_name, ".jpg") == NULL && strstr(entry->d_name, ".jpeg") == NULL) continue;
        if (n_paths == cap_paths) {
            int ncap = cap_paths ? cap_paths * 2 : 100;
            char **nv = (char **)realloc(paths, (size_t)ncap * sizeof *nv);
            if (!nv) break;
            paths = nv;
            cap_paths = ncap;
        }
        snprintf(filepath, sizeof filepath, "%s/%s", images_folder, entry->d_name);
        if (!(paths[n_paths] = strdup(filepath))) break;
        n_paths++;
    }
    int scan_failed = entry != NULL;
    closedir(dir);

    int rc = -1;
    const int timing = env_int("EC504_TIMING", 0); /* phase times on stderr */
    double t_phase[5] = {now_s(), 0, 0, 0, 0};
    Pool pool;
    pool_open(&pool, host_threads());
    ImageList imgs = {NULL, 0, 0};
    m1v_encoder *enc = NULL;
    uint8_t *batch_in[2] = {NULL, NULL}, *planes[2] = {NULL, NULL}, *batch_out = NULL;
    uint64_t *sizes = NULL;
    Group staged[2], written[2], warm, decoded;
    memset(staged, 0, sizeof staged);
    memset(written, 0, sizeof written);
    memset(&warm, 0, sizeof warm);
    memset(&decoded, 0, sizeof decoded);
    DecodeJob dj;
    memset(&dj, 0, sizeof dj);

    imgs.v = (Image *)calloc((size_t)(n_paths ? n_paths : 1), sizeof(Image));
    if (scan_failed || !imgs.v) {
        printf("Error: Memory reallocation failed for images array.\n");
        goto done;
    }
    imgs.cap = n_paths;
    dj.path = paths, dj.img = imgs.v, dj.pool = &pool;
    if (n_paths > 0 && pool.n_threads > 0) group_start(&pool, &warm, warm_task, NULL, 1);
    group_start(&pool, &decoded, decode_task, &dj, n_paths);
    group_wait_or(&pool, &decoded, &dj.have_first);

    /* As soon as one file has decoded its geometry is (unless the folder is inconsistent, which the checks
     * below reject exactly as before) the sequence's: set the GPU encoder and the pinned batch buffers up
     * on this thread while the pool decodes the rest. */
    const int write_bit = env_int("EC504_WRITE_BIT", 1);
    int batch = 0, mrc = M1V_OK, alloc_ok = 1;
    size_t frame_in = 0, bound = 0, out_cap = 0;
    if (dj.have_first && dj.first.channels >= 3) {
        frame_in = (size_t)dj.first.width * dj.first.height * dj.first.channels;
        /* default batch: about 96 MB of pixels, 4..64 frames (pinned memory costs time to get and to give back) */
        batch = env_int("EC504_BATCH", (int)((96u << 20) / (frame_in ? frame_in : 1)));
        if (!getenv("EC504_BATCH")) batch = batch < 4 ? 4 : batch > 64 ? 64 : batch;
        if (batch < 1) batch = 1;
        if (batch > n_paths) batch = n_paths;
        mrc = m1v_create(&enc, env_int("EC504_DEVICE", 0), dj.first.width, dj.first.height, dj.first.channels,
                         quality_factor, region ? M1V_MODE_FULL : M1V_MODE_STRICT, batch);
        if (mrc == M1V_OK) {
            bound = m1v_frame_bound(enc);
            /* pinned staging (copies to and from the GPU then run at the PCIe rate), two slots: while the GPU
             * works on one, the pool fills the other and drains the previous planes into image_<k>.bit files.
             * The output buffer starts at 1/16 of the worst case (white noise needs about 1/46 of it, pictures
             * built to be expensive about 1/11) and grows on demand (see the batch loop). */
            out_cap = bound * ((size_t)batch + 1) / 16;
            batch_out = (uint8_t *)m1v_alloc_host(out_cap);
            sizes = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)batch);
            alloc_ok = batch_out && sizes;
            for (int s = 0; s < (batch < n_paths ? 2 : 1); s++) {
                batch_in[s] = (uint8_t *)m1v_alloc_host(frame_in * (size_t)batch);
                if (write_bit) planes[s] = (uint8_t *)m1v_alloc_host((size_t)dj.first.width * dj.first.height * 3 * (size_t)batch);
                alloc_ok = alloc_ok && batch_in[s] && (!write_bit || planes[s]);
            }
        }
    }
    t_phase[1] = now_s();
    group_wait(&pool, &decoded);
    t_phase[2] = now_s();

    for (int i = 0; i < n_paths; i++) { /* unloadable files are reported and skipped, encoder.h:163-167 */
        if (!imgs.v[i].data) {
            printf("Error loading image %s\n", paths[i]);
            continue;
        }
        imgs.v[imgs.n++] = imgs.v[i];
    }
    for (int i = imgs.n; i < n_paths; i++) imgs.v[i].data = NULL;

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
    if (mrc != M1V_OK || !enc) {
        printf("Error: cannot set up the GPU encoder: %s\n", m1v_last_error());
        goto done;
    }
    if (!alloc_ok) {
        printf("Error: Memory allocation failed.\n");
        goto done;
    }
    if (batch > imgs.n) batch = imgs.n;
    const int n_batches = (imgs.n + batch - 1) / batch;
    const int n_slots = n_batches > 1 ? 2 : 1;

    StageJob sj[2];
    BitJob bj[2];
    sj[0] = (StageJob){imgs.v, batch_in[0], frame_in};
    group_start(&pool, &staged[0], stage_task, &sj[0], batch);
    for (int b = 0; b < n_batches; b++) { /* frame loop, encoder.h:196-486, one batch per pass */
        const int s = b % n_slots, base = b * batch;
        const int n = imgs.n - base < batch ? imgs.n - base : batch;
        group_wait(&pool, &staged[s]);
        if (b + 1 < n_batches) { /* the pool stages the next batch behind the GPU call below */
            const int nb = base + batch, nn = imgs.n - nb < batch ? imgs.n - nb : batch;
            sj[1 - s] = (StageJob){imgs.v + nb, batch_in[1 - s], frame_in};
            group_start(&pool, &staged[1 - s], stage_task, &sj[1 - s], nn);
        }
        group_wait(&pool, &written[s]); /* planes[s] may still be on their way to disk (batch b-2) */
        long total = m1v_encode_planes_host(enc, batch_in[s], n, base, batch_out, out_cap, sizes,
                                            write_bit ? planes[s] : NULL);
        if (total == M1V_E_NOSPACE && out_cap < bound * (size_t)batch) { /* rare: grow to the worst case, redo */
            m1v_free_host(batch_out);
            out_cap = bound * (size_t)batch;
            batch_out = (uint8_t *)m1v_alloc_host(out_cap);
            total = batch_out ? m1v_encode_planes_host(enc, batch_in[s], n, base, batch_out, out_cap, sizes,
                                                       write_bit ? planes[s] : NULL)
                              : M1V_E_HIP;
        }
        if (total < 0) {
            printf("Error: GPU encode failed: %s\n", m1v_last_error());
            goto done;
        }
        fwrite(batch_out, 1, (size_t)total, fp);
        if (write_bit) { /* encoder.h:461-465, written behind the next batch */
            bj[s] = (BitJob){bitstream_folder, planes[s], base + 1, W, H};
            group_start(&pool, &written[s], bit_task, &bj[s], n);
        }
    }
    printf("Image processing finished.\n");
    rc = 0;
    t_phase[3] = now_s();

done:
    group_wait(&pool, &warm); /* nothing may still reference what is freed below */
    group_wait(&pool, &decoded);
    for (int s = 0; s < 2; s++) {
        group_wait(&pool, &staged[s]);
        group_wait(&pool, &written[s]);
    }
    t_phase[4] = now_s();
    if (timing && rc == 0)
        fprintf(stderr, "ec504 timing: %d files, %d threads, batch %d: first decode + gpu setup + pinned buffers %.3f s, "
                        "rest of decode %.3f s, batches %.3f s, last .bit writes %.3f s\n", imgs.n, pool.n_threads + 1,
                batch, t_phase[1] - t_phase[0], t_phase[2] - t_phase[1], t_phase[3] - t_phase[2], t_phase[4] - t_phase[3]);
    pool_close(&pool);
    for (int s = 0; s < 2; s++) {
        m1v_free_host(batch_in[s]);
        m1v_free_host(planes[s]);
    }
    m1v_free_host(batch_out);
    free(sizes);
    m1v_destroy(enc);
    imgs.n = n_paths; /* every slot that still holds pixels */
    release_images(&imgs);
    for (int i = 0; i < n_paths; i++) free(paths[i]);
    free(paths);
    fclose(fp);
    return rc;
}

int mpeg_encode_procedure(const char *images_folder, const char *bitstream_folder,
                          const char *video_path, int quality_factor) {
    const char *r = getenv("EC504_ENCODE_REGION");
    int region = (r && (strcmp(r, "full") == 0 || strcmp(r, "FULL") == 0 || strcmp(r, "1") == 0)) ? 1 : 0;
    return mpeg_encode_procedure_region(images_folder, bitstream_folder, video_path, quality_factor, region);
}
