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
 *   EC504_DEVICES=<a,b,...>           one GPU encoder per entry, chunk c goes to entry c mod N; the same index may
 *                                     appear more than once (default "<d>,<d>", d = EC504_DEVICE: two encoders on one
 *                                     GPU, so that one chunk's upload runs under the other's kernels and downloads)
 *   EC504_KEEP_ENCODER=0              do not keep the GPU encoders and pinned buffers for the next call (default: keep ONE
 *                                     set — the last call's; a call that asks for something else frees it first;
 *                                     encoder_release_cache() and process exit free it)
 *   EC504_BATCH=<n>                   frames per device batch                  (default 64)
 *   EC504_HOST_THREADS=<n>            decode / staging / .bit-writer threads   (default: the CPUs this process may run
 *                                     on or 1.25 x its cgroup CPU quota, whichever is smaller, at most 64; an explicit
 *                                     value up to 512 is taken as given; 1 = everything on the calling thread, as the
 *                                     reference)
 *   EC504_EXTRA_SLOTS=<n>             pinned staging slots beyond one per encoder (default 1; each holds one chunk of
 *                                     pixels and, with .bit files, one chunk of planes)
 */
#define _GNU_SOURCE
#define _DEFAULT_SOURCE
#define EC504_NO_STB
#include "encoder.h"
#include "mpeg1_hip.h"

#include <errno.h>
#include <pthread.h>
#include <sched.h>
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

/* with mu held: the next index of group g itself, if g still has unclaimed tasks in the queue */
static int pool_take_from(Pool *p, Group *g, int *i) {
    Group *prev = NULL;
    for (Group *h = p->head; h; prev = h, h = h->link) {
        if (h != g) continue;
        *i = h->next++;
        if (h->next >= h->n) {
            if (prev) prev->link = h->link;
            else p->head = h->link;
            if (p->tail == h) p->tail = prev;
        }
        return 1;
    }
    return 0;
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

/* The same, in FRONT of everything queued (the set-up work a whole pipeline waits for must not queue behind 200 decodes). */
static void group_start_front(Pool *p, Group *g, void (*fn)(void *, int), void *ctx, int n) {
    g->fn = fn;
    g->ctx = ctx;
    g->n = n;
    g->next = g->done = 0;
    g->link = NULL;
    if (n <= 0) return;
    pthread_mutex_lock(&p->mu);
    g->link = p->head;
    p->head = g;
    if (!p->tail) p->tail = g;
    pthread_cond_broadcast(&p->work);
    pthread_mutex_unlock(&p->mu);
}

/* Returns when every task of g has run — or, with a flag, as soon as *flag is set (under p->mu, with a
 * broadcast on p->idle).  Meanwhile the caller runs tasks of g itself; tasks of OTHER groups only when the pool has no
 * thread of its own (EC504_HOST_THREADS=1: the serial program) — a caller that picks up a 12-ms decode while it waits for a
 * 1-ms staging copy paces the whole pipeline at one decode per chunk (measured: 14 ms per chunk whatever the thread count). */
static void group_wait_or(Pool *p, Group *g, const int *flag) {
    if (g->n <= 0) return;
    pthread_mutex_lock(&p->mu);
    while (g->done < g->n && !(flag && *flag)) {
        Group *h = g;
        int i;
        if (pool_take_from(p, g, &i) || (p->n_threads == 0 && pool_take(p, &h, &i))) {
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
static int g_task_timing = 0;              /* EC504_TIMING: sum the tasks' own durations (microseconds) and count them */
static long long g_task_us[4], g_task_n[4]; /* decode, stage, .bit write, GPU */
static double now_s(void);
#define TASK_TIMED(kind, stmt)                                                     \
    do {                                                                           \
        if (!g_task_timing) {                                                      \
            stmt;                                                                  \
        } else {                                                                   \
            const double t0_ = now_s();                                            \
            stmt;                                                                  \
            __atomic_fetch_add(&g_task_us[kind], (long long)((now_s() - t0_) * 1e6), __ATOMIC_RELAXED); \
            __atomic_fetch_add(&g_task_n[kind], 1, __ATOMIC_RELAXED);              \
        }                                                                          \
    } while (0)

typedef struct {
    char **path; /* the chunk's first file */
    Image *img;  /* img[i].data == NULL afterwards: the loader refused path[i] */
} DecodeJob;

static void decode_task(void *ctx, int i) { /* encoder.h:162 */
    DecodeJob *j = (DecodeJob *)ctx;
    Image *im = &j->img[i];
    if (im->data) return; /* the head file: decoded before the pipeline started */
    TASK_TIMED(0, im->data = g_load(j->path[i], &im->width, &im->height, &im->channels, 0));
}

static void warm_task(void *ctx, int i) { /* start the GPU runtime (device i of the list) while the first files decode */
    (void)m1v_warm_up(((const int *)ctx)[i]);
}

typedef struct {
    Image *img;    /* the batch's first image */
    uint8_t *dst;  /* pinned batch buffer */
    size_t frame_in;
} StageJob;

typedef struct {
    void **dst;
    size_t bytes;
} AllocItem;

static void alloc_task(void *ctx, int i) { /* one pinned buffer (page-locking ~100 MB takes tens of milliseconds: in parallel) */
    AllocItem *a = (AllocItem *)ctx + i;
    *a->dst = m1v_alloc_host(a->bytes);
}

static void stage_task(void *ctx, int i) {
    StageJob *j = (StageJob *)ctx;
    TASK_TIMED(1, {
        memcpy(j->dst + j->frame_in * (size_t)i, j->img[i].data, j->frame_in);
        if (g_free) g_free(j->img[i].data); /* the decoded copy is not needed again */
    });
    j->img[i].data = NULL;
}

typedef struct {
    const char *folder;
    const uint8_t *planes;
    int first_k, W, H;
} BitJob;

/* image_processing.c:753-787: int32 W, int32 H, then the Y, Cb, Cr planes (full resolution). */
static void bit_write(const BitJob *j, int i);
static void bit_task(void *ctx, int i) { TASK_TIMED(2, bit_write((const BitJob *)ctx, i)); }
static void bit_write(const BitJob *j, int i) {
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

static void remove_bit_file(const char *folder, int k) {
    char path[256];
    snprintf(path, sizeof path, "%s/image_%d.bit", folder, k);
    (void)remove(path);
}

/* ---- several GPU encoders in flight (SURVEY 8e inside the C library) -------------------------------
 * A LANE is one m1v_encoder with its own pinned output buffer.  Chunk c is encoded by lane c mod N as a pool task, so N
 * chunks are on GPUs at once — on N different GPUs (frames are independent given their global index: the chunk's first
 * frame index travels with it), or twice on the same GPU, where the upload of one chunk overlaps the kernels and the
 * downloads of the other.  The caller retires chunks in order and appends their frame records to the video: the file is
 * the gather. */
enum { MAX_LANES = 16, MAX_SLOTS = MAX_LANES + 8 };

typedef struct {
    m1v_encoder *enc;
    uint8_t *out;      /* pinned */
    size_t out_cap;
    uint64_t *sizes;
    /* the chunk in flight */
    const uint8_t *in;
    uint8_t *planes;
    int n, first;
    long total;        /* result of m1v_encode_planes_host */
    char err[256];     /* m1v_last_error() of the thread that ran the task (it is per thread) */
} Lane;

static void gpu_task(void *ctx, int i) {
    Lane *l = (Lane *)ctx;
    (void)i;
    TASK_TIMED(3, l->total = m1v_encode_planes_host(l->enc, l->in, l->n, l->first, l->out, l->out_cap, l->sizes, l->planes));
    if (l->total < 0) snprintf(l->err, sizeof l->err, "%s", m1v_last_error());
}

/* Everything a call needs on the device side, kept between calls when the next call asks for the same thing: creating
 * the encoders and page-locking ~100 MB per slot costs more than encoding a small folder. */
typedef struct {
    int valid;
    int devices[MAX_LANES], n_lanes, W, H, C, qf, region, batch, write_bit, n_slots;
    Lane lane[MAX_LANES];
    uint8_t *batch_in[MAX_SLOTS], *planes[MAX_SLOTS];
} GpuContext;

static GpuContext g_cache;
static pthread_mutex_t g_cache_mu = PTHREAD_MUTEX_INITIALIZER;

static void context_free(GpuContext *c) {
    for (int l = 0; l < c->n_lanes; l++) {
        m1v_destroy(c->lane[l].enc);
        m1v_free_host(c->lane[l].out);
        free(c->lane[l].sizes);
    }
    for (int s = 0; s < c->n_slots; s++) {
        m1v_free_host(c->batch_in[s]);
        m1v_free_host(c->planes[s]);
    }
    memset(c, 0, sizeof *c);
}

void encoder_release_cache(void) {
    pthread_mutex_lock(&g_cache_mu);
    if (g_cache.valid) context_free(&g_cache);
    pthread_mutex_unlock(&g_cache_mu);
}

static int context_matches(const GpuContext *c, const GpuContext *want) {
    if (!c->valid || c->n_lanes != want->n_lanes || c->W != want->W || c->H != want->H || c->C != want->C || c->qf != want->qf ||
        c->region != want->region || c->batch < want->batch || c->write_bit < want->write_bit || c->n_slots < want->n_slots)
        return 0;
    for (int l = 0; l < c->n_lanes; l++)
        if (c->devices[l] != want->devices[l]) return 0;
    return 1;
}

/* EC504_DEVICES, or two lanes on EC504_DEVICE */
static int parse_devices(int devices[MAX_LANES]) {
    const char *v = getenv("EC504_DEVICES");
    int n = 0;
    if (v && *v) {
        while (*v && n < MAX_LANES) {
            char *end;
            long d = strtol(v, &end, 10);
            if (end == v) break;
            devices[n++] = (int)d;
            v = *end == ',' ? end + 1 : end;
        }
    }
    if (n == 0) {
        devices[0] = devices[1] = env_int("EC504_DEVICE", 0);
        n = 2;
    }
    return n;
}

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* CPUs' worth of time the process may use per period: the cgroup's quota (v2 cpu.max, v1 cpu.cfs_quota_us), or 0 = unlimited */
static int cgroup_cpu_quota(void) {
    long quota = -1, period = 100000;
    FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (f) {
        char q[32];
        if (fscanf(f, "%31s %ld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atol(q);
        fclose(f);
    } else if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) != NULL) {
        if (fscanf(f, "%ld", &quota) != 1) quota = -1;
        fclose(f);
        if ((f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) != NULL) {
            if (fscanf(f, "%ld", &period) != 1) period = 100000;
            fclose(f);
        }
    }
    if (quota <= 0 || period <= 0) return 0;
    return (int)((quota + period - 1) / period);
}

/* Default: what the process can actually run at once — the CPUs it may be scheduled on, or its cgroup's CPU quota when that
 * is smaller (+25 %: some tasks wait for the GPU or the file system) — and at most 64.  Measured on the 256-CPU host with a
 * 16-CPU quota (profiles/r04_cli_phase_times.txt): a decode takes 10.5 ms alone, 11 ms among 16 threads, 40 ms among 64 (the
 * quota throttles them); 128 or 256 threads also slow the GPU runtime's start-up.  EC504_HOST_THREADS / encoder_set_host_threads
 * are taken as given (up to 512). */
static int host_threads(void) {
    int n = env_int("EC504_HOST_THREADS", g_threads);
    if (n <= 0) {
        cpu_set_t set;
        long cpus = sched_getaffinity(0, sizeof set, &set) == 0 ? (long)CPU_COUNT(&set) : sysconf(_SC_NPROCESSORS_ONLN);
        const int quota = cgroup_cpu_quota();
        n = cpus < 1 ? 1 : (int)cpus;
        if (quota > 0 && quota + quota / 4 < n) n = quota + quota / 4;
        if (n > 64) n = 64;
    }
    return n > 512 ? 512 : n;
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

    /* The frame loop (encoder.h:196-486) as a pipeline over CHUNKS of `batch` consecutive files:
     *
     *     pool:   decode chunks c+1 .. c+look | stage chunk c+1 into pinned slot | write chunk c-1's .bit files
     *     caller: chunk c through the GPU (one upload, frame records + planes back), append to the video
     *
     * so decoded pixels in flight are bounded by a few chunks whatever the folder holds.  The reference
     * loads and checks EVERYTHING before it encodes (encoder.h:140-183); a folder it would reject is
     * recognised here when the offending file is reached, and whatever was written until then is taken back
     * (video truncated to its 27-byte prolog, side files removed), which is the state the reference leaves. */
    enum { RING = 48 }; /* decode groups in flight, at most RING-1 chunks ahead of the GPU */
    int rc = -1, mismatch = 0;
    const int timing = env_int("EC504_TIMING", 0); /* phase times on stderr */
    g_task_timing = timing;
    memset(g_task_us, 0, sizeof g_task_us);
    memset(g_task_n, 0, sizeof g_task_n);
    double t_phase[5] = {now_s(), 0, 0, 0, 0}, t_created = 0;
    double t_wait[5] = {0, 0, 0, 0, 0}, t_mark; /* the caller's waits inside the chunk loop: staged, decoded, .bit writers, GPU, fwrite */
#define TIMED(k, stmt) do { t_mark = now_s(); stmt; t_wait[k] += now_s() - t_mark; } while (0)
    Pool pool;
    pool_open(&pool, host_threads());
    ImageList imgs = {NULL, 0, 0};
    GpuContext cx;
    memset(&cx, 0, sizeof cx);
    Group staged[MAX_SLOTS], written[MAX_SLOTS], on_gpu[MAX_LANES], decoded[RING], warm;
    DecodeJob dj[RING];
    StageJob sj[MAX_SLOTS];
    BitJob bj[MAX_SLOTS];
    memset(staged, 0, sizeof staged);
    memset(written, 0, sizeof written);
    memset(on_gpu, 0, sizeof on_gpu);
    memset(decoded, 0, sizeof decoded);
    memset(&warm, 0, sizeof warm);
    int frames_done = 0; /* frames retired so far = frame records in the video = .bit files written (or being written) */
    int *chunk_frames = NULL, *chunk_first = NULL; /* loaded frames of a chunk, global index of its first frame */
    int warm_devices[MAX_LANES], n_warm = 0;

    imgs.v = (Image *)calloc((size_t)(n_paths ? n_paths : 1), sizeof(Image));
    if (scan_failed || !imgs.v) {
        printf("Error: Memory reallocation failed for images array.\n");
        goto done;
    }
    imgs.n = imgs.cap = n_paths;
    cx.n_lanes = parse_devices(cx.devices);
    for (int l = 0; l < cx.n_lanes; l++) { /* the distinct devices: their runtimes start while the first files decode */
        int seen = 0;
        for (int k = 0; k < n_warm; k++) seen |= warm_devices[k] == cx.devices[l];
        if (!seen) warm_devices[n_warm++] = cx.devices[l];
    }
    if (n_paths > 0 && pool.n_threads > 0) group_start(&pool, &warm, warm_task, warm_devices, n_warm);

    /* head: the first file that decodes fixes the geometry (and with it the batch size) */
    int head = 0;
    for (; head < n_paths; head++) {
        Image *im = &imgs.v[head];
        im->data = g_load(paths[head], &im->width, &im->height, &im->channels, 0);
        if (im->data) break;
        printf("Error loading image %s\n", paths[head]); /* reported and skipped, encoder.h:163-167 */
    }
    if (head == n_paths) { /* check_dimensions, image_processing.c:48-66 */
        printf("No images found in directory.\n");
        goto done;
    }
    const int W = imgs.v[head].width, H = imgs.v[head].height, C = imgs.v[head].channels;
    const size_t frame_in = (size_t)W * H * C, frame_planes = (size_t)W * H * 3;
    /* default batch: about 96 MB of pixels, 4..64 frames (pinned memory costs time to get and to give back) */
    int batch = env_int("EC504_BATCH", (int)((96u << 20) / (frame_in ? frame_in : 1)));
    if (!getenv("EC504_BATCH")) batch = batch < 4 ? 4 : batch > 64 ? 64 : batch;
    if (batch < 1) batch = 1;
    if (batch > n_paths - head) batch = n_paths - head;
    const int n_chunks = (n_paths - head + batch - 1) / batch;
    if (cx.n_lanes > n_chunks) cx.n_lanes = n_chunks; /* no more encoders than chunks */
    int N = cx.n_lanes; /* (may still shrink: a lane whose encoder cannot be created is dropped) */
    /* enough chunks of decoding in flight to keep every thread busy (1.5 files per thread), 2..47 */
    int look = (3 * (pool.n_threads + 1) / 2 + batch - 1) / batch;
    look = look < N + 1 ? N + 1 : look;
    look = look > RING - 1 ? RING - 1 : look;
#define CHUNK_FIRST(c) (head + (c) * batch)
#define CHUNK_FILES(c) (n_paths - CHUNK_FIRST(c) < batch ? n_paths - CHUNK_FIRST(c) : batch)
    int next_decode = 0;
    for (; next_decode < n_chunks && next_decode < look; next_decode++) {
        dj[next_decode] = (DecodeJob){paths + CHUNK_FIRST(next_decode), imgs.v + CHUNK_FIRST(next_decode)};
        group_start(&pool, &decoded[next_decode], decode_task, &dj[next_decode], CHUNK_FILES(next_decode));
    }
    t_phase[1] = now_s();

    /* GPU encoders and pinned buffers, on this thread, while the pool decodes: taken over from the previous call when it
     * asked for the same thing */
    if (C < 3) { /* image_processing.c:69-73 prints this and the reference then crashes */
        printf("Error: Image does not have correct color channels for RBG to YCbCr conversion.\n");
        goto done;
    }
    const int write_bit = env_int("EC504_WRITE_BIT", 1);
    cx.W = W, cx.H = H, cx.C = C, cx.qf = quality_factor, cx.region = region, cx.batch = batch, cx.write_bit = write_bit;
    {   /* chunks on the lanes + those being staged or still feeding their .bit writers (EC504_EXTRA_SLOTS, default 1) */
        int extra = env_int("EC504_EXTRA_SLOTS", 1);
        extra = extra < 1 ? 1 : (extra > MAX_SLOTS - N ? MAX_SLOTS - N : extra);
        cx.n_slots = n_chunks > N ? (N + extra > n_chunks ? n_chunks : N + extra) : N;
    }
    chunk_frames = (int *)calloc((size_t)n_chunks + 1, sizeof(int));
    chunk_first = (int *)calloc((size_t)n_chunks + 1, sizeof(int));
    int alloc_ok = chunk_frames && chunk_first;
    pthread_mutex_lock(&g_cache_mu);
    if (context_matches(&g_cache, &cx)) {
        cx = g_cache; /* (batch, slots and planes may be larger than asked for) */
        memset(&g_cache, 0, sizeof g_cache);
    } else if (g_cache.valid) {
        context_free(&g_cache); /* one set at a time: the old encoders and pinned buffers go before the new ones come */
    }
    pthread_mutex_unlock(&g_cache_mu);
    size_t bound = 0;
    if (!cx.valid) {
        /* pinned staging (copies to and from the GPU then run at the PCIe rate), page-locked by the pool IN FRONT of the
         * queued decodes while this thread creates the encoders.  A lane's output buffer starts at 1/16 of the worst case
         * (white noise needs about 1/46 of it, pictures built to be expensive about 1/11) and grows on demand (see below). */
        AllocItem items[MAX_LANES + 2 * MAX_SLOTS];
        int n_items = 0;
        Group allocs;
        memset(&allocs, 0, sizeof allocs);
        bound = m1v_frame_bound_for(W, H, region ? M1V_MODE_FULL : M1V_MODE_STRICT);
        for (int l = 0; l < N; l++) {
            cx.lane[l].out_cap = bound * ((size_t)batch + 1) / 16;
            items[n_items++] = (AllocItem){(void **)&cx.lane[l].out, cx.lane[l].out_cap};
        }
        for (int s = 0; s < cx.n_slots; s++) {
            items[n_items++] = (AllocItem){(void **)&cx.batch_in[s], frame_in * (size_t)batch};
            if (write_bit) items[n_items++] = (AllocItem){(void **)&cx.planes[s], frame_planes * (size_t)batch};
        }
        group_wait(&pool, &warm); /* (the runtime of every device in the list is up: page-locking needs one) */
        group_start_front(&pool, &allocs, alloc_task, items, n_items);
        cx.valid = 1; /* from here on: whatever exists is freed by context_free */
        for (int l = 0; l < N; l++) {
            cx.lane[l].sizes = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)batch);
            if (m1v_create(&cx.lane[l].enc, cx.devices[l], W, H, C, quality_factor, region ? M1V_MODE_FULL : M1V_MODE_STRICT,
                           batch) != M1V_OK) {
                cx.lane[l].enc = NULL;
                if (l == 0) {
                    printf("Error: cannot set up the GPU encoder: %s\n", m1v_last_error());
                    alloc_ok = 0;
                } else {
                    printf("Note: encoder %d of %d could not be created (%s): continuing with %d\n", l + 1, N, m1v_last_error(), l);
                }
                group_wait(&pool, &allocs);
                for (int k = l; k < N; k++) { /* the lanes that will not run give their buffers back */
                    m1v_free_host(cx.lane[k].out);
                    free(cx.lane[k].sizes);
                    cx.lane[k].out = NULL;
                    cx.lane[k].sizes = NULL;
                }
                cx.n_lanes = N = l;
                break;
            }
        }
        t_created = now_s();
        group_wait(&pool, &allocs);
        if (N == 0) goto done;
        for (int l = 0; l < N; l++) alloc_ok = alloc_ok && cx.lane[l].out && cx.lane[l].sizes;
        for (int s = 0; s < cx.n_slots; s++) alloc_ok = alloc_ok && cx.batch_in[s] && (!write_bit || cx.planes[s]);
    } else {
        t_created = now_s();
        bound = m1v_frame_bound(cx.lane[0].enc);
    }
    if (!alloc_ok) {
        printf("Error: Memory allocation failed.\n");
        goto done;
    }
    const int S = cx.n_slots;
    t_phase[2] = now_s();

    /* Iteration c: chunk c+1 is joined from its decode and handed to the pool for staging; chunk c (staged) goes to its
     * lane; chunk c-N+1 is retired: its records are appended to the video, its .bit files handed to the pool. */
    for (int c = -1; c < n_chunks + N - 1; c++) {
        if (c >= 0 && c < n_chunks) TIMED(0, group_wait(&pool, &staged[c % S]));
        if (c >= 0 && next_decode < n_chunks) { /* keep `look` chunks of decoding ahead of the GPU */
            const int r = next_decode % RING; /* ring entry of a chunk <= c-1, joined at least one pass ago */
            dj[r] = (DecodeJob){paths + CHUNK_FIRST(next_decode), imgs.v + CHUNK_FIRST(next_decode)};
            group_start(&pool, &decoded[r], decode_task, &dj[r], CHUNK_FILES(next_decode));
            next_decode++;
        }
        if (c + 1 < n_chunks) { /* join chunk c+1's decode, check it, let the pool stage it */
            const int first = CHUNK_FIRST(c + 1), files = CHUNK_FILES(c + 1), slot = (c + 1) % S;
            Image *v = imgs.v + first;
            int n_next = 0;
            TIMED(1, group_wait(&pool, &decoded[(c + 1) % RING]));
            for (int i = 0; i < files; i++) {
                if (!v[i].data) {
                    printf("Error loading image %s\n", paths[first + i]);
                    continue;
                }
                if (v[i].width != W || v[i].height != H) {
                    if (!mismatch) printf("Error: Image dimensions do not match\n");
                    mismatch = 1;
                } else if (v[i].channels != C) {
                    if (!mismatch) printf("Error: Image channel counts do not match\n");
                    mismatch = 1;
                }
                v[n_next++] = v[i];
            }
            for (int i = n_next; i < files; i++) v[i].data = NULL;
            if (mismatch) goto done;
            chunk_frames[c + 1] = n_next;
            chunk_first[c + 1] = c >= 0 ? chunk_first[c] + chunk_frames[c] : 0;
            /* the slot's previous chunk (c+1-S) left its lane at least one pass ago; its .bit files may still be on their way */
            TIMED(2, group_wait(&pool, &written[slot]));
            sj[slot] = (StageJob){v, cx.batch_in[slot], frame_in};
            group_start_front(&pool, &staged[slot], stage_task, &sj[slot], n_next); /* in front of the decodes that run ahead */
        }
        if (c >= 0 && c < n_chunks && chunk_frames[c] > 0) { /* chunk c onto its lane (free: its previous chunk, c-N, was retired one pass ago) */
            Lane *l = &cx.lane[c % N];
            l->in = cx.batch_in[c % S];
            l->planes = write_bit ? cx.planes[c % S] : NULL;
            l->n = chunk_frames[c];
            l->first = chunk_first[c];
            l->total = M1V_E_HIP;
            group_start_front(&pool, &on_gpu[c % N], gpu_task, l, 1);
        }
        const int retire = c - N + 1;
        if (retire >= 0 && retire < n_chunks && chunk_frames[retire] > 0) {
            Lane *l = &cx.lane[retire % N];
            const int slot = retire % S, n_cur = chunk_frames[retire];
            TIMED(3, group_wait(&pool, &on_gpu[retire % N]));
            if (l->total == M1V_E_NOSPACE && l->out_cap < bound * (size_t)batch) { /* rare: grow to the worst case, redo */
                m1v_free_host(l->out);
                l->out_cap = bound * (size_t)batch;
                l->out = (uint8_t *)m1v_alloc_host(l->out_cap);
                l->total = M1V_E_HIP;
                if (l->out) gpu_task(l, 0);
            }
            if (l->total < 0) {
                printf("Error: GPU encode failed: %s\n", l->err);
                goto done;
            }
            TIMED(4, fwrite(l->out, 1, (size_t)l->total, fp));
            if (write_bit) { /* encoder.h:461-465, written behind the following chunks */
                bj[slot] = (BitJob){bitstream_folder, cx.planes[slot], frames_done + 1, W, H};
                group_start_front(&pool, &written[slot], bit_task, &bj[slot], n_cur);
            }
            frames_done += n_cur;
        }
    }
    printf("Image processing finished.\n");
    rc = 0;
    t_phase[3] = now_s();

done:
    group_wait(&pool, &warm); /* nothing may still reference what is freed below */
    for (int r = 0; r < RING; r++) group_wait(&pool, &decoded[r]);
    for (int l = 0; l < MAX_LANES; l++) group_wait(&pool, &on_gpu[l]);
    for (int s = 0; s < MAX_SLOTS; s++) {
        group_wait(&pool, &staged[s]);
        group_wait(&pool, &written[s]);
    }
    if (mismatch) { /* the reference rejects such a folder before it encodes anything */
        fflush(fp);
        if (ftruncate(fileno(fp), (off_t)sizeof prolog) != 0) perror("ftruncate");
        for (int k = 1; k <= frames_done; k++) remove_bit_file(bitstream_folder, k);
    }
    t_phase[4] = now_s();
    if (timing && rc == 0)
        fprintf(stderr, "ec504 timing: %d frames, %d threads, batch %d, %d encoder(s): first decode %.3f s, gpu encoders %.3f s, "
                        "pinned buffers %.3f s, chunks %.3f s (the caller waited: staging %.3f, decode %.3f, .bit writers %.3f, GPU %.3f, "
                        "video fwrite %.3f), last .bit writes %.3f s\n", frames_done, pool.n_threads + 1, batch,
                cx.n_lanes, t_phase[1] - t_phase[0], t_created - t_phase[1], t_phase[2] - t_created, t_phase[3] - t_phase[2],
                t_wait[0], t_wait[1], t_wait[2], t_wait[3], t_wait[4], t_phase[4] - t_phase[3]);
    if (timing && rc == 0)
        fprintf(stderr, "ec504 tasks: decode %lld x %.2f ms, stage %lld x %.2f ms, .bit write %lld x %.2f ms, GPU chunk %lld x %.2f ms "
                        "(mean duration of a task while the others run)\n",
                g_task_n[0], g_task_n[0] ? g_task_us[0] / 1e3 / g_task_n[0] : 0.0, g_task_n[1], g_task_n[1] ? g_task_us[1] / 1e3 / g_task_n[1] : 0.0,
                g_task_n[2], g_task_n[2] ? g_task_us[2] / 1e3 / g_task_n[2] : 0.0, g_task_n[3], g_task_n[3] ? g_task_us[3] / 1e3 / g_task_n[3] : 0.0);
    pool_close(&pool);
    if (cx.valid) { /* keep the device side for the next call, or give it back */
        int keep = rc == 0 && env_int("EC504_KEEP_ENCODER", 1);
        for (int l = 0; keep && l < cx.n_lanes; l++) keep = cx.lane[l].enc && cx.lane[l].out && cx.lane[l].sizes;
        pthread_mutex_lock(&g_cache_mu);
        if (keep) {
            static int at_exit_registered = 0;
            if (g_cache.valid) context_free(&g_cache);
            g_cache = cx;
            /* given back at process exit too (registered now, i.e. after the GPU runtime's own exit handlers: runs before them) */
            if (!at_exit_registered && atexit(encoder_release_cache) == 0) at_exit_registered = 1;
        } else {
            context_free(&cx);
        }
        pthread_mutex_unlock(&g_cache_mu);
    }
    free(chunk_frames);
    free(chunk_first);
    release_images(&imgs);
    for (int i = 0; i < n_paths; i++) free(paths[i]);
    free(paths);
    fclose(fp);
    return rc;
#undef TIMED
#undef CHUNK_FIRST
#undef CHUNK_FILES
}

int mpeg_encode_procedure(const char *images_folder, const char *bitstream_folder,
                          const char *video_path, int quality_factor) {
    const char *r = getenv("EC504_ENCODE_REGION");
    int region = (r && (strcmp(r, "full") == 0 || strcmp(r, "FULL") == 0 || strcmp(r, "1") == 0)) ? 1 : 0;
    return mpeg_encode_procedure_region(images_folder, bitstream_folder, video_path, quality_factor, region);
}
