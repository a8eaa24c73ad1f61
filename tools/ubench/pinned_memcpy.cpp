// CPU memcpy bandwidth into and out of pinned host memory, by allocation flavour (the staging copies and the .bit writers of
// the folder pipeline, csrc/encoder_host.c):  hipcc -O2 -o /tmp/pinned_memcpy tools/ubench/pinned_memcpy.cpp -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = 96u << 20;
    char *src = (char *)malloc(bytes);
    memset(src, 1, bytes);
    struct { const char *name; unsigned flags; } kinds[] = {{"malloc (not pinned)", 0xffffffffu}, {"hipHostMallocDefault", hipHostMallocDefault},
        {"hipHostMallocNonCoherent", hipHostMallocNonCoherent}, {"hipHostMallocPortable", hipHostMallocPortable},
        {"hipHostMallocNumaUser", hipHostMallocNumaUser}};
    for (auto &k : kinds) {
        char *dst = nullptr;
        if (k.flags == 0xffffffffu) dst = (char *)malloc(bytes);
        else if (hipHostMalloc((void **)&dst, bytes, k.flags) != hipSuccess) { printf("%-28s allocation failed\n", k.name); continue; }
        memset(dst, 0, bytes);
        for (int threads : {1, 16}) {
            double best_w = 0, best_r = 0;
            for (int rep = 0; rep < 3; rep++) {
                std::vector<std::thread> th;
                double t0 = now();
                for (int t = 0; t < threads; t++) th.emplace_back([&, t] { memcpy(dst + bytes / threads * t, src + bytes / threads * t, bytes / threads); });
                for (auto &x : th) x.join();
                double w = bytes / (now() - t0) / 1e9;
                th.clear();
                t0 = now();
                for (int t = 0; t < threads; t++) th.emplace_back([&, t] { memcpy(src + bytes / threads * t, dst + bytes / threads * t, bytes / threads); });
                for (auto &x : th) x.join();
                double r = bytes / (now() - t0) / 1e9;
                best_w = w > best_w ? w : best_w;
                best_r = r > best_r ? r : best_r;
            }
            printf("%-28s %2d thread(s): copy into it %6.1f GB/s, out of it %6.1f GB/s\n", k.name, threads, best_w, best_r);
        }
        if (k.flags == 0xffffffffu) free(dst); else (void)hipHostFree(dst);
    }
    return 0;
}
