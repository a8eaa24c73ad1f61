// tools/ubench/dma_probe.hip — what global_load_lds_dwordx4 (LDS-DMA) does on gfx950, checked on the GPU before the
// tile kernel relies on it: destination = M0 + lane * 16, per-lane source = SGPR base + 32-bit VGPR offset, EXEC-masked
// lanes write nothing, source alignment below 16 bytes, and what `offset:` is added to.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/dma_probe.hip -o /tmp/dma_probe && /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define DMA16(voff, ldsdst, sbase)                                                                                    \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep) : "v"(voff), "s"(ldsdst), "s"(sbase) : "memory")
#define DMA16_OFF(voff, ldsdst, sbase)                                                                                \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 offset:32\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep) : "v"(voff), "s"(ldsdst), "s"(sbase) : "memory")

// test: 0 = plain, 1 = lanes 0..31 only, 2 = lanes 0..47, 3 = source misaligned by `mis` bytes, 4 = offset:32
__global__ void probe(const uint8_t *src, uint8_t *out, int test, int mis) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) lds[i] = 0xEE;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds + 256u;
    uint32_t keep;
    // per-lane source: a permutation, so that "lane order" and "address order" differ
    const uint32_t voff = (uint32_t)((lane * 37) % 64) * 48u + (uint32_t)mis;
    if (test == 0 || test == 3) DMA16(voff, base, src);
    if (test == 1 && lane < 32) DMA16(voff, base, src);
    if (test == 2 && lane < 48) DMA16(voff, base, src);
    if (test == 4) DMA16_OFF(voff, base, src);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 4096; i += 64) out[i] = lds[i];
}

// 24-byte rows read back by three ds_read_b64 at 8-byte aligned addresses
__global__ void read24(const uint8_t *src, uint32_t *out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) lds[i] = src[i];
    __syncthreads();
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)lds + lane * 24u;
    unsigned long long r0, r1, r2;
    asm volatile("ds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:8\n\tds_read_b64 %2, %3 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2) : "v"(la));
    out[lane * 6 + 0] = (uint32_t)r0; out[lane * 6 + 1] = (uint32_t)(r0 >> 32);
    out[lane * 6 + 2] = (uint32_t)r1; out[lane * 6 + 3] = (uint32_t)(r1 >> 32);
    out[lane * 6 + 4] = (uint32_t)r2; out[lane * 6 + 5] = (uint32_t)(r2 >> 32);
}

int main() {
    std::vector<uint8_t> h(8192);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint8_t)((i * 7 + (i >> 8) * 13) & 0xff);
    uint8_t *d_src, *d_out;
    hipMalloc(&d_src, h.size());
    hipMalloc(&d_out, 4096);
    hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice);
    int bad_total = 0;
    struct { int test, mis; const char *what; } cases[] = {
        {0, 0, "64 lanes, 16-byte aligned sources"}, {1, 0, "lanes 0..31 active"}, {2, 0, "lanes 0..47 active"},
        {3, 8, "sources misaligned by 8"}, {3, 4, "sources misaligned by 4"}, {3, 1, "sources misaligned by 1"},
        {4, 0, "offset:32 (expect: added to the source AND to the LDS address?)"}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_src, d_out, c.test, c.mis);
        std::vector<uint8_t> o(4096);
        if (hipMemcpy(o.data(), d_out, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("%s: HIP error\n", c.what); return 2; }
        const int active = c.test == 1 ? 32 : (c.test == 2 ? 48 : 64);
        int bad = 0, touched_outside = 0;
        // model A: lds[256 + lane*16 + b] = src[voff + b]; for test 4 try both "offset added to LDS too" and "source only"
        int bad_src_only = 0, bad_both = 0;
        for (int lane = 0; lane < 64; lane++)
            for (int b = 0; b < 16; b++) {
                const uint32_t voff = (uint32_t)((lane * 37) % 64) * 48u + c.mis;
                if (c.test == 4) {
                    if (o[256 + lane * 16 + b] != h[voff + 32 + b]) bad_src_only++;
                    if (o[256 + 32 + lane * 16 + b] != h[voff + 32 + b]) bad_both++;
                } else if (lane < active) {
                    if (o[256 + lane * 16 + b] != h[voff + b]) bad++;
                } else if (o[256 + lane * 16 + b] != 0xEE) touched_outside++;
            }
        for (int i = 0; i < 256; i++) if (o[i] != 0xEE) touched_outside++;
        for (int i = 256 + 1024 + (c.test == 4 ? 32 : 0); i < 4096; i++) if (o[i] != 0xEE) touched_outside++;
        if (c.test == 4)
            printf("%-62s source-only model: %d bad bytes; source+LDS model: %d bad bytes\n", c.what, bad_src_only, bad_both);
        else
            printf("%-62s %d bad bytes, %d bytes touched outside\n", c.what, bad, touched_outside);
        bad_total += bad + touched_outside;
    }
    uint32_t *d_o32;
    hipMalloc(&d_o32, 64 * 6 * 4);
    hipLaunchKernelGGL(read24, dim3(1), dim3(64), 0, 0, d_src, d_o32);
    std::vector<uint32_t> o32(64 * 6);
    hipMemcpy(o32.data(), d_o32, o32.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64 * 24; i++) if (((o32[i / 4] >> (8 * (i & 3))) & 0xff) != h[i]) bad++;
    printf("%-62s %d bad bytes\n", "3 x ds_read_b64 at 24-byte stride (8-byte aligned)", bad);
    bad_total += bad;
    printf(bad_total ? "PROBE: assumptions violated\n" : "PROBE: plain / masked / 24-byte-read assumptions hold\n");
    return 0;
}
