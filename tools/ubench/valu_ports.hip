// Do two VALU instruction classes share one issue port on gfx950?  Each SIMD runs 4 waves; in the mixed
// runs half of them execute class A and half class B (wave-uniform branch).  If the classes serialise
// the mixed time is the mean of the pure times; if they overlap it is max(A, B) / 2-ish.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define RUN(asmline)                                                                                \
    for (int it = 0; it < iters; it++)                                                              \
        asm volatile(REP16(asmline) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0), "v"(c1));

#define A_ADD "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n"
#define A_CVT "v_cvt_f32_ubyte1 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_ubyte0 %3, %3\n"
#define A_MAD "v_mad_i32_i24 %0, %0, %4, %5\n v_mad_i32_i24 %1, %1, %4, %5\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %5\n"
#define A_FMA "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
#define A_CMP "v_cmp_ne_u32 vcc, %0, %4\n v_cmp_ne_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %5\n v_cmp_lt_u32 vcc, %3, %5\n"
#define A_FRA "v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n"

template <int A, int B>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c0 = 12345, c1 = 77;
    int which = ((threadIdx.x >> 6) & 1) ? B : A;
    if (which == 0) RUN(A_ADD)
    if (which == 1) RUN(A_CVT)
    if (which == 2) RUN(A_MAD)
    if (which == 3) RUN(A_FMA)
    if (which == 4) { asm volatile("" ::: "vcc"); RUN(A_CMP) }
    if (which == 5) RUN(A_FRA)
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

template <int A, int B>
float run(unsigned *d) {
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, 10, d);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, iters, d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    unsigned *d; (void)hipMalloc(&d, 256 * 1024 * 4 * 4);
    const char *nm[6] = {"add", "cvt", "mad24", "fma", "cmp", "fract"};
    float pure[6] = {run<0, 0>(d), run<1, 1>(d), run<2, 2>(d), run<3, 3>(d), run<4, 4>(d), run<5, 5>(d)};
    for (int i = 0; i < 6; i++) printf("pure %-6s %7.3f ms\n", nm[i], pure[i]);
#define MIX(a, b) { float m = run<a, b>(d); printf("mix %-6s+%-6s %7.3f ms   serial would be %7.3f, overlapped %7.3f\n", nm[a], nm[b], m, 0.5f * (pure[a] + pure[b]), 0.5f * (pure[a] > pure[b] ? pure[a] : pure[b])); }
    MIX(0, 1) MIX(0, 2) MIX(1, 2) MIX(3, 1) MIX(3, 2) MIX(0, 3) MIX(0, 4) MIX(1, 4) MIX(0, 5) MIX(1, 5)
    return 0;
}
