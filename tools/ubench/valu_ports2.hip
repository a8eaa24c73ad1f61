// Which VALU instruction classes of gfx950 execute beside each other?  Second round (round 1: valu_ports.hip):
// more classes, every pair.  Each SIMD runs 4 waves; in a mixed run waves 0,2 execute class A and waves 1,3 class B
// (wave-uniform branch).  Serialised classes give the mean of the two pure times, classes on different units give
// about half the larger one.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define RUN(asmline)                                                                                \
    for (int it = 0; it < iters; it++)                                                              \
        asm volatile(REP16(asmline) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0), "v"(c1) : "vcc");

#define I_ADDU "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_sub_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n"
#define I_LOGI "v_and_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_lshrrev_b32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3\n"
#define I_ADDF "v_add_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_add_f32 %2, %2, %5\n v_mul_f32 %3, %3, %5\n"
#define I_FMA "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
#define I_CVT "v_cvt_f32_ubyte1 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_ubyte0 %3, %3\n"
#define I_MAD "v_mad_i32_i24 %0, %0, %4, %5\n v_mad_i32_i24 %1, %1, %4, %5\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %5\n"
#define I_CMP "v_cmp_ne_u32 vcc, %0, %4\n v_cmp_ne_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %5\n v_cmp_lt_u32 vcc, %3, %5\n"
#define I_PERM "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_bfi_b32 %2, %4, %2, %5\n v_and_or_b32 %3, %3, %4, %5\n"
#define I_MIN3 "v_min3_u32 %0, %0, %4, %5\n v_min3_u32 %1, %1, %4, %5\n v_max_u32 %2, %2, %4\n v_min_u32 %3, %3, %5\n"
#define I_DOT "v_dot4_u32_u8 %0, %0, %4, %5\n v_dot4_u32_u8 %1, %1, %4, %5\n v_dot2_i32_i16 %2, %2, %4, %5\n v_dot2_i32_i16 %3, %3, %4, %5\n"
#define I_MOV "v_mov_b32 %0, %4\n v_mov_b32 %1, %5\n v_mov_b32 %2, %4\n v_mov_b32 %3, %5\n"
#define I_FDEN "v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3\n"
constexpr int NC = 12;

template <int A, int B>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c0 = 12345, c1 = 77;
    int which = ((threadIdx.x >> 6) & 1) ? B : A;
    if (which == 0) RUN(I_ADDU)
    if (which == 1) RUN(I_LOGI)
    if (which == 2) RUN(I_ADDF)
    if (which == 3) RUN(I_FMA)
    if (which == 4) RUN(I_CVT)
    if (which == 5) RUN(I_MAD)
    if (which == 6) RUN(I_CMP)
    if (which == 7) RUN(I_PERM)
    if (which == 8) RUN(I_MIN3)
    if (which == 9) RUN(I_DOT)
    if (which == 10) RUN(I_MOV)
    if (which == 11) { // fma with a denormal multiplicand (12345 as float bits) and a large multiplier
        c1 = 0x7d000000u; // 2^123
        a0 = a1 = a2 = a3 = 0;
        RUN(I_FDEN)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

template <int A, int B>
float run(unsigned *d) {
    const int iters = 1000, blocks = 256 * 4;
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

const char *nm[NC] = {"addu", "logic", "addf", "fma", "cvt", "mad24", "cmp", "perm", "min3", "dot", "mov", "fmaden"};
float pure[NC], mix[NC][NC];

template <int A, int B>
void fill(unsigned *d) {
    mix[A][B] = run<A, B>(d);
    if constexpr (B + 1 < NC) fill<A, B + 1>(d);
    else if constexpr (A + 1 < NC) fill<A + 1, A + 1>(d);
}

int main() {
    unsigned *d; (void)hipMalloc(&d, 256 * 1024 * 4 * 4);
    (void)run<3, 3>(d); (void)run<3, 3>(d); // clocks up
    fill<0, 0>(d);
    for (int i = 0; i < NC; i++) pure[i] = mix[i][i];
    printf("pure (ms, 4 waves/SIMD all running the class; 16k instructions per wave):\n");
    for (int i = 0; i < NC; i++) printf("  %-7s %7.3f\n", nm[i], pure[i]);
    printf("mixed: overlap = (serial - measured) / (serial - parallel); serial = mean of pures, parallel = max/2\n%8s", "");
    for (int j = 0; j < NC; j++) printf("%8s", nm[j]);
    printf("\n");
    for (int i = 0; i < NC; i++) {
        printf("%-8s", nm[i]);
        for (int j = 0; j < NC; j++) {
            if (j <= i) { printf("%8s", j == i ? "-" : ""); continue; }
            float ser = 0.5f * (pure[i] + pure[j]), par = 0.5f * (pure[i] > pure[j] ? pure[i] : pure[j]);
            printf("%8.2f", (ser - mix[i][j]) / (ser - par));
        }
        printf("\n");
    }
    return 0;
}
