// Do instruction classes that run beside each other when they come from DIFFERENT waves (valu_ports2.hip) also do so
// when ONE wave's stream interleaves them?  Every wave runs the same stream of 4-instruction groups on independent
// registers, 4 waves per SIMD.  Reported: ns per GROUP per wave per SIMD and the same in cycles at 2.0 GHz; compare with
// the sum of the members' pure costs (2 cycles: add/logic/fma/mul/mov, 4 cycles: cvt/cmp/mad24/perm/floor).
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(x) x x x x x x x x x x x x x x x x
#define RUN(asmline)                                                                                \
    for (int it = 0; it < iters; it++)                                                              \
        asm volatile(REP16(asmline) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c0), "v"(c1) : "vcc");

template <int P>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c0 = 12345, c1 = 77;
    if (P == 0) RUN("v_cvt_f32_ubyte1 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_cvt_f32_ubyte2 %2, %2\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 1) RUN("v_cvt_f32_ubyte1 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 2) RUN("v_cvt_f32_ubyte1 %0, %0\n v_add_u32 %1, %1, %4\n v_cvt_f32_ubyte2 %2, %2\n v_add_u32 %3, %3, %5\n")
    if (P == 3) RUN("v_add_u32 %0, %0, %4\n v_fma_f32 %1, %1, %4, %5\n v_add_u32 %2, %2, %5\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 4) RUN("v_cmp_lt_u32 vcc, %0, %4\n v_fma_f32 %1, %1, %4, %5\n v_cmp_lt_u32 vcc, %2, %5\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 5) RUN("v_mad_i32_i24 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_mad_i32_i24 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 6) RUN("v_mad_i32_i24 %0, %0, %4, %5\n v_cvt_f32_ubyte1 %1, %1\n v_mad_i32_i24 %2, %2, %4, %5\n v_cvt_f32_ubyte0 %3, %3\n")
    if (P == 7) RUN("v_floor_f32 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_floor_f32 %2, %2\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 8) RUN("v_cvt_f32_ubyte1 %0, %0\n v_add_f32 %1, %1, %4\n v_cvt_i32_f32 %2, %2\n v_sub_f32 %3, %3, %5\n")
    if (P == 9) RUN("v_cvt_f32_ubyte1 %0, %0\n v_fma_f32 %1, %1, %4, %5\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %5\n")
    if (P == 10) RUN("v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte2 %1, %1\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 11) RUN("v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte2 %1, %1\n v_cvt_f32_ubyte3 %2, %2\n v_cvt_f32_ubyte0 %3, %3\n")
    if (P == 12) RUN("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 13) RUN("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %5\n v_add_u32 %3, %3, %5\n")
    if (P == 14) RUN("v_cvt_f32_ubyte1 %0, %0\n v_mov_b32 %1, %4\n v_cvt_f32_ubyte2 %2, %2\n v_mov_b32 %3, %5\n")
    if (P == 15) RUN("v_rndne_f32 %0, %0\n v_add_f32 %1, %1, %4\n v_trunc_f32 %2, %2\n v_mul_f32 %3, %3, %5\n")
    if (P == 16) RUN("v_cndmask_b32_e64 %0, %0, %4, s[40:41]\n v_fma_f32 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %5, s[40:41]\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 17) RUN("v_perm_b32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_bfe_u32 %2, %2, 8, 8\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 18) RUN("v_mul_i32_i24 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_mul_i32_i24 %2, %2, %5\n v_add_f32 %3, %3, %5\n")
    if (P == 19) RUN("v_add3_u32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_lshl_add_u32 %2, %2, 3, %5\n v_fma_f32 %3, %3, %4, %5\n")
    if (P == 20) RUN("v_cvt_f32_i32 %0, %0\n v_mul_f32 %1, %1, %4\n v_cvt_i32_f32 %2, %2\n v_and_b32 %3, %3, %5\n")
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3;
}

template <int P>
void run(const char *name, int expect, unsigned *d) {
    const int iters = 1000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<P>), dim3(blocks), dim3(256), 0, 0, 10, d);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<P>), dim3(blocks), dim3(256), 0, 0, iters, d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double groups_per_simd = 4.0 * iters * 16; // 4 waves per SIMD, 16 groups per iteration
    double ns = ms * 1e6 / groups_per_simd;
    printf("%-46s %7.3f ms  %6.2f ns/group = %5.2f cycles @2.0GHz   (sum of pure costs %d)\n", name, ms, ns, ns * 2.0, expect);
}

int main() {
    unsigned *d; (void)hipMalloc(&d, 256 * 1024 * 4 * 4);
    run<12>("warm", 8, d); run<12>("fma x4", 8, d); run<13>("add_u32 x4", 8, d); run<11>("cvt x4", 16, d);
    run<0>("cvt fma cvt fma", 12, d); run<1>("cvt fma fma fma", 10, d); run<10>("cvt cvt fma fma", 12, d);
    run<2>("cvt addu cvt addu", 12, d); run<3>("addu fma addu fma", 8, d); run<9>("cvt fma addu addu", 10, d);
    run<4>("cmp fma cmp fma", 12, d); run<5>("mad24 fma mad24 fma", 12, d); run<6>("mad24 cvt mad24 cvt", 16, d);
    run<7>("floor fma floor fma", 12, d); run<8>("cvt addf cvt subf", 12, d); run<14>("cvt mov cvt mov", 12, d);
    run<15>("rndne addf trunc mulf", 12, d); run<16>("cndmask_e64 fma x2", 12, d); run<17>("perm fma bfe fma", 12, d);
    run<18>("mul24 addf mul24 addf", 12, d); run<19>("add3 fma lshl_add fma", 12, d); run<20>("cvt_f_i mulf cvt_i_f and", 12, d);
    return 0;
}
