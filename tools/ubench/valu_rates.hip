// Instruction issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the VALU
// ops k_encode_strips is made of.  Each test runs N independent chains x R repetitions of one opcode in
// inline asm, 4 waves per SIMD resident, and reports (cycles * SIMDs) / (waves * instructions).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(asmline)                                                                              \
    for (int it = 0; it < iters; it++) {                                                           \
        asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[40:41], 0x3333\n s_mov_b64 s[42:43], 0xf0f0\n" REP16(asmline) \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1) : "v"(c0), "v"(c1) \
                     : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");                \
    }

template <int OP>
__global__ __launch_bounds__(256) void k(int iters, unsigned *out, float2 *fo) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c0 = 12345, c1 = 77;
    unsigned long long b0 = a0, b1 = a1; // 64-bit regs for packed / f64 ops
    if (OP == 0) BODY("v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %7\n v_add_u32 %3, %3, %7\n")
    if (OP == 1) BODY("v_mad_i32_i24 %0, %0, %6, %7\n v_mad_i32_i24 %1, %1, %6, %7\n v_mad_i32_i24 %2, %2, %6, %7\n v_mad_i32_i24 %3, %3, %6, %7\n")
    if (OP == 2) BODY("v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte2 %2, %2\n v_cvt_f32_ubyte0 %3, %3\n")
    if (OP == 3) BODY("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n")
    if (OP == 4) BODY("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n")
    if (OP == 5) BODY("v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n")
    if (OP == 6) BODY("v_pk_fma_f32 %4, %4, %4, %5\n v_pk_fma_f32 %5, %5, %5, %4\n v_pk_fma_f32 %4, %4, %4, %5\n v_pk_fma_f32 %5, %5, %5, %4\n")
    if (OP == 7) BODY("v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cndmask_b32 %2, %2, %7, vcc\n v_cndmask_b32 %3, %3, %7, vcc\n")
    if (OP == 8) BODY("v_cmp_ne_u32 vcc, %0, %6\n v_cmp_ne_u32 vcc, %1, %6\n v_cmp_ne_u32 vcc, %2, %7\n v_cmp_ne_u32 vcc, %3, %7\n")
    if (OP == 9) BODY("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n")
    if (OP == 10) BODY("v_mul_f32 %0, %0, %6\n v_mul_f32 %1, %1, %6\n v_mul_f32 %2, %2, %7\n v_mul_f32 %3, %3, %7\n")
    if (OP == 11) BODY("v_ashrrev_i32 %0, 3, %0\n v_ashrrev_i32 %1, 3, %1\n v_ashrrev_i32 %2, 3, %2\n v_ashrrev_i32 %3, 3, %3\n")
    if (OP == 12) BODY("v_mul_lo_u32 %0, %0, %6\n v_mul_lo_u32 %1, %1, %6\n v_mul_lo_u32 %2, %2, %7\n v_mul_lo_u32 %3, %3, %7\n")
    if (OP == 13) BODY("v_mul_f64 %4, %4, %4\n v_mul_f64 %5, %5, %5\n v_add_f64 %4, %4, %5\n v_add_f64 %5, %5, %4\n")
    if (OP == 14) BODY("v_or3_b32 %0, %0, %6, %7\n v_or3_b32 %1, %1, %6, %7\n v_or3_b32 %2, %2, %6, %7\n v_or3_b32 %3, %3, %6, %7\n")
    if (OP == 15) BODY("v_mul_i32_i24 %0, %0, %6\n v_mul_i32_i24 %1, %1, %6\n v_mul_i32_i24 %2, %2, %7\n v_mul_i32_i24 %3, %3, %7\n")
    if (OP == 16) BODY("v_lshl_or_b32 %0, %0, 3, %6\n v_lshl_or_b32 %1, %1, 3, %6\n v_lshl_or_b32 %2, %2, 3, %7\n v_lshl_or_b32 %3, %3, 3, %7\n")
    if (OP == 17) BODY("v_pk_add_f32 %4, %4, %5\n v_pk_add_f32 %5, %5, %4\n v_pk_mul_f32 %4, %4, %5\n v_pk_mul_f32 %5, %5, %4\n")
    if (OP == 18) BODY("v_pk_add_u16 %0, %0, %6\n v_pk_add_u16 %1, %1, %6\n v_pk_sub_i16 %2, %2, %7\n v_pk_sub_i16 %3, %3, %7\n")
    if (OP == 19) BODY("v_add3_u32 %0, %0, %6, %7\n v_add3_u32 %1, %1, %6, %7\n v_add3_u32 %2, %2, %6, %7\n v_add3_u32 %3, %3, %6, %7\n")
    if (OP == 20) BODY("v_cvt_f64_u32 %4, %0\n v_cvt_f64_u32 %5, %1\n v_cvt_i32_f64 %2, %4\n v_cvt_i32_f64 %3, %5\n")
    if (OP == 21) BODY("v_dot4_u32_u8 %0, %0, %6, %7\n v_dot4_u32_u8 %1, %1, %6, %7\n v_dot4_u32_u8 %2, %2, %6, %7\n v_dot4_u32_u8 %3, %3, %6, %7\n")
    if (OP == 22) BODY("v_mad_u32_u24 %0, %0, %6, %7\n v_mad_u32_u24 %1, %1, %6, %7\n v_bfe_u32 %2, %2, 8, 8\n v_bfe_u32 %3, %3, 8, 8\n")
    if (OP == 23) BODY("v_perm_b32 %0, %0, %6, %7\n v_perm_b32 %1, %1, %6, %7\n v_alignbit_b32 %2, %2, %6, 8\n v_alignbit_b32 %3, %3, %6, 8\n")
    if (OP == 24) BODY("v_cndmask_b32_e64 %0, %0, %6, s[40:41]\n v_cndmask_b32_e64 %1, %1, %6, s[40:41]\n v_cndmask_b32_e64 %2, %2, %7, s[42:43]\n v_cndmask_b32_e64 %3, %3, %7, s[42:43]\n")
    if (OP == 25) BODY("v_and_b32 %0, %0, %6\n v_or_b32 %1, %1, %6\n v_xor_b32 %2, %2, %7\n v_lshrrev_b32 %3, 1, %3\n")
    if (OP == 26) BODY("v_max_i32 %0, %0, %6\n v_min_i32 %1, %1, %6\n v_max_f32 %2, %2, %7\n v_sub_u32 %3, %3, %7\n")
    if (OP == 27) BODY("v_med3_i32 %0, %0, %6, %7\n v_bfe_i32 %1, %1, 8, 8\n v_bfi_b32 %2, %2, %6, %7\n v_lshl_add_u32 %3, %3, 2, %7\n")
    if (OP == 28) BODY("v_cmp_ne_u32_e64 s[40:41], %0, %6\n v_cmp_ne_u32_e64 s[42:43], %1, %6\n v_cmp_ge_f32_e64 s[44:45], |%2|, %7\n v_cmp_lt_i32_e64 s[46:47], %3, %7\n")
    if (OP == 29) BODY("v_mov_b32 %0, %6\n v_mov_b32 %1, %7\n v_mov_b32 %2, %6\n v_mov_b32 %3, %7\n")
    if (OP == 30) BODY("v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_addc_co_u32 %2, vcc, %2, %2, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n")
    if (OP == 31) BODY("v_cvt_pk_i16_i32 %0, %0, %6\n v_cvt_pk_i16_i32 %1, %1, %6\n v_pack_b32_f16 %2, %2, %7\n v_pack_b32_f16 %3, %3, %7\n")
    if (OP == 32) BODY("v_cndmask_b32 %0, %0, %6, vcc\n v_cndmask_b32 %1, %1, %6, vcc\n v_cndmask_b32 %2, %2, %7, vcc\n v_cndmask_b32 %3, %3, %7, vcc\n")
    if (OP == 33) BODY("v_add_u32_sdwa %0, %0, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_mul_u32_u24_sdwa %2, %2, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_mul_u32_u24_sdwa %3, %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n")
    if (OP == 34) BODY("v_cmp_ne_u32 vcc, %0, %6\n v_cndmask_b32 %1, %1, %6, vcc\n v_cmp_ne_u32 vcc, %2, %7\n v_cndmask_b32 %3, %3, %7, vcc\n")
    if (OP == 35) BODY("v_cmp_ne_u32_e64 s[44:45], %0, %6\n v_cndmask_b32_e64 %1, %1, %6, s[44:45]\n v_cmp_ne_u32_e64 s[46:47], %2, %7\n v_cndmask_b32_e64 %3, %3, %7, s[46:47]\n")
    if (OP == 36) BODY("v_cmp_ne_u32 vcc, %0, %6\n v_addc_co_u32 %1, vcc, %1, %1, vcc\n v_cmp_ne_u32 vcc, %2, %7\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n")
    if (OP == 37) BODY("v_mul_f32 %0, %0, %6 clamp\n v_mul_f32 %1, |%1|, %6 clamp\n v_min_u32 %2, %2, %7\n v_min_u32 %3, %3, %7\n")
    if (OP == 38) BODY("v_bfi_b32 %0, %6, %0, %7\n v_bfi_b32 %1, %6, %1, %7\n v_and_or_b32 %2, %2, %6, %7\n v_and_or_b32 %3, %3, %6, %7\n")
    if (OP == 39) BODY("v_sub_f32_e64 %0, %6, |%0|\n v_sub_f32_e64 %1, %6, |%1|\n v_sub_f32_e64 %2, %7, |%2|\n v_sub_f32_e64 %3, %7, |%3|\n")
    if (OP == 40) BODY("v_alignbit_b32 %0, %0, %6, 31\n v_alignbit_b32 %1, %1, %6, 31\n v_alignbit_b32 %2, %2, %7, 31\n v_alignbit_b32 %3, %3, %7, 31\n")
    if (OP == 41) BODY("v_max3_f32 %0, %0, %6, %7\n v_max3_f32 %1, %1, %6, %7\n v_max3_f32 %2, %2, %6, %7\n v_max3_f32 %3, %3, %6, %7\n")
    if (OP == 42) BODY("v_trunc_f32 %0, %0\n v_floor_f32 %1, %1\n v_rndne_f32 %2, %2\n v_trunc_f32 %3, %3\n")
    if (OP == 43) BODY("v_dot2_i32_i16 %0, %0, %6, %7\n v_dot2_i32_i16 %1, %1, %6, %7\n v_dot2_u32_u16 %2, %2, %6, %7\n v_dot2_u32_u16 %3, %3, %6, %7\n")
    if (OP == 44) BODY("v_cvt_pk_u8_f32 %0, %0, %6, %7\n v_cvt_pk_u8_f32 %1, %1, %6, %7\n v_cvt_pk_u8_f32 %2, %2, %6, %7\n v_cvt_pk_u8_f32 %3, %3, %6, %7\n")
    if (OP == 45) BODY("v_max_f32 %0, %0, %6\n v_max_f32 %1, %1, %6\n v_min_f32 %2, %2, %7\n v_min_f32 %3, %3, %7\n")
    if (OP == 46) BODY("v_sub_f32 %0, %6, %0\n v_add_f32 %1, %6, %1\n v_sub_f32 %2, %7, %2\n v_add_f32 %3, %7, %3\n")
    if (OP == 47) BODY("v_fma_f32 %0, |%0|, %6, %7\n v_fma_f32 %1, -%1, %6, %7\n v_fma_f32 %2, %2, %6, -%7\n v_fma_f32 %3, %3, %6, %7 clamp\n")
    if (OP == 48) BODY("v_mad_u32_u16 %0, %0, %6, %7\n v_mad_u32_u16 %1, %1, %6, %7\n v_mad_i32_i16 %2, %2, %6, %7\n v_mad_i32_i16 %3, %3, %6, %7\n")
    if (OP == 49) BODY("v_fma_mix_f32 %0, %0, %6, %7 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %1, %6, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %2, %6, %7 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %3, %6, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n")
    if (OP == 50) BODY("v_and_b32 %0, 0xff00ff, %0\n v_and_b32 %1, 0xff00ff, %1\n v_lshrrev_b32 %2, 8, %2\n v_lshrrev_b32 %3, 8, %3\n")
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (unsigned)b0 ^ (unsigned)b1;
}

template <int OP>
void run(const char *name, unsigned *d, float2 *f) {
    const int iters = 2000, blocks = 256 * 4; // 4 blocks of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, 10, d, f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, iters, d, f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)blocks * 4 / 1024.0 * iters * 64; // waves per SIMD * instructions per wave
    printf("%-44s %8.3f ms  %6.2f ns per wave-instr per SIMD  (= %5.2f cycles @2.4GHz)\n", name, ms,
           ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}

int main() {
    unsigned *d; float2 *f;
    hipMalloc(&d, 256 * 1024 * 4 * 4); hipMalloc(&f, 1024);
    run<0>("v_add_u32", d, f);        run<1>("v_mad_i32_i24", d, f);   run<15>("v_mul_i32_i24", d, f);
    run<12>("v_mul_lo_u32", d, f);    run<11>("v_ashrrev_i32", d, f);  run<19>("v_add3_u32", d, f);
    run<14>("v_or3_b32", d, f);       run<16>("v_lshl_or_b32", d, f);  run<7>("v_cndmask_b32", d, f);
    run<8>("v_cmp_ne_u32", d, f);     run<2>("v_cvt_f32_ubyteN", d, f); run<9>("v_cvt_f32_i32", d, f);
    run<4>("v_cvt_i32_f32", d, f);    run<3>("v_fract_f32", d, f);     run<10>("v_mul_f32", d, f);
    run<5>("v_fma_f32", d, f);        run<6>("v_pk_fma_f32", d, f);    run<17>("v_pk_add_f32/v_pk_mul_f32", d, f);
    run<18>("v_pk_add_u16/v_pk_sub_i16", d, f); run<13>("v_mul_f64/v_add_f64", d, f); run<20>("v_cvt_f64_u32/v_cvt_i32_f64", d, f);
    run<21>("v_dot4_u32_u8", d, f);   run<22>("v_mad_u32_u24/v_bfe_u32", d, f); run<23>("v_perm_b32/v_alignbit_b32", d, f);
    run<32>("v_cndmask_b32 (vcc, e32)", d, f); run<24>("v_cndmask_b32_e64 (sgpr pair)", d, f);
    run<25>("v_and/v_or/v_xor/v_lshrrev", d, f); run<26>("v_max_i32/v_min_i32/v_max_f32/v_sub_u32", d, f);
    run<27>("v_med3_i32/v_bfe_i32/v_bfi_b32/v_lshl_add_u32", d, f); run<28>("v_cmp_*_e64 -> sgpr", d, f);
    run<29>("v_mov_b32", d, f); run<30>("v_addc_co_u32", d, f); run<31>("v_cvt_pk_i16_i32/v_pack_b32_f16", d, f);
    run<33>("v_add_u32_sdwa/v_mul_u32_u24_sdwa", d, f);
    run<34>("pair v_cmp(vcc)+v_cndmask_e32(vcc)", d, f); run<35>("pair v_cmp_e64(sgpr)+v_cndmask_e64", d, f);
    run<36>("pair v_cmp(vcc)+v_addc_co_u32", d, f); run<37>("v_mul_f32 clamp / v_min_u32", d, f); run<38>("v_bfi_b32/v_and_or_b32", d, f);
    run<39>("v_sub_f32_e64 with |abs|", d, f); run<40>("v_alignbit_b32", d, f); run<41>("v_max3_f32", d, f);
    run<42>("v_trunc/floor/rndne_f32", d, f); run<43>("v_dot2_i32_i16/u32_u16", d, f); run<44>("v_cvt_pk_u8_f32", d, f);
    run<45>("v_max_f32/v_min_f32", d, f); run<46>("v_sub_f32/v_add_f32", d, f); run<47>("v_fma_f32 with modifiers", d, f);
    run<48>("v_mad_u32_u16/v_mad_i32_i16", d, f); run<49>("v_fma_mix_f32 (f16 src0)", d, f); run<50>("v_and_b32 literal / v_lshrrev", d, f);
    return 0;
}
