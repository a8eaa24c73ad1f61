#!/usr/bin/env python3
"""Generates tools/ubench/stream_probe.hip: long single-wave instruction streams that differ in ONE thing at a time, to find
out what decides whether half-rate ops (cvt/floor/...) and float ops of one stream run beside each other
(profiles/r02_valu_stream_mix.txt: "cvt fma cvt fma" 8.5 cycles per group) or after each other (profiles/r02_synth_rows.txt).
Each instruction works in place on one of R rotating registers; every wave of the device runs the same stream."""
import sys
OPS = {
    "c": "v_cvt_f32_ubyte1 {d}, {d}",
    "f": "v_fma_f32 {d}, {d}, %{k0}, %{k1}",
    "a": "v_add_f32 {d}, {d}, %{k0}",
    "s": "v_sub_f32 {d}, {d}, %{k1}",
    "m": "v_mul_f32 {d}, {d}, %{k0}",
    "n": "v_and_b32 {d}, 0xffff8000, {d}",
    "N": "v_and_b32 {d}, %{k0}, {d}",
    "x": "v_max3_f32 {d}, {d}, %{k0}, %{k1}",
    "l": "v_floor_f32 {d}, {d}",
    "i": "v_cvt_i32_f32 {d}, {d}",
    "j": "v_cvt_f32_i32 {d}, {d}",
    "u": "v_mul_i32_i24 {d}, {d}, %{k0}",
    "r": "v_ashrrev_i32 {d}, 10, {d}",
    "p": "v_cmp_lt_u32 vcc, {d}, %{k0}",
    "+": "v_add_u32 {d}, {d}, %{k0}",
    "k": "v_fmac_f32 {d}, %{k0}, %{k1}",
    "L": "v_lshl_add_u32 {d}, {d}, 2, %{k0}",
    "q": "v_mul_lo_u32 {d}, {d}, %{k0}",
    "M": "v_mad_i32_i24 {d}, {d}, %{k0}, %{k1}",
    "U": "v_mul_u32_u24 {d}, {d}, %{k0}",
    "h": "v_mul_hi_i32_i24 {d}, {d}, %{k0}",
    "K": "v_mul_i32_i24 {d}, 0xb5, {d}",
    "P": "v_perm_b32 {d}, {d}, %{k0}, %{k1}",
    "X": "v_fma_mix_f32 {d}, {d}, %{k0}, %{k1} op_sel_hi:[1,0,0]",
    "Y": "v_fma_mix_f32 {d}, {d}, %{k0}, %{k1} op_sel:[1,0,0] op_sel_hi:[1,0,0]",
    "R": "v_lshrrev_b32 {d}, 8, {d}",
}
tests = []
def t(name, pattern, regs=12, waves=5, reps=None):
    n = len(pattern)
    reps = reps or max(1, 180 // n)
    tests.append((name, pattern * reps, regs, waves))
# 1: reproduce the short-pattern result, then move towards the synthetic row stream one step at a time
t("cf x, 4 regs, 4 waves", "cf", 4, 4)
t("cf x, 12 regs, 4 waves", "cf", 12, 4)
t("cf x, 12 regs, 5 waves", "cf", 12, 5)
t("cf x, 12 regs, 2 waves", "cf", 12, 2)
t("cff", "cff")
t("ccffff", "ccffff")
t("cccfffff f (3 cvt 6 fma)", "cccffffff")
t("c x24 then f x48", "c" * 24 + "f" * 48, reps=2)
t("only c", "c")
t("only f", "f")
t("only a (add_f32)", "a")
t("only k (fmac, VOP2)", "k")
t("ca", "ca")
t("ck (cvt + fmac VOP2)", "ck")
t("cfn (and literal)", "cfn")
t("cfN (and register)", "cfN")
t("cf+ (add_u32)", "cf+")
t("cffn", "cffn")
t("cfx (max3)", "cfx")
t("lf (floor fma)", "lf")
t("la (floor add)", "la")
t("if (cvt_i32 fma)", "if")
t("uf (mul24 fma)", "uf")
t("pf (cmp fma)", "pf")
t("xf (max3 fma)", "xf")
t("xa (max3 add)", "xa")
# the colour conversion of one pixel pair and the pieces of a row pass
t("colour pair: cccfffns cccfffns x", "cccfffnscccfffnsx")
t("colour pair interleaved: cfcfcfns cfcfcfns x", "cfcfcfnscfcfcfnsx")
t("colour pair, ops ordered c f c f c f s n", "cfcfcfsncfcfcfsnx")
t("butterfly: a x12 m f f a m f f a a a m f f s x6 l x4", "aaaaaaaaaaaa" + "mffamffaaamff" + "ssssss" + "llll")
t("quant: mi x8", "mi" * 8)
t("quant: mmmmmmmm iiiiiiii", "m" * 8 + "i" * 8)

# the synthetic row step of gen_synth_rows.py (flt_cvt) in this grammar, whole and with one piece left out / replaced
colour = "".join("cccfffns" + ("x" if px % 2 else "") for px in range(8)) + "p"
row = "a" * 12 + "mffamffa" + "aamff" + "s" * 6 + "l" * 4 + "miurj" * 2
col = "a" * 12 + "fff" + "afff" + "amffamff" + "ssss" + "aas" + "l" * 6 + "mlfl" * 2
quant = "mi" * 8
t("row step: colour + row + col + quant", colour + row + col + quant, reps=1)
t("row step x2 in one asm", colour + row + col + quant, reps=2)
t("  colour only", colour, reps=1)
t("  colour x3", colour, reps=3)
t("  row only x4", row, reps=4)
t("  col only x4", col, reps=4)
t("  quant only x10", quant, reps=10)
t("  colour + row", colour + row, reps=1)
t("  colour + quant", colour + quant, reps=1)
t("  row + col + quant", row + col + quant, reps=2)
t("  row step, mul24 -> mul_f32", (colour + row + col + quant).replace("u", "m"), reps=1)
t("  row step, no and (n -> s)", (colour + row + col + quant).replace("n", "s"), reps=1)
t("  row step, 4 waves", colour + row + col + quant, waves=4, reps=1)
t("  row step, 3 waves", colour + row + col + quant, waves=3, reps=1)
t("  row step, 8 waves", colour + row + col + quant, waves=8, reps=1)

# what does the 24-bit integer multiply cost, and next to what?
step = colour + row + col + quant
t("only u (mul_i32_i24)", "u")
t("miurj", "miurj")
t("mimrj (u -> mul_f32)", "mimrj")
t("miLLLrj (u -> 3 lshl_add)", "miLLLrj")
t("miKrj (u with inline constant)", "miKrj")
t("miUrj (mul_u32_u24)", "miUrj")
t("miMrj (mad_i32_i24)", "miMrj")
t("miqrj (mul_lo_u32)", "miqrj")
t("ffffu", "ffffu")
t("ffffffffu", "ffffffffu")
t("aaaaaaaau", "aaaaaaaau")
t("ffffffffi", "ffffffffi")
t("ffffffffr", "ffffffffr")
t("fffffffuf ffffffffu f: 1 in 16", "f" * 15 + "u")
t("1 u in 32 f", "f" * 31 + "u", reps=6)
t("1 u in 64 f", "f" * 63 + "u", reps=3)
t("1 u in 64 a", "a" * 63 + "u", reps=3)
t("1 u in 64 c", "c" * 63 + "u", reps=3)
t("1 u in 64 +", "+" * 63 + "u", reps=3)
t("1 L in 64 f", "f" * 63 + "L", reps=3)
t("1 q in 64 f", "f" * 63 + "q", reps=3)
t("1 + in 64 f", "f" * 63 + "+", reps=3)
t("1 r in 64 f", "f" * 63 + "r", reps=3)
t("1 n in 64 f", "f" * 63 + "n", reps=3)
t("1 c in 64 f", "f" * 63 + "c", reps=3)
t("1 l in 64 f", "f" * 63 + "l", reps=3)
t("1 i in 64 f", "f" * 63 + "i", reps=3)
t("1 x in 64 f", "f" * 63 + "x", reps=3)
t("1 p in 64 f", "f" * 63 + "p", reps=3)
t("1 f in 64 +", "+" * 63 + "f", reps=3)
t("  row step, u -> LLL", step.replace("u", "LLL"), reps=1)
t("  row step, u -> U", step.replace("u", "U"), reps=1)
t("  row step, u -> q", step.replace("u", "q"), reps=1)
t("  row step, u -> +", step.replace("u", "+"), reps=1)
t("  row step, r -> s", step.replace("r", "s"), reps=1)

# colour conversion with the bytes taken as f16 denormals by v_fma_mix_f32 (2 bytes per v_perm_b32) instead of cvt + fma
mix_pair = "PPP" + "XYXYXY" + "ns" + "ns" + "x"                  # two pixels: 3 perms, 6 mixed fmas
mix_pair_and = "NRN" * 1 + "NN" + "XYXYXY" + "ns" + "ns" + "x"     # two pixels = 1.5 dwords: and / shift+and unpack (~4.5 ops)
t("colour pair (cvt + fma), reference", "cccfffnscccfffnsx")
t("colour pair, perm + fma_mix", mix_pair)
t("colour pair, and/shift + fma_mix", mix_pair_and)
t("only X (fma_mix)", "X")
t("only P (perm)", "P")
t("PX", "PX")
t("Xf", "Xf")
t("Xa", "Xa")
colour_mix = "".join(mix_pair for _ in range(4)) + "p"
t("row step, cvt colour (reference)", colour + row.replace("u", "q") + col + quant, reps=1)
t("row step, perm + fma_mix colour", colour_mix + row.replace("u", "q") + col + quant, reps=1)

out = ['// GENERATED by gen_stream_probe.py, see that file.', '#include <hip/hip_runtime.h>', '#include <stdio.h>',
       'template <int P> __global__ __launch_bounds__(256) void k(int iters, unsigned *out) {',
       '    unsigned v[12]; for (int i = 0; i < 12; i++) v[i] = threadIdx.x + i; unsigned c0 = 12345, c1 = 77;']
for i, (name, pat, regs, waves) in enumerate(tests):
    body = "\\n".join(OPS[c].format(d=f"%{j % regs}", k0=12, k1=13) for j, c in enumerate(pat)) + "\\n"
    out.append(f'    if (P == {i}) for (int it = 0; it < iters; it++) asm volatile("{body}" : ' +
               ", ".join(f'"+v"(v[{j}])' for j in range(12)) + ' : "v"(c0), "v"(c1) : "vcc");')
out += ['    unsigned x = 0; for (int i = 0; i < 12; i++) x ^= v[i]; out[blockIdx.x * 256 + threadIdx.x] = x;', '}',
        'template <int P> void run(const char *name, int n, int nS, int waves, unsigned *d) {',
        '    const int iters = 2000, blocks = 256 * waves;', '    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);',
        '    hipLaunchKernelGGL((k<P>), dim3(blocks), dim3(256), 0, 0, 10, d); (void)hipEventRecord(e0);',
        '    hipLaunchKernelGGL((k<P>), dim3(blocks), dim3(256), 0, 0, iters, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);',
        '    float ms; (void)hipEventElapsedTime(&ms, e0, e1);',
        '    double cyc = ms * 1e6 / ((double)waves * iters) * 2.0;',
        '    printf("%-52s %d waves  %3d instr (%3d half-rate)  %7.1f cycles@2GHz  %5.2f per instr   sum model %4d  overlap model %4d\\n", name, waves, n, nS, cyc, cyc / n, nS * 4 + (n - nS) * 2, nS * 4 > (n - nS) * 2 ? nS * 4 : n * 2);',
        '}', 'int main() {', '    unsigned *d; (void)hipMalloc(&d, 256 * 1024 * 8 * 4);', '    run<9>("warm", 1, 0, 4, d); run<9>("warm", 1, 0, 4, d);']
for i, (name, pat, regs, waves) in enumerate(tests):
    nS = sum(pat.count(c) for c in "cxlijupqMUhKPXY")
    out.append(f'    run<{i}>("{name}", {len(pat)}, {nS}, {waves}, d);')
out += ['    return 0;', '}']
open(__file__.replace("gen_stream_probe.py", "stream_probe.hip"), "w").write("\n".join(out) + "\n")
print(len(tests), "tests")
