#!/usr/bin/env python3
"""Generates tools/ubench/synth_rows.hip: synthetic instruction streams with the class mix of one block row of the encode
kernel's pixel stage (colour conversion of 8 pixels + one 8-point FDCT pass + quantise/stage share), in several
formulations, to price a formulation BEFORE it is written: integer butterfly (round 1), float butterfly, unpack by
cvt_f32_ubyte or by and + denormal fma.  Registers rotate so that nothing depends on the previous few instructions."""
import itertools, random
random.seed(1)
R = 12  # rotating vector registers
def reg(i): return f"%{i % R}"
class S:
    def __init__(s): s.l = []; s.i = 0
    def r(s): s.i += 1; return reg(s.i)
    def add(s, op): s.l.append(op)
    def cvt(s): d = s.r(); s.add(f"v_cvt_f32_ubyte1 {d}, {d}")
    def fma(s): d = s.r(); s.add(f"v_fma_f32 {d}, {d}, %12, %13")
    def addf(s): d = s.r(); s.add(f"v_add_f32 {d}, {d}, %12")
    def subf(s): d = s.r(); s.add(f"v_sub_f32 {d}, {d}, %13")
    def mulf(s): d = s.r(); s.add(f"v_mul_f32 {d}, {d}, %12")
    def addu(s): d = s.r(); s.add(f"v_add_u32 {d}, {d}, %12")
    def subu(s): d = s.r(); s.add(f"v_sub_u32 {d}, {d}, %13")
    def andb(s): d = s.r(); s.add(f"v_and_b32 {d}, 0xffff8000, {d}")
    def lshr(s): d = s.r(); s.add(f"v_lshrrev_b32 {d}, 15, {d}")
    def ashr(s): d = s.r(); s.add(f"v_ashrrev_i32 {d}, 10, {d}")
    def orb(s): d = s.r(); s.add(f"v_or_b32 {d}, {d}, %12")
    def cmp(s): d = s.r(); s.add(f"v_cmp_lt_u32 vcc, {d}, %12")
    def mad24(s): d = s.r(); s.add(f"v_mad_i32_i24 {d}, {d}, %12, %13")
    def mul24(s): d = s.r(); s.add(f"v_mul_i32_i24 {d}, {d}, %12")
    def floor(s): d = s.r(); s.add(f"v_floor_f32 {d}, {d}")
    def cvti(s): d = s.r(); s.add(f"v_cvt_i32_f32 {d}, {d}")
    def cvtf(s): d = s.r(); s.add(f"v_cvt_f32_i32 {d}, {d}")
    def max3(s): d = s.r(); s.add(f"v_max3_f32 {d}, {d}, %12, %13")
    def max3u(s): d = s.r(); s.add(f"v_max3_u32 {d}, {d}, %12, %13")
    def pkfma(s): d = s.r(); s.i += 1; s.add(f"v_pk_fma_f32 %{14 + (s.i // 2) % 2}, %{14 + (s.i // 2) % 2}, %16, %17")
    def text(s): return "\\n".join(s.l) + "\\n"

def colour_int(s, unpack):  # round-1 form: value = bits >> 15, flag = (bits & 0x7fff) >= limit, per pixel
    for px in range(8):
        for c in range(3):
            (s.cvt() if unpack == "cvt" else s.andb())
        for c in range(3): s.fma()
        s.lshr(); s.andb(); s.cmp()
def colour_flt(s, unpack):  # float form: trunc by and, fraction by sub, max3 over pairs, one compare per row
    for px in range(8):
        for c in range(3):
            (s.cvt() if unpack == "cvt" else s.andb())
        for c in range(3): s.fma()
        s.andb(); s.subf()
        if px % 2: s.max3()
    s.cmp()
def colour_flt_pk(s):  # float form, FMAs packed two pixels at a time
    for px in range(0, 8, 2):
        for c in range(6): s.cvt()
        for c in range(3): s.pkfma()
        s.andb(); s.subf(); s.andb(); s.subf(); s.max3()
    s.cmp()
def colour_flt_half(s):  # float form, bytes 0 and 3 of each dword by and / shift (denormal operands), bytes 1, 2 by cvt
    for px in range(8):
        for c in range(3):
            (s.cvt() if (px * 3 + c) % 4 in (1, 2) else s.andb())
        for c in range(3): s.fma()
        s.andb(); s.subf()
        if px % 2: s.max3()
    s.cmp()
def colour_flt_den15(s):  # float form, every byte by shift/and at ONE scale: 6 integer ops per dword of 4 bytes
    for px in range(8):
        for c in range(3): s.andb()
        if px % 2 == 0:
            for c in range(3): s.lshr()   # 24 bytes = 6 dwords -> 12 extra shifts per row, 3 every other pixel
        for c in range(3): s.fma()
        s.andb(); s.subf()
        if px % 2: s.max3()
    s.cmp()
def row_int(s):
    for _ in range(12): s.addu()
    s.mul24(); s.mad24(); s.mad24(); s.addu(); s.mul24(); s.mad24(); s.mad24(); s.addu()
    s.addu(); s.addu(); s.mul24(); s.mad24(); s.mad24()
    for _ in range(6): s.subu()
    for _ in range(4): s.ashr()
    s.mul24(); s.ashr(); s.mul24(); s.ashr()
def row_flt(s):
    for _ in range(12): s.addf()
    s.mulf(); s.fma(); s.fma(); s.addf(); s.mulf(); s.fma(); s.fma(); s.addf()
    s.addf(); s.addf(); s.mulf(); s.fma(); s.fma()
    for _ in range(6): s.subf()
    for _ in range(4): s.floor()
    for _ in range(2): s.mulf(); s.cvti(); s.mul24(); s.ashr(); s.cvtf()
def col_flt(s):
    for _ in range(12): s.addf()
    s.fma(); s.fma(); s.fma()            # c0/c4
    s.addf(); s.fma(); s.fma(); s.fma()  # c2/c6
    s.addf(); s.mulf(); s.fma(); s.fma(); s.addf(); s.mulf(); s.fma(); s.fma()
    for _ in range(4): s.subf()
    s.addf(); s.addf(); s.subf()
    for _ in range(6): s.floor()
    for _ in range(2): s.mulf(); s.floor(); s.fma(); s.floor()
def col_int(s):
    for _ in range(12): s.addu()
    s.mul24(); s.mad24(); s.mad24(); s.addu(); s.mul24(); s.mad24(); s.mad24(); s.addu()
    s.addu(); s.addu(); s.mul24(); s.mad24(); s.mad24()
    for _ in range(6): s.subu()
    for _ in range(6): s.addu()
    for _ in range(6): s.ashr()
    s.ashr(); s.mad24(); s.ashr(); s.ashr(); s.mad24(); s.ashr()
def quant_int(s):  # 8 coefficients: cvt, mul, cvt (+ byte store not modelled)
    for _ in range(8): s.cvtf(); s.mulf(); s.cvti()
def quant_flt(s):
    for _ in range(8): s.mulf(); s.cvti()

variants = {}
def build(name, fns):
    s = S()
    for f in fns: f(s)
    variants[name] = s
build("int_cvt", [lambda s: colour_int(s, "cvt"), row_int, col_int, quant_int])
build("flt_cvt", [lambda s: colour_flt(s, "cvt"), row_flt, col_flt, quant_flt])
build("flt_pk", [colour_flt_pk, row_flt, col_flt, quant_flt])
build("flt_half", [colour_flt_half, row_flt, col_flt, quant_flt])
build("flt_den15", [colour_flt_den15, row_flt, col_flt, quant_flt])
build("int_den", [lambda s: colour_int(s, "and"), row_int, col_int, quant_int])
build("flt_den", [lambda s: colour_flt(s, "and"), row_flt, col_flt, quant_flt])
# the same instructions, shuffled within the whole row-step (what perfect mixing would give)
for nm in ("int_cvt", "flt_cvt"):
    s = S(); s.l = variants[nm].l[:]; random.shuffle(s.l); variants[nm + "_shuffled"] = s

# the same instructions again, ordered for the issue model of profiles/r02_valu_stream_mix.txt: every half-rate op
# (cvt / floor / min3 / cmp / mul24: "S") directly followed by a float op ("F"), integer ops ("I") only between float ops
def klass(op):
    o = op.split()[0]
    if o.startswith(("v_cvt", "v_floor", "v_max3", "v_cmp", "v_mul_i32_i24", "v_mad_i32_i24")): return "S"
    if o.endswith("_f32"): return "F"
    return "I"
def paired(ops, style):
    S_, F_, I_ = ([o for o in ops if klass(o) == k] for k in "SFI")
    out = []
    if style == "sf":          # S F S F ... then the remaining F with the I ops spread among them
        while S_ and F_: out += [S_.pop(0), F_.pop(0)]
        out += S_
        rest = F_[:]
        step = max(1, len(rest) // (len(I_) + 1))
        for i, o in enumerate(rest):
            out.append(o)
            if I_ and i % step == step - 1 and i + 1 < len(rest): out.append(I_.pop(0))
        out += I_
    elif style == "sff":       # S F F S F F ... (as many F as there are) then I at the end
        per = max(1, len(F_) // max(1, len(S_)))
        while S_:
            out.append(S_.pop(0))
            for _ in range(per):
                if F_: out.append(F_.pop(0))
        out += F_ + I_
    elif style == "si":        # worst case by the model: I ops next to S ops
        while S_ and I_: out += [S_.pop(0), I_.pop(0)]
        out += S_ + F_ + I_
    elif style == "blocks":    # all S, then all F, then all I
        out = S_ + F_ + I_
    return out
for st in ("sf", "sff", "si", "blocks"):
    s = S(); s.l = paired(variants["flt_cvt"].l, st); variants["flt_cvt_" + st] = s
# rotated copies of the natural order: waves of one SIMD run the same stream at different phases
ROT = 5
base_l = variants["flt_cvt"].l

out = ['// GENERATED by gen_synth_rows.py - synthetic per-row instruction streams, see that file.',
       '#include <hip/hip_runtime.h>', '#include <stdio.h>',
       'template <int P> __global__ __launch_bounds__(256) void k(int iters, unsigned *out) {',
       '    unsigned v[12]; for (int i = 0; i < 12; i++) v[i] = threadIdx.x + i; unsigned c0 = 12345, c1 = 77; unsigned long long w0 = threadIdx.x, w1 = 3, w2 = 0x3f8000003f800000ull, w3 = 5;']
names = list(variants)
for i, nm in enumerate(names):
    out.append(f'    if (P == {i}) for (int it = 0; it < iters; it++) asm volatile("{variants[nm].text()}" : ' +
               ", ".join(f'"+v"(v[{j}])' for j in range(12)) + ' : "v"(c0), "v"(c1), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");')
nrot = len(names)
out.append(f'    if (P == {nrot}) {{ const int ph = blockIdx.x % {ROT};')
for r in range(ROT):
    k = len(base_l) * r // ROT
    t = S(); t.l = base_l[k:] + base_l[:k]
    out.append(f'      if (ph == {r}) for (int it = 0; it < iters; it++) asm volatile("{t.text()}" : ' +
               ", ".join(f'"+v"(v[{j}])' for j in range(12)) + ' : "v"(c0), "v"(c1), "v"(w0), "v"(w1), "v"(w2), "v"(w3) : "vcc");')
out.append('    }')
out += ['    unsigned x = 0; for (int i = 0; i < 12; i++) x ^= v[i]; out[blockIdx.x * 256 + threadIdx.x] = x;', '}',
        'template <int P> void run(const char *name, int n, int waves, unsigned *d) {',
        '    const int iters = 2000, blocks = 256 * waves;', '    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);',
        '    hipLaunchKernelGGL((k<P>), dim3(blocks), dim3(256), 0, 0, 10, d); (void)hipEventRecord(e0);',
        '    hipLaunchKernelGGL((k<P>), dim3(blocks), dim3(256), 0, 0, iters, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);',
        '    float ms; (void)hipEventElapsedTime(&ms, e0, e1);',
        '    double ns = ms * 1e6 / ((double)waves * iters);',
        '    printf("%-20s %d waves/SIMD  %4d instr/row-step  %8.1f ns per row-step per wave per SIMD = %7.1f cycles @2.0GHz = %5.2f per instr\\n", name, waves, n, ns, ns * 2.0, ns * 2.0 / n);',
        '}', 'int main() {', '    unsigned *d; (void)hipMalloc(&d, 256 * 1024 * 8 * 4);', '    run<0>("warm", 1, 4, d);']
for w in (5,):
    for i, nm in enumerate(names):
        out.append(f'    run<{i}>("{nm}", {len(variants[nm].l)}, {w}, d);')
out.append(f'    run<{nrot}>("flt_cvt_rotated_per_block", {len(base_l)}, 5, d);')
for i, nm in enumerate(names):
    if nm.startswith("flt_cvt"): out.append(f'    run<{i}>("{nm} (1 wave)", {len(variants[nm].l)}, 1, d);')
out += ['    return 0;', '}']
open(__file__.replace("gen_synth_rows.py", "synth_rows.hip"), "w").write("\n".join(out) + "\n")
for nm in names: print(nm, len(variants[nm].l))
