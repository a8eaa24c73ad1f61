"""CPU-side checks of the product library: it loads, exports every symbol the headers declare, and its
device code is what the design says (no fused fp64 multiply-add in the colour conversion).  No compute
calls — there is no GPU here."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ec504_imageencoder_amd", "libencoder.so")


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "ec504_imageencoder_amd", "csrc")], check=True)
    return LIB


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:m1v|mpeg|encoder)_[a-z0-9_]+)\s*\(", text)))


def test_exports_every_declared_symbol(built):
    import ctypes
    L = ctypes.CDLL(built)
    names = _declared("mpeg1_hip.h")
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), n
    from ec504_imageencoder_amd import _ffi
    assert sorted(_ffi.MPEG1_HIP_SYMBOLS) == names
    drop = _declared("encoder.h")
    assert drop == sorted(_ffi.ENCODER_H_SYMBOLS)
    for n in drop:
        assert hasattr(L, n), n


def test_no_gpu_means_loud_failure(built):
    from ec504_imageencoder_amd import EncoderError, Mpeg1Encoder, _ffi
    if _ffi.lib().m1v_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(EncoderError) as ei:
        Mpeg1Encoder(352, 288, 12, "full")
    assert ei.value.code == _ffi.E_NODEVICE


def test_file_prolog_bytes(built):
    from ec504_imageencoder_amd import file_prolog
    assert file_prolog() == bytes.fromhex("000001ba2100010001c33367" "000001bb0009c33367" "0021ffe0e0e6")


def test_device_code_is_gfx950_and_unfused(built):
    """The colour conversion must evaluate image_processing.c:104-106 without FMA contraction."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    bundler = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"
    if not (os.path.exists(objdump) and os.path.exists(bundler)):
        pytest.skip("llvm tools absent")
    obj = os.path.join(ROOT, "ec504_imageencoder_amd", "csrc", "m1v_kernels.o")
    if not os.path.exists(obj):
        pytest.skip("object file absent")
    import glob
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(obj, os.path.join(td, "k.o"))
        subprocess.run([objdump, "--offloading", "k.o"], cwd=td, capture_output=True, text=True)
        cos = glob.glob(os.path.join(td, "k.o.*gfx950*"))
        if not cos:
            pytest.skip("cannot extract the gfx950 code object")
        out = subprocess.run([objdump, "-d", cos[0]], capture_output=True, text=True)
    asm = out.stdout
    assert "v_mul_f64" in asm and "v_add_f64" in asm
    assert not re.search(r"v_fma(c)?_f64", asm), "fp64 FMA found: colour conversion would not be bit-exact"


def test_encode_kernels_do_not_spill(built):
    """The dominant kernel is vector-issue bound at 5 waves per SIMD: its speed hinges on fitting 96 VGPRs with no
    scratch (builds that spilled 130-200 B/lane measured 1.7x-2.4x slower).  Read the code object's metadata."""
    readobj = "/opt/rocm/lib/llvm/bin/llvm-readobj"
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    obj = os.path.join(ROOT, "ec504_imageencoder_amd", "csrc", "m1v_kernels.o")
    if not (os.path.exists(readobj) and os.path.exists(obj)):
        pytest.skip("llvm tools or object absent")
    import glob
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(obj, os.path.join(td, "k.o"))
        subprocess.run([objdump, "--offloading", "k.o"], cwd=td, capture_output=True, text=True)
        cos = glob.glob(os.path.join(td, "k.o.*gfx950*"))
        if not cos:
            pytest.skip("cannot extract the gfx950 code object")
        notes = subprocess.run([readobj, "--notes", cos[0]], capture_output=True, text=True).stdout
    kernels = {}
    name = None
    fields = {}
    for line in notes.splitlines():
        t = line.strip()
        if t.startswith("- .") or t.startswith("-   ."):
            if name:
                kernels[name] = fields
            name, fields = None, {}
            t = t.lstrip("- ").strip()
        m = re.match(r"\.(name|vgpr_count|private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count):\s*(\S+)", t)
        if m:
            if m.group(1) == "name":
                name = m.group(2)
            else:
                fields[m.group(1)] = int(m.group(2))
    if name:
        kernels[name] = fields
    dense = {k: v for k, v in kernels.items() if "k_encode_dense" in k}
    assert len(dense) == 8, sorted(kernels)          # input modes 0, 1, 2, 3 x narrow/wide staging
    for k, v in dense.items():
        assert v.get("vgpr_count", 999) <= 96, (k, v)
        # no input mode may touch scratch (round 2: the register-heavy modes recompute the colour sums in their rare
        # branch instead of keeping them alive, see convert_row's LEAN)
        assert v.get("private_segment_fixed_size", 0) == 0, (k, v)
    strips = {k: v for k, v in kernels.items() if "k_encode_strips" in k}
    assert len(strips) == 2
    for k, v in strips.items():
        assert v.get("private_segment_fixed_size", 0) == 0, (k, v)


def test_hot_kernel_consumes_pixel_rows_as_they_arrive(built):
    """The dense kernel issues its table loads and its sixteen pixel loads outside any branch, so the compiler can
    count them: the code tables are waited for with the pixels still in flight (vmcnt(16)) and every row is converted
    as soon as ITS loads are back (vmcnt(14), (12) ... (0)).  One branch around one load silently turns all of these
    into vmcnt(0) — correct, and 9 % slower — so the disassembly is checked."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    obj = os.path.join(ROOT, "ec504_imageencoder_amd", "csrc", "m1v_kernels.o")
    if not (os.path.exists(objdump) and os.path.exists(obj)):
        pytest.skip("llvm tools or object absent")
    import glob
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(obj, os.path.join(td, "k.o"))
        subprocess.run([objdump, "--offloading", "k.o"], cwd=td, capture_output=True, text=True)
        cos = glob.glob(os.path.join(td, "k.o.*gfx950*"))
        if not cos:
            pytest.skip("cannot extract the gfx950 code object")
        asm = subprocess.run([objdump, "-d", cos[0]], capture_output=True, text=True).stdout
    for variant in ("k_encode_denseILi1ELb1", "k_encode_denseILi1ELb0"):      # the two aligned-input (mode 1) variants
        m = re.search(r"<_ZN\S*%s\S*>:\n(.*?)\n\n" % variant, asm, re.S)
        assert m, variant
        waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", m.group(1))]
        assert 16 in waits, (variant, waits[:12])
        i = waits.index(16)
        seq = [w for w in waits[i:] if w % 2 == 0 and w <= 14]
        want = [14, 12, 10, 8, 6, 4, 2, 0]
        it = iter(seq)
        assert all(any(w == x for x in it) for w in want), (variant, waits[:40])


def test_hot_kernel_shape_of_round_2(built):
    """Round-2 properties of the dense kernel, read off the disassembly so that a refactor cannot lose them silently:
    the FDCT runs on the float pipe and no 24-bit integer multiply is left in the kernel (v_mul_i32_i24 / v_mad_i32_i24
    cost 20-45 cycles each inside a float stream, profiles/r02_stream_probe.txt; the sixteen (x * 181) >> 17 of the row
    pass use the 32-bit multiplier), the quantised levels leave as 64 LDS byte stores, and the waves of a
    workgroup meet only twice on the common path (scan, store) plus twice on the global-memory fallback (arena slot, clear)."""
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    obj = os.path.join(ROOT, "ec504_imageencoder_amd", "csrc", "m1v_kernels.o")
    if not (os.path.exists(objdump) and os.path.exists(obj)):
        pytest.skip("llvm tools or object absent")
    import glob
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(obj, os.path.join(td, "k.o"))
        subprocess.run([objdump, "--offloading", "k.o"], cwd=td, capture_output=True, text=True)
        cos = glob.glob(os.path.join(td, "k.o.*gfx950*"))
        if not cos:
            pytest.skip("cannot extract the gfx950 code object")
        asm = subprocess.run([objdump, "-d", cos[0]], capture_output=True, text=True).stdout
    m = re.search(r"<_ZN\S*k_encode_denseILi1ELb1\S*>:\n(.*?)\n\n", asm, re.S)
    assert m
    ops = [l.split()[0] for l in m.group(1).splitlines() if l.strip() and not l.strip().startswith(("/", ";"))]
    count = lambda name: sum(1 for o in ops if o.startswith(name))
    assert count("v_mad_i32_i24") == 0 and count("v_mul_i32_i24") == 0, (count("v_mad_i32_i24"), count("v_mul_i32_i24"))
    assert count("ds_write_b8") == 64
    assert count("v_floor_f32") == 112 and count("v_min3_f32") == 32
    assert count("s_barrier") <= 4, count("s_barrier")


def _gfx950_disassembly():
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    obj = os.path.join(ROOT, "ec504_imageencoder_amd", "csrc", "m1v_kernels.o")
    if not (os.path.exists(objdump) and os.path.exists(obj)):
        pytest.skip("llvm tools or object absent")
    import glob
    import shutil
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        shutil.copy(obj, os.path.join(td, "k.o"))
        subprocess.run([objdump, "--offloading", "k.o"], cwd=td, capture_output=True, text=True)
        cos = glob.glob(os.path.join(td, "k.o.*gfx950*"))
        if not cos:
            pytest.skip("cannot extract the gfx950 code object")
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readobj", "--notes", cos[0]], capture_output=True, text=True).stdout
        return subprocess.run([objdump, "-d", cos[0]], capture_output=True, text=True).stdout, notes


def test_tile_kernel_shape_of_round_3(built):
    """Round-3 properties of the tile kernel (csrc/m1v_tiles.h), read off the code object:
    * its pixels arrive by LDS-DMA only: 16 global_load_lds_dwordx4 (two per row-step) and 3 global_load_lds_dword (the
      wave's VLC table), no global_load into registers;
    * the ring is two row-steps deep and every row is read as soon as ITS two instructions have landed: the vmcnt waits
      of the row loop are 2, 2, 2, 2, 2, 2, 2, 0 (one wait per row, no branch between the wave kinds);
    * the waves of a tile meet twice on the common path (bit counts, image complete);
    * 96 VGPRs or fewer (5 waves per SIMD: the row-pass outputs stay unpacked, m1v_tiles.h M1V_TILE_KEEP) and NO scratch: a
      spill would also join the vmcnt queue the row loop counts on;
    * every LDS read issued from inline asm has its s_waitcnt in the same statement (nothing can sit between them)."""
    asm, notes = _gfx950_disassembly()
    for variant in ("k_encode_tilesILb1", "k_encode_tilesILb0"):
        m = re.search(r"<_ZN\S*%s\S*>:\n(.*?)\n\n" % variant, asm, re.S)
        assert m, variant
        lines = [l.split("//")[0].strip() for l in m.group(1).splitlines() if l.strip() and not l.strip().startswith(("/", ";"))]
        ops = [l.split()[0] for l in lines]
        count = lambda name: sum(1 for o in ops if o.startswith(name))
        assert count("global_load_lds_dwordx4") == 16 and count("global_load_lds_dword") == 19, variant
        assert count("global_load_dword") == 0 and count("scratch_") == 0 and count("flat_load") == 0, variant
        first_read = next(i for i, l in enumerate(lines) if l.startswith("ds_read_b64"))
        waits = [int(x) for l in lines[:first_read + 2500] for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", l)]
        assert waits[:8] == [2, 2, 2, 2, 2, 2, 2, 0], (variant, waits[:12])
        assert count("s_barrier") <= 4, (variant, count("s_barrier"))   # 2 on the common path + 2 on the arena fallback
        # asm LDS reads: the three ds_read_b64 of a row (offsets 0, 8, 16) and the ds_read2_b32 groups of the mask are each
        # followed directly by their wait
        rows = [i for i, l in enumerate(lines) if l.startswith("ds_read_b64") and l.endswith("offset:16")
                and lines[i - 1].startswith("ds_read_b64") and lines[i - 1].endswith("offset:8")]
        assert len(rows) == 8, (variant, len(rows))
        for i in rows:
            assert lines[i + 1].startswith("s_waitcnt lgkmcnt(0)"), (variant, lines[i:i + 2])
        groups = [i for i, l in enumerate(lines) if l.startswith("ds_read2_b32") and not lines[i + 1].startswith("ds_read2_b32")]
        assert groups, variant
        for i in groups:
            assert lines[i + 1].startswith("s_waitcnt lgkmcnt(0)"), (variant, lines[i:i + 2])
    recs = re.findall(r"\.name:\s*(\S*k_encode_tiles\S*).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+)", notes, re.S)
    assert len(recs) == 2, recs
    for nm, scratch, vgprs in recs:
        assert int(scratch) == 0 and int(vgprs) <= 96, (nm, scratch, vgprs)


def test_rounding_mode_of_the_pixel_stage(built):
    """The tile kernels take (x * 181) >> 17 as floor(t * 181/128) with the multiply rounded toward minus infinity
    (csrc/fdct_f32.h, fdct_row_f<float, true>), which is only right while the wave IS in that mode.  In the code object:
    * a kernel multiplies by 181/128 (literal 0x3fb50000) exactly when it switches MODE's fp32 rounding bits, once, to 2;
    * the switch comes before the first of those multiplies;
    * nothing of the compiler's integer-division expansion (v_rcp_iflag_f32, u32 <-> f32 conversions), whose error analysis
      assumes round-to-nearest, comes after the switch."""
    asm, _ = _gfx950_disassembly()
    kernels = re.findall(r"<(_ZN\S*)>:\n(.*?)\n\n", asm, re.S)
    assert kernels
    switching = []
    for name, body in kernels:
        lines = [l.split("//")[0].strip() for l in body.splitlines() if l.strip() and not l.strip().startswith(("/", ";"))]
        switches = [i for i, l in enumerate(lines) if l.startswith("s_setreg")]
        muls = [i for i, l in enumerate(lines) if l.startswith("v_mul_f32") and "0x3fb50000" in l]
        assert bool(switches) == bool(muls), name
        if not switches:
            continue
        switching.append(name)
        assert len(switches) == 1 and "HW_REG_MODE, 0, 2" in lines[switches[0]].replace("hwreg(", "").replace(")", ""), (name, [lines[i] for i in switches])
        assert lines[switches[0]].rstrip().endswith(", 2"), lines[switches[0]]
        assert len(muls) == 16 and switches[0] < muls[0], name
        late = [l for l in lines[switches[0]:] if l.startswith(("v_rcp", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_div_"))]
        assert not late, (name, late)
    assert sorted(n[n.index("k_"):][:18] for n in switching) == ["k_coefficient_tile", "k_encode_tilesILb0", "k_encode_tilesILb1"], switching



def test_assemble_kernel_shape_of_round_4(built):
    """Round-4 properties of k_assemble (csrc/m1v_assemble.h), read off the code object.  It replaced round 3's gather, whose
    lanes exchanged a table through wave-private LDS ordered by a release fence only (advisor, round 3): here every LDS value
    that one lane writes and another reads crosses a workgroup barrier.
    * the old kernels are gone from the library (k_gather_segments, k_tile_layout, k_frame_offsets, k_gather, k_frame_layout);
    * between the LDS writes of the placement table (ds_write_b128) and the first source load there is an s_barrier; the
      kernel has at least three (placement, image complete, per further chunk / pass);
    * placement is ds_or_b32 only (no returning LDS atomic, no LDS compare-and-swap), and no wave-scope fence is relied on;
    * 64 VGPRs or fewer (8 waves per SIMD) and no scratch; both instantiations (32-bit and 64-bit scratch offsets) exist."""
    asm, notes = _gfx950_disassembly()
    names = re.findall(r"<(_ZN\S*)>:", asm)
    for gone in ("k_gather_segments", "k_tile_layout", "k_frame_offsets", "k_frame_layout", "8k_gatherE"):
        assert not any(gone in n for n in names), gone
    for variant in ("k_assembleILb0", "k_assembleILb1"):
        m = re.search(r"<_ZN\S*%s\S*>:\n(.*?)\n\n" % variant, asm, re.S)
        assert m, variant
        lines = [l.split("//")[0].strip() for l in m.group(1).splitlines() if l.strip() and not l.strip().startswith(("/", ";"))]
        ops = [l.split()[0] for l in lines]
        table_writes = [i for i, o in enumerate(ops) if o == "ds_write_b128"]
        barriers = [i for i, o in enumerate(ops) if o == "s_barrier"]
        ors = [i for i, o in enumerate(ops) if o == "ds_or_b32"]
        source_loads = [i for i, o in enumerate(ops) if o == "global_load_dwordx4"]
        assert table_writes and barriers and ors and source_loads, variant
        # (in the code's order the scan of a LATER chunk follows the scatter: look at the first occurrence of each)
        first_src = min(i for i in source_loads if i > table_writes[0])
        assert any(table_writes[0] < b < first_src for b in barriers), variant
        # (the compiler lays the blocks of the pass / chunk loops out of source order, so only counts are checked for the second
        #  barrier: scatter -> barrier -> store loop is one of at least three barriers of the kernel; the GPU suite checks bytes)
        assert len(barriers) >= 3, variant
        assert not any(o.startswith(("ds_or_rtn", "ds_cmpst", "ds_cmpswap")) for o in ops), variant
    recs = re.findall(r"\.name:\s*(\S*k_assemble\S*).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+)", notes, re.S)
    assert len(recs) == 2, recs
    for nm, scratch, vgprs in recs:
        assert int(scratch) == 0 and int(vgprs) <= 64, (nm, scratch, vgprs)
