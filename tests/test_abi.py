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
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n
    from ec504_imageencoder_amd import _ffi
    assert sorted(_ffi.MPEG1_HIP_SYMBOLS) == names


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
    out = subprocess.run([objdump, "--offloading", "-d", obj], capture_output=True, text=True)
    if "v_mul_f64" not in out.stdout:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            co = os.path.join(td, "dev.co")
            r = subprocess.run([bundler, "--type=o", "--unbundle", f"--input={obj}", f"--output={co}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], capture_output=True, text=True)
            if r.returncode != 0:
                pytest.skip("cannot unbundle: " + r.stderr[:200])
            out = subprocess.run([objdump, "-d", co], capture_output=True, text=True)
    asm = out.stdout
    assert "v_mul_f64" in asm and "v_add_f64" in asm
    assert not re.search(r"v_fma(c)?_f64", asm), "fp64 FMA found: colour conversion would not be bit-exact"
