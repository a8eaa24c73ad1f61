"""ctypes binding of the REAL reference, built by `make -C oracle _ref` from /root/reference's own
sources (oracle/_ref/, git-ignored).  TEST INFRASTRUCTURE ONLY: used to pin the oracle and to
generate tests/golden/.  Prototypes follow /root/reference/include/{image_processing.h:8-30,
bit_vector.h:9-42,mpeg1_blk.h:6-12,jpeg_handler.h:6-11}.

The reference prints to stdout from most functions; `quiet()` redirects fd 1 to /dev/null.
"""
import contextlib
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
REF_SRC = "/root/reference"

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class BitVector(C.Structure):  # bit_vector.h:9-14
    _fields_ = [("value", C.POINTER(C.c_char)), ("bits", C.c_longlong), ("cursor", C.c_longlong),
                ("cap", C.c_longlong)]


class Image(C.Structure):  # jpeg_handler.h:6-11
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("channels", C.c_int), ("data", _u8p)]


def available():
    return os.path.exists(os.path.join(REF_DIR, "libencoder_ref.so"))


def ensure_built():
    """Build oracle/_ref when the reference sources are present (authoring container)."""
    if os.path.exists(os.path.join(REF_SRC, "include", "encoder.h")):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref"], check=True,
                       stdout=subprocess.DEVNULL)
    return available()


@contextlib.contextmanager
def quiet():
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    try:
        yield
    finally:
        _libc().fflush(None)
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)


_c = None


def _libc():
    global _c
    if _c is None:
        _c = C.CDLL(None)
        _c.free.argtypes = [C.c_void_p]
    return _c


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(REF_DIR, "libencoder_ref.so"))
        L.fast_DCT.argtypes = [_u8p, _f64p]
        L.scale_quantization_matrix.argtypes = [_i32p, C.c_int]
        L.quantization.argtypes = [_f64p, _i32p, C.c_int]
        L.zigzag_scanning.argtypes = [_i32p, _i32p]
        L.run_length_encode.argtypes = [_i32p, _i32p]
        L.run_length_encode.restype = C.c_void_p
        L.bitvector_new.argtypes = [C.c_char_p, C.c_longlong]
        L.bitvector_new.restype = C.POINTER(BitVector)
        L.encode_block_header_i.argtypes = [C.c_ubyte, _i32p, C.POINTER(BitVector)]
        L.encode_block_end.argtypes = [C.POINTER(BitVector)]
        L.encode_macroblock_header_i.argtypes = [C.c_uint, C.c_short, C.POINTER(BitVector)]
        L.mpeg1_slice.argtypes = [C.c_uint8, C.c_uint8, C.POINTER(BitVector)]
        L.convert_rgb_to_ycbcr.argtypes = [C.POINTER(Image), C.POINTER(_u8p), C.POINTER(_u8p), C.POINTER(_u8p)]
        L.subsampling_420.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.POINTER(_u8p), C.POINTER(_u8p)]
        _lib = L
    return _lib


def _bv_string(bv):
    n = bv.contents.cap
    raw = C.string_at(bv.contents.value, (n + 7) // 8)
    return "".join(f"{x:08b}" for x in raw)[:n]


def fast_dct(block_u8):
    blk = np.ascontiguousarray(block_u8, np.uint8).reshape(64)
    out = np.empty(64, np.float64)
    lib().fast_DCT(blk.ctypes.data_as(_u8p), out.ctypes.data_as(_f64p))
    return out


def scale_qmatrix(qf):
    q = np.empty(64, np.int32)
    lib().scale_quantization_matrix(q.ctypes.data_as(_i32p), int(qf))
    return q


def quant_zigzag(dct_f64, qf):
    d = np.ascontiguousarray(dct_f64, np.float64)
    qz = np.empty(64, np.int32)
    zz = np.empty(64, np.int32)
    lib().quantization(d.ctypes.data_as(_f64p), qz.ctypes.data_as(_i32p), int(qf))
    lib().zigzag_scanning(qz.ctypes.data_as(_i32p), zz.ctypes.data_as(_i32p))
    return zz


def run_length(zz):
    pairs = np.full(136, 0x5A5A5A5A, np.int32)  # reference overflows int[128] by 2 for 64 non-zeros
    z = np.ascontiguousarray(zz, np.int32)
    lib().run_length_encode(z.ctypes.data_as(_i32p), pairs.ctypes.data_as(_i32p))
    return pairs


def block_bits(is_luma, zz):
    """run_length_encode -> encode_block_header_i -> encode_block_end on a fresh bit vector."""
    L = lib()
    pairs = run_length(zz)
    with quiet():
        bv = L.bitvector_new(b"", 8)
        L.encode_block_header_i(int(is_luma), pairs.ctypes.data_as(_i32p), bv)
        L.encode_block_end(bv)
    s = _bv_string(bv)
    _libc().free(C.cast(bv.contents.value, C.c_void_p))
    return s


def slice_and_mb_bits(strip):
    L = lib()
    with quiet():
        bv = L.bitvector_new(b"", 8)
        L.mpeg1_slice(1, strip & 0xFF, bv)
        L.encode_macroblock_header_i(1, 1, bv)
    return _bv_string(bv)


def convert(rgb, channels=3):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    npx = rgb.size // channels
    img = Image(npx, 1, channels, rgb.ctypes.data_as(_u8p))
    y, cb, cr = _u8p(), _u8p(), _u8p()
    with quiet():
        lib().convert_rgb_to_ycbcr(C.byref(img), C.byref(y), C.byref(cb), C.byref(cr))
    out = []
    for p in (y, cb, cr):
        out.append(np.ctypeslib.as_array(p, shape=(npx,)).copy())
        _libc().free(C.cast(p, C.c_void_p))
    return out


def subsample(Cb, Cr, W, H):
    a, b = _u8p(), _u8p()
    cb = np.ascontiguousarray(Cb, np.uint8)
    cr = np.ascontiguousarray(Cr, np.uint8)
    lib().subsampling_420(cb.ctypes.data_as(_u8p), cr.ctypes.data_as(_u8p), W, H, C.byref(a), C.byref(b))
    n = (W // 2) * (H // 2)
    out = []
    for p in (a, b):
        out.append(np.ctypeslib.as_array(p, shape=(n,)).copy())
        _libc().free(C.cast(p, C.c_void_p))
    return out


def run_encoder(images_dir, bit_dir, video_path, qf, mode="strict"):
    """Runs the reference driver binary (stdout discarded).  bit_dir must exist (encoder.h:75 opens
    the video before it would mkdir the folder)."""
    exe = os.path.join(REF_DIR, f"ref_encoder_{mode}")
    return subprocess.run([exe, images_dir, bit_dir, video_path, str(qf)], stdout=subprocess.DEVNULL).returncode


def dump_rgb(images_dir, prefix):
    """stb decode in the driver's enumeration order -> (names, [HxWxC uint8 arrays])."""
    import struct
    subprocess.run([os.path.join(REF_DIR, "ref_dump_rgb"), images_dir, prefix], check=True)
    names = open(prefix + ".order").read().split("\n")[:-1]
    raw = open(prefix + ".rgb", "rb").read()
    n = struct.unpack_from("<i", raw, 0)[0]
    off, frames = 4, []
    for _ in range(n):
        w, h, c = struct.unpack_from("<iii", raw, off)
        off += 12
        frames.append(np.frombuffer(raw, np.uint8, w * h * c, off).reshape(h, w, c).copy())
        off += w * h * c
    return names, frames
