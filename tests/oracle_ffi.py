"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference's hot path (oracle/mpeg1_oracle.c).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
MODE_STRICT, MODE_FULL = 0, 1
E_ARG, E_UNENCODABLE, E_NOSPACE = -1, -2, -3

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


class OrcBits(C.Structure):
    _fields_ = [("buf", _u8p), ("cap_bytes", C.c_size_t), ("nbits", C.c_size_t)]


def build():
    """(Re)build liboracle.so with gcc; cheap, idempotent."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        src = os.path.join(ORACLE_DIR, "mpeg1_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            build()
        L = C.CDLL(so)
        L.orc_region.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_convert_rgb_to_ycbcr.argtypes = [_u8p, C.c_int, C.c_size_t, _u8p, _u8p, _u8p]
        L.orc_subsample_420.argtypes = [_u8p, _u8p, C.c_int, C.c_int, _u8p, _u8p]
        L.orc_fdct.argtypes = [_u8p, _i32p]
        L.orc_scale_qmatrix.argtypes = [C.c_int, _i32p]
        L.orc_quant_zigzag.argtypes = [_i32p, _i32p, _i32p]
        L.orc_run_length.argtypes = [_i32p, _i32p]
        L.orc_run_length.restype = C.c_int
        L.orc_bits_init.argtypes = [C.POINTER(OrcBits)]
        L.orc_bits_free.argtypes = [C.POINTER(OrcBits)]
        L.orc_bits_put.argtypes = [C.POINTER(OrcBits), C.c_uint32, C.c_int]
        L.orc_encode_block.argtypes = [C.c_int, _i32p, C.POINTER(OrcBits)]
        L.orc_encode_block.restype = C.c_int
        L.orc_frame_coefficients.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _i32p]
        L.orc_frame_coefficients.restype = C.c_int
        L.orc_encode_frame.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       _u8p, C.c_size_t]
        L.orc_encode_frame.restype = C.c_long
        L.orc_frame_bound.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_frame_bound.restype = C.c_size_t
        L.orc_file_prolog.argtypes = [_u8p]
        L.orc_file_prolog.restype = C.c_size_t
        L.orc_encode_frames.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, _u8p, C.c_size_t, _u64p]
        L.orc_encode_frames.restype = C.c_long
        L.orc_write_bit_file.argtypes = [C.c_char_p, _u8p, _u8p, _u8p, C.c_int, C.c_int]
        L.orc_write_bit_file.restype = C.c_int
        L.orc_synth_frame.argtypes = [_u8p, C.c_size_t, C.c_uint64, C.c_uint64]
        _lib = L
    return _lib


def _p8(a):
    return a.ctypes.data_as(_u8p)


def _p32(a):
    return a.ctypes.data_as(_i32p)


def region(mode, W, H):
    xe, ye = C.c_int(), C.c_int()
    lib().orc_region(mode, W, H, C.byref(xe), C.byref(ye))
    return xe.value, ye.value


def n_blocks(mode, W, H):
    xe, ye = region(mode, W, H)
    return (xe // 16) * (ye // 16) * 6


def convert(rgb, channels=3):
    """rgb: uint8 [..., channels] -> (Y, Cb, Cr) flat uint8 arrays."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    npx = rgb.size // channels
    Y, Cb, Cr = (np.empty(npx, np.uint8) for _ in range(3))
    lib().orc_convert_rgb_to_ycbcr(_p8(rgb), channels, npx, _p8(Y), _p8(Cb), _p8(Cr))
    return Y, Cb, Cr


def subsample(Cb, Cr, W, H):
    a = np.empty((H // 2) * (W // 2), np.uint8)
    b = np.empty_like(a)
    lib().orc_subsample_420(_p8(np.ascontiguousarray(Cb)), _p8(np.ascontiguousarray(Cr)), W, H, _p8(a), _p8(b))
    return a, b


def fdct(block_u8):
    blk = np.ascontiguousarray(block_u8, dtype=np.uint8).reshape(64)
    out = np.empty(64, np.int32)
    lib().orc_fdct(_p8(blk), _p32(out))
    return out


def scale_qmatrix(qf):
    q = np.empty(64, np.int32)
    lib().orc_scale_qmatrix(int(qf), _p32(q))
    return q


def quant_zigzag(dct, q):
    zz = np.empty(64, np.int32)
    lib().orc_quant_zigzag(_p32(np.ascontiguousarray(dct, np.int32)), _p32(np.ascontiguousarray(q, np.int32)), _p32(zz))
    return zz


def run_length(zz):
    pairs = np.zeros(130, np.int32)
    n = lib().orc_run_length(_p32(np.ascontiguousarray(zz, np.int32)), _p32(pairs))
    return pairs, n


def encode_block_bits(is_luma, zz):
    """Returns (rc, bitstring) for one block (DC/AC/EOB)."""
    b = OrcBits()
    L = lib()
    L.orc_bits_init(C.byref(b))
    rc = L.orc_encode_block(int(is_luma), _p32(np.ascontiguousarray(zz, np.int32)), C.byref(b))
    nb = b.nbits
    raw = bytes(bytearray(b.buf[i] for i in range((nb + 7) // 8)))
    L.orc_bits_free(C.byref(b))
    s = "".join(f"{x:08b}" for x in raw)[:nb]
    return rc, s


def frame_coefficients(rgb, W, H, qf, mode, channels=3):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    out = np.empty(n_blocks(mode, W, H) * 64, np.int32)
    rc = lib().orc_frame_coefficients(_p8(rgb), W, H, channels, qf, mode, _p32(out))
    if rc != 0:
        raise ValueError(f"orc_frame_coefficients rc={rc}")
    return out.reshape(-1, 64)


def frame_bound(W, H, mode):
    return lib().orc_frame_bound(W, H, mode)


def encode_frame(rgb, W, H, frame_index, qf, mode, channels=3):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    cap = frame_bound(W, H, mode)
    out = np.empty(cap, np.uint8)
    n = lib().orc_encode_frame(_p8(rgb), W, H, channels, frame_index, qf, mode, _p8(out), cap)
    if n < 0:
        raise ValueError(f"orc_encode_frame rc={n}")
    return out[:n].tobytes()


def file_prolog():
    out = np.empty(27, np.uint8)
    lib().orc_file_prolog(_p8(out))
    return out.tobytes()


def encode_frames(rgb, n_frames, W, H, first_index, qf, mode, channels=3, threads=1, cap=None):
    """rgb: contiguous n_frames*W*H*channels bytes -> (bytes, sizes[n_frames])."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    assert rgb.size == n_frames * W * H * channels
    if cap is None:
        # typical output is far below the worst case bound; retry bigger on NOSPACE
        cap = max(1 << 20, n_frames * (W * H // 4 + 4096))
    sizes = np.zeros(max(n_frames, 1), np.uint64)
    while True:
        out = np.empty(cap, np.uint8)
        n = lib().orc_encode_frames(_p8(rgb), n_frames, W, H, channels, first_index, qf, mode, threads,
                                    _p8(out), cap, sizes.ctypes.data_as(_u64p))
        if n == E_NOSPACE:
            cap *= 4
            continue
        if n < 0:
            raise ValueError(f"orc_encode_frames rc={n}")
        return out[:n].tobytes(), sizes[:n_frames].copy()


def encode_sequence(rgb, n_frames, W, H, qf, mode, channels=3, threads=1):
    """Whole .mpeg file image: PACK SYS + frames 0..n-1."""
    body, _ = encode_frames(rgb, n_frames, W, H, 0, qf, mode, channels, threads)
    return file_prolog() + body


def synth_frames(n_frames, W, H, seed=504, first_index=0, channels=3):
    nbytes = W * H * channels
    out = np.empty((n_frames, nbytes), np.uint8)
    for f in range(n_frames):
        lib().orc_synth_frame(_p8(out[f]), nbytes, seed, first_index + f)
    return out.reshape(n_frames, H, W, channels)
