"""ctypes binding of the in-tree libencoder.so (include/mpeg1_hip.h + include/encoder.h).

The library is the product: hand-written HIP kernels for gfx950 behind a C-ABI.  There is no CPU
fallback — if the shared object is missing or cannot be loaded this module raises.
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# EC504_LIBENCODER: a library variant (tools/mkvariant.sh) instead of the in-tree build, for the tuning tools
LIB_PATH = os.environ.get("EC504_LIBENCODER") or os.path.join(PKG_DIR, "libencoder.so")

MODE_STRICT, MODE_FULL = 0, 1
OK, E_ARG, E_UNENCODABLE, E_NOSPACE, E_HIP, E_NODEVICE, E_SCRATCH = 0, -1, -2, -3, -4, -5, -6
STATUS_UNENCODABLE, STATUS_NOSPACE, STATUS_SCRATCH = 1, 2, 4

_u8p = C.POINTER(C.c_uint8)

# every symbol include/mpeg1_hip.h declares (tests check that the library exports all of them)
MPEG1_HIP_SYMBOLS = [
    "m1v_device_count", "m1v_warm_up", "m1v_last_error", "m1v_create", "m1v_destroy", "m1v_strips", "m1v_mb_rows",
    "m1v_frame_bound", "m1v_frame_bound_for", "m1v_frame_bytes_in", "m1v_file_prolog", "m1v_encode_device", "m1v_encode_host",
    "m1v_encode_planes_host",
    "m1v_set_pipelined", "m1v_flush", "m1v_alloc_host", "m1v_free_host", "m1v_alloc_device", "m1v_free_device",
    "m1v_coefficients_device", "m1v_convert_device", "m1v_convert_host", "m1v_subsample_device", "m1v_synth_device",
    "m1v_profile_enable", "m1v_profile_read", "m1v_profile_read_times", "m1v_debug_set_lds_words", "m1v_debug_set_dense_threads",
    "m1v_debug_set_input_mode", "m1v_reserve_scratch", "m1v_scratch_bytes", "m1v_debug_set_path", "m1v_path_in_use", "m1v_debug_fail_alloc",
    "m1v_delivery_create", "m1v_delivery_destroy", "m1v_delivery_step", "m1v_delivery_flush", "m1v_delivery_wait", "m1v_delivery_bytes",
]
DELIVERY_NONE = 2


ENCODER_H_SYMBOLS = ["mpeg_encode_procedure", "mpeg_encode_procedure_region", "encoder_set_image_loader",
                     "encoder_set_host_threads", "encoder_release_cache"]


class EncoderLibraryMissing(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EncoderLibraryMissing(
            f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.m1v_device_count.restype = C.c_int
    L.m1v_last_error.restype = C.c_char_p
    L.m1v_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.m1v_create.restype = C.c_int
    L.m1v_destroy.argtypes = [vp]
    L.m1v_destroy.restype = None
    for name in ("m1v_strips", "m1v_mb_rows"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.c_int
    for name in ("m1v_frame_bound", "m1v_frame_bytes_in"):
        getattr(L, name).argtypes = [vp]
        getattr(L, name).restype = C.c_size_t
    L.m1v_file_prolog.argtypes = [_u8p]
    L.m1v_file_prolog.restype = C.c_size_t
    L.m1v_encode_device.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_size_t, vp, vp, vp, vp]
    L.m1v_encode_device.restype = C.c_int
    L.m1v_set_pipelined.argtypes = [vp, C.c_int]
    L.m1v_set_pipelined.restype = C.c_int
    L.m1v_flush.argtypes = [vp, vp]
    L.m1v_flush.restype = C.c_int
    L.m1v_alloc_host.argtypes = [C.c_size_t]
    L.m1v_alloc_host.restype = vp
    L.m1v_free_host.argtypes = [vp]
    L.m1v_free_host.restype = None
    L.m1v_warm_up.argtypes = [C.c_int]
    L.m1v_warm_up.restype = C.c_int
    L.m1v_encode_host.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_size_t, vp]
    L.m1v_encode_host.restype = C.c_long
    L.m1v_encode_planes_host.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_size_t, vp, vp]
    L.m1v_encode_planes_host.restype = C.c_long
    L.m1v_coefficients_device.argtypes = [vp, vp, C.c_int, vp, vp]
    L.m1v_coefficients_device.restype = C.c_int
    L.m1v_convert_device.argtypes = [vp, vp, C.c_int, vp, vp]
    L.m1v_convert_device.restype = C.c_int
    L.m1v_convert_host.argtypes = [vp, vp, C.c_int, vp]
    L.m1v_convert_host.restype = C.c_int
    L.m1v_subsample_device.argtypes = [vp, vp, vp, vp, vp, vp]
    L.m1v_subsample_device.restype = C.c_int
    L.m1v_synth_device.argtypes = [vp, C.c_size_t, C.c_int, C.c_uint64, C.c_uint64, vp]
    L.m1v_synth_device.restype = C.c_int
    L.m1v_profile_enable.argtypes = [vp, C.c_int]
    L.m1v_profile_enable.restype = C.c_int
    L.m1v_profile_read.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.m1v_profile_read.restype = C.c_int
    L.m1v_profile_read_times.argtypes = [vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
    L.m1v_profile_read_times.restype = C.c_int
    L.m1v_debug_set_input_mode.argtypes = [vp, C.c_int]
    L.m1v_debug_set_input_mode.restype = C.c_int
    L.m1v_reserve_scratch.argtypes = [vp, C.c_int]
    L.m1v_reserve_scratch.restype = C.c_int
    L.m1v_scratch_bytes.argtypes = [vp]
    L.m1v_scratch_bytes.restype = C.c_size_t
    L.m1v_debug_set_lds_words.argtypes = [vp, C.c_int]
    L.m1v_debug_set_lds_words.restype = C.c_int
    L.m1v_debug_set_dense_threads.argtypes = [vp, C.c_int]
    L.m1v_debug_set_dense_threads.restype = C.c_int
    L.m1v_debug_set_path.argtypes = [vp, C.c_int]
    L.m1v_debug_set_path.restype = C.c_int
    L.m1v_path_in_use.argtypes = [vp]
    L.m1v_path_in_use.restype = C.c_int
    L.m1v_debug_fail_alloc.argtypes = [C.c_int]
    L.m1v_debug_fail_alloc.restype = None
    L.m1v_delivery_create.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.m1v_delivery_create.restype = C.c_int
    L.m1v_delivery_destroy.argtypes = [vp]
    L.m1v_delivery_destroy.restype = None
    L.m1v_delivery_step.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.m1v_delivery_step.restype = C.c_int
    L.m1v_delivery_flush.argtypes = [vp]
    L.m1v_delivery_flush.restype = C.c_int
    L.m1v_delivery_wait.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(vp)]
    L.m1v_delivery_wait.restype = C.c_int
    L.m1v_delivery_bytes.argtypes = [vp, C.c_int]
    L.m1v_delivery_bytes.restype = C.c_uint64
    L.mpeg_encode_procedure.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
    L.mpeg_encode_procedure.restype = C.c_int
    L.mpeg_encode_procedure_region.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.mpeg_encode_procedure_region.restype = C.c_int
    L.encoder_set_image_loader.argtypes = [C.c_void_p, C.c_void_p]
    L.encoder_set_image_loader.restype = None
    _lib = L
    return L


def last_error():
    return lib().m1v_last_error().decode(errors="replace")
