"""Host-side mirror of the reference's operator for this path.

The reference has one operator, `mpeg_encode_procedure(images_folder, bitstream_folder, video_path,
quality_factor)` (include/encoder.h:20), whose per-frame body is the hot path.  `Mpeg1Encoder` is
that body as a batched device operator: frames in (HBM-resident uint8 RGB), contiguous frame
records out, both as torch CUDA tensors.  torch is plumbing here (device memory, streams); all
arithmetic happens in libencoder.so's HIP kernels.
"""
import ctypes as C

from . import _ffi


class EncoderError(RuntimeError):
    def __init__(self, code, where):
        super().__init__(f"{where}: rc={code}: {_ffi.last_error()}")
        self.code = code


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Mpeg1Encoder:
    """One picture geometry + quality factor on one GPU.

    mode: "strict" = the 96x144 region the unmodified reference encodes (encoder.h:238,248),
          "full"   = every macroblock (the reference with its loop bounds restored).
    """

    def __init__(self, width, height, quality_factor=12, mode="full", channels=3, max_frames=300, device=0):
        self._h = C.c_void_p(0)
        self.width, self.height, self.channels = int(width), int(height), int(channels)
        self.quality_factor, self.max_frames, self.device = int(quality_factor), int(max_frames), int(device)
        self.mode = {"strict": _ffi.MODE_STRICT, "full": _ffi.MODE_FULL}[mode] if isinstance(mode, str) else int(mode)
        rc = _ffi.lib().m1v_create(C.byref(self._h), self.device, self.width, self.height, self.channels,
                                   self.quality_factor, self.mode, self.max_frames)
        if rc != _ffi.OK:
            self._h = C.c_void_p(0)
            raise EncoderError(rc, "m1v_create")
        L = _ffi.lib()
        self.strips, self.mb_rows = L.m1v_strips(self._h), L.m1v_mb_rows(self._h)
        self.frame_bound = L.m1v_frame_bound(self._h)
        self.frame_bytes_in = L.m1v_frame_bytes_in(self._h)
        self.blocks_per_frame = self.strips * self.mb_rows * 6

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _ffi.lib().m1v_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:        # at interpreter shutdown module globals may already be gone; the process is ending anyway
            self.close()
        except Exception:
            pass

    # ---- the hot path -------------------------------------------------------------------------
    def encode(self, rgb, first_frame_index=0, out=None, sizes=None, meta=None):
        """rgb: uint8 CUDA tensor [n, H, W, C] (contiguous).  Asynchronous on torch's current stream.
        Returns (out, sizes, meta): out uint8[cap] frame records back to back, sizes uint64-as-int64[n],
        meta int64[2] = (total bytes, status bits)."""
        import torch
        n = rgb.shape[0]
        assert rgb.is_cuda and rgb.dtype == torch.uint8 and rgb.is_contiguous()
        assert rgb.numel() == n * self.frame_bytes_in
        if out is None:
            out = torch.empty(self.default_out_capacity(n), dtype=torch.uint8, device=rgb.device)
        if sizes is None:
            sizes = torch.empty(max(n, 1), dtype=torch.int64, device=rgb.device)
        if meta is None:
            meta = torch.zeros(2, dtype=torch.int64, device=rgb.device)
        rc = _ffi.lib().m1v_encode_device(self._h, _ptr(rgb), n, int(first_frame_index), _ptr(out), out.numel(),
                                          _ptr(sizes), C.c_void_p(meta.data_ptr()), C.c_void_p(meta.data_ptr() + 8),
                                          _stream())
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_encode_device")
        return out, sizes, meta

    def set_pipelined(self, enable=True):
        """Overlap each batch's layout + gather (internal stream) with the next batch's encode kernel.
        Outputs of a batch are complete only behind flush(); callers double-buffer `out`."""
        rc = _ffi.lib().m1v_set_pipelined(self._h, 1 if enable else 0)
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_set_pipelined")

    def flush(self, stream=None):
        """Make `stream` (torch stream, default: the current one) wait for all pending gathers."""
        import torch
        st = stream if stream is not None else torch.cuda.current_stream()
        rc = _ffi.lib().m1v_flush(self._h, C.c_void_p(st.cuda_stream))
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_flush")

    def default_out_capacity(self, n):
        # typical output is far below the worst-case bound; callers that need the guarantee pass
        # `out` of frame_bound * n bytes.  NOSPACE is reported through the status word.
        return int(min(self.frame_bound, self.frame_bytes_in // 2 + 4096) * max(n, 1))

    def encode_to_bytes(self, rgb, first_frame_index=0):
        """Synchronous convenience: returns (bytes, [sizes])."""
        import torch
        out, sizes, meta = self.encode(rgb, first_frame_index)
        self.flush()
        torch.cuda.synchronize(rgb.device)
        total, status = (int(x) for x in meta.cpu())
        status &= 0xFFFFFFFF
        if status & _ffi.STATUS_UNENCODABLE:
            raise EncoderError(_ffi.E_UNENCODABLE, "encode: |level| >= 256 (the reference crashes on this input)")
        if status & (_ffi.STATUS_NOSPACE | _ffi.STATUS_SCRATCH):
            if status & _ffi.STATUS_SCRATCH:        # more runs overflowed their compact slot than the arena holds
                self.reserve_scratch(True)
            if status & _ffi.STATUS_NOSPACE:
                out = torch.empty(self.frame_bound * rgb.shape[0], dtype=torch.uint8, device=rgb.device)
            return self._retry_bytes(rgb, first_frame_index, out)
        return out[:total].cpu().numpy().tobytes(), [int(s) for s in sizes[:rgb.shape[0]].cpu()]

    def _retry_bytes(self, rgb, first_frame_index, out):
        import torch
        out, sizes, meta = self.encode(rgb, first_frame_index, out=out)
        self.flush()
        torch.cuda.synchronize(rgb.device)
        total, status = (int(x) for x in meta.cpu())
        status &= 0xFFFFFFFF
        if status & _ffi.STATUS_NOSPACE:            # scratch was the first obstacle, the output buffer is the second
            out = torch.empty(self.frame_bound * rgb.shape[0], dtype=torch.uint8, device=rgb.device)
            out, sizes, meta = self.encode(rgb, first_frame_index, out=out)
            self.flush()
            torch.cuda.synchronize(rgb.device)
            total, status = (int(x) for x in meta.cpu())
            status &= 0xFFFFFFFF
        if status & _ffi.STATUS_UNENCODABLE:
            raise EncoderError(_ffi.E_UNENCODABLE, "encode: |level| >= 256 (the reference crashes on this input)")
        if status:
            raise EncoderError(_ffi.E_NOSPACE if status & _ffi.STATUS_NOSPACE else _ffi.E_SCRATCH, "encode")
        return out[:total].cpu().numpy().tobytes(), [int(s) for s in sizes[:rgb.shape[0]].cpu()]

    def encode_host(self, rgb_np, first_frame_index=0, with_planes=False):
        """numpy uint8 [n,H,W,C] through the host-buffer entry point (PCIe inclusive).  with_planes: also return the
        uint8 [n,3,H*W] Y/Cb/Cr planes from the same upload (m1v_encode_planes_host: the image_<k>.bit content)."""
        import numpy as np
        rgb_np = np.ascontiguousarray(rgb_np, dtype=np.uint8)
        n = rgb_np.shape[0]
        cap = self.frame_bound * max(n, 1)
        out = np.empty(cap, np.uint8)
        sizes = np.zeros(max(n, 1), np.uint64)
        planes = np.empty((n, 3, self.height * self.width), np.uint8) if with_planes else None
        rc = _ffi.lib().m1v_encode_planes_host(self._h, rgb_np.ctypes.data, n, int(first_frame_index), out.ctypes.data,
                                               cap, sizes.ctypes.data, planes.ctypes.data if with_planes and n else None)
        if rc < 0:
            raise EncoderError(rc, "m1v_encode_planes_host")
        res = out[:rc].tobytes(), [int(s) for s in sizes[:n]]
        return res + (planes,) if with_planes else res

    # ---- partial pipelines --------------------------------------------------------------------
    def coefficients(self, rgb):
        """int16 [n, strips*mb_rows*6, 64] zigzag-ordered quantised levels (BASELINE config 2)."""
        import torch
        n = rgb.shape[0]
        out = torch.empty((n, self.blocks_per_frame, 64), dtype=torch.int16, device=rgb.device)
        rc = _ffi.lib().m1v_coefficients_device(self._h, _ptr(rgb), n, _ptr(out), _stream())
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_coefficients_device")
        return out

    def convert(self, rgb):
        """uint8 [n, 3, H*W]: Y, full-resolution Cb, full-resolution Cr."""
        import torch
        n = rgb.shape[0]
        out = torch.empty((n, 3, self.height * self.width), dtype=torch.uint8, device=rgb.device)
        rc = _ffi.lib().m1v_convert_device(self._h, _ptr(rgb), n, _ptr(out), _stream())
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_convert_device")
        return out

    def subsample(self, cb, cr):
        import torch
        n = (self.width // 2) * (self.height // 2)
        a = torch.empty(n, dtype=torch.uint8, device=cb.device)
        b = torch.empty(n, dtype=torch.uint8, device=cb.device)
        rc = _ffi.lib().m1v_subsample_device(self._h, _ptr(cb), _ptr(cr), _ptr(a), _ptr(b), _stream())
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_subsample_device")
        return a, b

    def synth(self, n_frames, seed=504, first_frame_index=0, device=None, out=None):
        """Device-generated synthetic frames (identical to oracle orc_synth_frame by definition).  out: reuse this tensor."""
        import torch
        dev = device or torch.device("cuda", self.device)
        rgb = out if out is not None else torch.empty((n_frames, self.height, self.width, self.channels), dtype=torch.uint8, device=dev)
        assert rgb.numel() == n_frames * self.frame_bytes_in and rgb.is_contiguous()
        rc = _ffi.lib().m1v_synth_device(_ptr(rgb), self.frame_bytes_in, n_frames, seed, first_frame_index, _stream())
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_synth_device")
        return rgb

    # ---- measurement ----------------------------------------------------------------------------
    def profile(self, enable=True):
        _ffi.lib().m1v_profile_enable(self._h, 1 if enable else 0)

    def profile_read(self):
        n, ms = C.c_int(0), C.c_double(0.0)
        rc = _ffi.lib().m1v_profile_read(self._h, C.byref(n), C.byref(ms))
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_profile_read")
        return n.value, ms.value

    def profile_read_times(self, cap=4096):
        """Durations (ms) of the dominant kernel's launches since profile(True), in launch order."""
        buf, n = (C.c_float * cap)(), C.c_int(0)
        rc = _ffi.lib().m1v_profile_read_times(self._h, buf, cap, C.byref(n))
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_profile_read_times")
        return [float(buf[i]) for i in range(min(n.value, cap))]

    def reserve_scratch(self, worst_case=True):
        """Size the overflow arena for every run (True) or return to the default 1/256 (False); see mpeg1_hip.h."""
        rc = _ffi.lib().m1v_reserve_scratch(self._h, 1 if worst_case else 0)
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_reserve_scratch")

    def scratch_bytes(self):
        return int(_ffi.lib().m1v_scratch_bytes(self._h))

    def debug_set_input_mode(self, mode):
        """Test hook: -1 automatic, 0 byte loads, 2 funnel-shifted 28-byte loads (see mpeg1_hip.h)."""
        rc = _ffi.lib().m1v_debug_set_input_mode(self._h, int(mode))
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_debug_set_input_mode")

    def debug_set_path(self, path):
        """Test hook: which encode kernel serves the batches: -1 by geometry, 0 runs, 1 tiles (see mpeg1_hip.h)."""
        rc = _ffi.lib().m1v_debug_set_path(self._h, {"auto": -1, "runs": 0, "tiles": 1}.get(path, path))
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_debug_set_path")

    @property
    def path(self):
        return "tiles" if _ffi.lib().m1v_path_in_use(self._h) == 1 else "runs"

    def debug_set_lds_words(self, words):
        rc = _ffi.lib().m1v_debug_set_lds_words(self._h, int(words))
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_debug_set_lds_words")

    def debug_set_dense_threads(self, threads):
        rc = _ffi.lib().m1v_debug_set_dense_threads(self._h, int(threads))
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_debug_set_dense_threads")


def file_prolog():
    buf = (C.c_uint8 * 27)()
    _ffi.lib().m1v_file_prolog(buf)
    return bytes(buf)


# ---- the coarse entry point, as the reference spells it --------------------------------------------
_LOAD_FN = C.CFUNCTYPE(C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int)
_FREE_FN = C.CFUNCTYPE(None, C.c_void_p)
_keepalive = {}


def set_image_loader(load):
    """Register a Python image loader with the library (include/encoder.h: encoder_set_image_loader).
    `load(path: str) -> numpy uint8 array [H, W, C]` or None.  C callers normally get stb_image registered by
    including encoder.h; Python callers can plug in any decoder (pixels then are that decoder's, not stb's)."""
    import numpy as np
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]

    def _load(path, pw, ph, pc, desired):
        arr = load(path.decode())
        if arr is None:
            return None
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        h, w, c = arr.shape
        buf = libc.malloc(arr.nbytes)
        C.memmove(buf, arr.ctypes.data, arr.nbytes)
        pw[0], ph[0], pc[0] = w, h, c
        return buf

    def _free(p):
        libc.free(p)

    cb = (_LOAD_FN(_load), _FREE_FN(_free))
    _keepalive["loader"] = cb
    _ffi.lib().encoder_set_image_loader(C.cast(cb[0], C.c_void_p), C.cast(cb[1], C.c_void_p))


def mpeg_encode_procedure(images_folder, bitstream_folder, video_path, quality_factor, region=None):
    """int mpeg_encode_procedure(images_folder, bitstream_folder, video_path, quality_factor) — the reference's
    one public operator (include/encoder.h:20), same arguments, same return codes (0 ok, 1 cannot open video,
    -1 folder/images/dimension problems).  region: None = library default (the reference's 96x144 corner unless
    EC504_ENCODE_REGION=full), "strict" or "full"."""
    L = _ffi.lib()
    args = [str(images_folder).encode(), str(bitstream_folder).encode(), str(video_path).encode(), int(quality_factor)]
    if region is None:
        return L.mpeg_encode_procedure(*args)
    return L.mpeg_encode_procedure_region(*args, 1 if region == "full" else 0)
