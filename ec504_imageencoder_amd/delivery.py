"""Delivery of the frame records to the HOST, overlapped with the next batch's encode (north_star: the path ends in "a
contiguous host bitstream"; the reference writes it with bitvector_fwrite, include/encoder.h:445).

The overlap lives in the C library (m1v_delivery_* in include/mpeg1_hip.h, csrc/m1v_kernels.hip): two device output buffers
and two pinned host buffers owned by the delivery object, the device-to-host copy of batch k on an internal stream under the
encode of batch k+1, one host wait per step for the 16 bytes (total, status) of the batch that is about to travel.  A batch
that ran out of overflow scratch (M1V_STATUS_SCRATCH) is encoded again with the worst case reserved; any other status bit
fails the step.  `HostDelivery` is the ctypes mirror of that object (a `main.c`-style C caller uses the same five functions).
"""
import ctypes as C

from . import _ffi
from .encoder import EncoderError, _ptr, _stream


class _Arrival:
    """What `HostDelivery.delivered[slot]` holds: synchronize() blocks until that slot's copy has reached the host."""

    def __init__(self, hd, slot):
        self.hd, self.slot = hd, slot

    def synchronize(self):
        self.hd._wait(self.slot)


class HostDelivery:
    def __init__(self, enc, n_frames, capacity=None, n_buffers=2):
        assert n_buffers == 2, "the library double-buffers"
        self.enc, self.n_frames = enc, n_frames
        cap = int(capacity if capacity is not None else enc.default_out_capacity(n_frames))
        self._h = C.c_void_p(0)
        rc = _ffi.lib().m1v_delivery_create(enc._h, cap, C.byref(self._h))
        if rc != _ffi.OK:
            self._h = C.c_void_p(0)
            raise EncoderError(rc, "m1v_delivery_create")
        self.capacity = cap
        self.delivered = [None, None]
        self.last = None            # (slot, total bytes) of the newest batch on its way / delivered
        self.bytes_delivered = 0
        self._keep = [None, None]   # the input of the batch in each slot: must outlive the start of its copy
        self._step = 0

    def close(self):
        if self._h:
            _ffi.lib().m1v_delivery_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _started(self, slot):
        if slot == _ffi.DELIVERY_NONE:
            return
        if slot < 0:
            raise EncoderError(slot, "m1v_delivery_step")
        self.delivered[slot] = _Arrival(self, slot)
        total = int(_ffi.lib().m1v_delivery_bytes(self._h, slot))   # on the host since the step's wait
        self.last = (slot, total)
        self.bytes_delivered += total

    def _wait(self, slot):
        host, nbytes, sizes = C.c_void_p(0), C.c_uint64(0), C.c_void_p(0)
        rc = _ffi.lib().m1v_delivery_wait(self._h, slot, C.byref(host), C.byref(nbytes), C.byref(sizes))
        if rc != _ffi.OK:
            raise EncoderError(rc, "m1v_delivery_wait")
        return host.value, int(nbytes.value), sizes.value

    def step(self, rgb, first_frame_index=0):
        """Encode `rgb` (device-resident) on the current stream; the batch before it starts travelling behind it."""
        slot_in = self._step & 1
        self._step += 1
        self._keep[slot_in] = rgb
        n = int(rgb.shape[0])
        self._started(_ffi.lib().m1v_delivery_step(self._h, _ptr(rgb), n, int(first_frame_index), _stream()))

    def fence(self):
        import torch
        self._started(_ffi.lib().m1v_delivery_flush(self._h))
        if self.last is not None:
            self._wait(self.last[0])
        torch.cuda.synchronize()

    def result(self):
        """The newest delivered batch as a uint8 tensor over the library's pinned buffer (waits for its arrival)."""
        if self.last is None:
            return None
        import numpy as np
        import torch
        host, total, _ = self._wait(self.last[0])
        buf = (C.c_uint8 * total).from_address(host)
        return torch.from_numpy(np.ctypeslib.as_array(buf))

    def frame_sizes(self, n):
        """Byte counts of the newest delivered batch's `n` frame records."""
        import numpy as np
        _, _, sizes = self._wait(self.last[0])
        return np.ctypeslib.as_array((C.c_uint64 * n).from_address(sizes)).copy()
