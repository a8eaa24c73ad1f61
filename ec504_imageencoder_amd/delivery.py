"""Delivery of the frame records to the HOST, overlapped with the next batch's encode (north_star: the path ends in "a
contiguous host bitstream"; the reference writes it with bitvector_fwrite, include/encoder.h:445).

`HostDelivery` is the single-GPU analogue of sharding.StepPipeline: batch k+1 encodes on the main stream while batch k's
records travel device-to-host on a side stream into one of two pinned buffers.  The copy needs the batch's byte count on
the host, so once per step the host waits for ONE event — "the 16 bytes (total, status) of batch k have reached pinned
memory" — and at that moment the main stream already holds the encode of batch k+1.  Nothing is allocated inside the loop.
A batch that ran out of overflow scratch (M1V_STATUS_SCRATCH, recoverable) is encoded again with the worst case reserved, as
sharding.StepPipeline does; a batch with any other status bit is not delivered: step() raises.
"""
from . import _ffi


class HostDelivery:
    def __init__(self, enc, n_frames, capacity=None, n_buffers=2):
        import torch
        self.torch, self.enc = torch, enc
        dev = torch.device("cuda", enc.device)
        cap = int(capacity if capacity is not None else enc.default_out_capacity(n_frames))
        self.outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(n_buffers)]
        self.sizes = [torch.empty(max(n_frames, 1), dtype=torch.int64, device=dev) for _ in range(n_buffers)]
        self.metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(n_buffers)]
        self.host = [torch.empty(cap, dtype=torch.uint8, pin_memory=True) for _ in range(n_buffers)]
        self.host_meta = [torch.zeros(2, dtype=torch.int64, pin_memory=True) for _ in range(n_buffers)]
        self.side = torch.cuda.Stream(device=dev)
        self.encoded = [torch.cuda.Event() for _ in range(n_buffers)]
        self.counted = [torch.cuda.Event() for _ in range(n_buffers)]
        self.delivered = [None] * n_buffers
        self.args = [None] * n_buffers   # what buffer b was encoded from (for the scratch retry)
        self.pending, self.step_no = [], 0
        self.last = None            # (buffer index, total bytes) of the newest delivered batch
        self.bytes_delivered = 0

    def _deliver(self, b):
        torch = self.torch
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.encoded[b])
            self.host_meta[b].copy_(self.metas[b], non_blocking=True)
            self.counted[b].record(self.side)
            self.counted[b].synchronize()       # the step's only host wait; the next encode is already queued
            total, status = int(self.host_meta[b][0]), int(self.host_meta[b][1]) & 0xFFFFFFFF
            if status == _ffi.STATUS_SCRATCH and self.args[b] is not None:
                # recoverable: reserve the worst case (waits for the device), encode the same frames again on this stream
                rgb, first = self.args[b]
                self.enc.reserve_scratch(True)
                self.enc.encode(rgb, first, out=self.outs[b], sizes=self.sizes[b], meta=self.metas[b])
                self.host_meta[b].copy_(self.metas[b], non_blocking=True)
                self.side.synchronize()
                total, status = int(self.host_meta[b][0]), int(self.host_meta[b][1]) & 0xFFFFFFFF
            if status:
                raise RuntimeError(f"encode status {status:#x}: the batch's output is undefined and is not delivered")
            if total > self.host[b].numel():
                raise RuntimeError("pinned buffer too small")
            self.host[b][:total].copy_(self.outs[b][:total], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.side)
            self.delivered[b] = ev
        self.last = (b, total)
        self.bytes_delivered += total

    def step(self, rgb, first_frame_index=0):
        """Encode `rgb` (device-resident) into buffer b; deliver the previous batch behind it."""
        b = self.step_no % len(self.outs)
        self.step_no += 1
        if self.delivered[b] is not None:
            self.torch.cuda.current_stream().wait_event(self.delivered[b])   # buffer b has left for the host: free again
        self.enc.encode(rgb, first_frame_index, out=self.outs[b], sizes=self.sizes[b], meta=self.metas[b])
        self.encoded[b].record()
        self.args[b] = (rgb, first_frame_index)
        self.pending.append(b)
        if len(self.pending) > 1:
            self._deliver(self.pending.pop(0))

    def fence(self):
        while self.pending:
            self._deliver(self.pending.pop(0))
        self.torch.cuda.synchronize()

    def result(self):
        """After fence(): the newest delivered batch as a pinned uint8 tensor (a view of one of the two buffers)."""
        if self.last is None:
            return None
        b, total = self.last
        return self.host[b][:total]
