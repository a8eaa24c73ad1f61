"""Multi-GPU sharding of the path: frames are independent given their global index, so rank r
encodes a contiguous block of frames and the per-rank bitstreams are gathered on rank 0 — the path's only
exchange.  One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node; "gloo"
in the CPU tests).  No reduction anywhere.

`StepPipeline` is the step loop `bench.py` runs (and the CPU tests drive with a fake encoder): encode of step k on
the main stream, exchange of step k-1 on a side stream, two output buffers.  The exchange needs each rank's byte
count on the host (point-to-point transfers take host-side sizes), so once per step the host waits for ONE event:
"the 8-byte counts of step k-1 have reached pinned memory".  At that moment the main stream already holds the
encode of step k, so the GPU is never idle behind the wait; nothing else in a step synchronises, allocates or
copies through pageable memory.

Two transports for the blobs:
  "xgmi"  every rank > 0 sends its blob to rank 0 with grouped send/recv (RCCL: 7 peers, 7 xGMI links).  Rank 0
          ingests (world - 1) x ~34 MB per 300 x 1080p step.
  "host"  every rank copies its blob device-to-host into ITS slice of one pinned host buffer shared by the ranks of
          the node (offsets = exclusive prefix of the counts).  No incast on rank 0: the copies ride each GPU's own
          PCIe link, and the product of the path is a host bitstream anyway.
"""
import contextlib
import os


def shard_range(n_frames, world, rank):
    """Contiguous, balanced split: returns (first_global_index, count) of this rank."""
    base, extra = divmod(n_frames, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def gather_bitstreams(blob, nbytes, dst=None, group=None):
    """One-shot form of the exchange (synchronous): blob holds this rank's frame records in its first `nbytes`
    bytes (device = the backend's device).  Returns on rank 0 a uint8 tensor with all ranks' records in rank
    (= frame) order and the list of per-rank byte counts; on other ranks (None, counts)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.tensor([int(nbytes)], dtype=torch.int64, device=blob.device)
    totals = torch.empty(world, dtype=torch.int64, device=blob.device)
    dist.all_gather_into_tensor(totals, mine, group=group)
    counts = [int(x) for x in totals.cpu()]
    if world == 1:
        return blob[:counts[0]], counts
    if rank == 0:
        need = sum(counts)
        if dst is None or dst.numel() < need:
            dst = torch.empty(need, dtype=torch.uint8, device=blob.device)
        dst[:counts[0]].copy_(blob[:counts[0]])
    for w in _post_blobs(blob, counts, dst, rank, group):
        w.wait()
    return (dst[:sum(counts)], counts) if rank == 0 else (None, counts)


def _post_blobs(blob, counts, dst, rank, group):
    """Grouped send/recv of the per-rank blobs to rank 0; returns the work handles (empty list: nothing to move)."""
    import torch.distributed as dist
    ops = []
    if rank == 0:
        off = counts[0]
        for r in range(1, len(counts)):
            if counts[r]:
                ops.append(dist.P2POp(dist.irecv, dst[off:off + counts[r]], r, group=group))
            off += counts[r]
    elif counts[rank]:
        ops.append(dist.P2POp(dist.isend, blob[:counts[rank]], 0, group=group))
    return dist.batch_isend_irecv(ops) if ops else []


class _CudaRuntime:
    """Streams and events of torch.cuda (HIP on ROCm)."""

    def __init__(self, device):
        import torch
        self.torch, self.device = torch, device

    def side_stream(self):
        return self.torch.cuda.Stream(device=self.device)

    def event(self):
        return self.torch.cuda.Event()

    def record(self, ev, stream=None):
        ev.record(stream if stream is not None else self.torch.cuda.current_stream(self.device))

    def wait(self, ev, stream=None):
        (stream if stream is not None else self.torch.cuda.current_stream(self.device)).wait_event(ev)

    def host_wait(self, ev):
        ev.synchronize()

    def on(self, stream):
        return self.torch.cuda.stream(stream)

    def synchronize(self):
        self.torch.cuda.synchronize(self.device)

    def pinned(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, pin_memory=True)


class _HostRuntime:
    """The same interface without a device (CPU tests over gloo): everything is already ordered."""

    def __init__(self):
        import torch
        self.torch = torch

    def side_stream(self):
        return None

    def event(self):
        return object()

    def record(self, ev, stream=None):
        pass

    def wait(self, ev, stream=None):
        pass

    def host_wait(self, ev):
        pass

    def on(self, stream):
        return contextlib.nullcontext()

    def synchronize(self):
        pass

    def pinned(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype)


def shared_host_buffer(nbytes, rank, name, pin=True):
    """One host buffer visible to every rank of the node (a file in /dev/shm, mapped by all), page-locked so that
    device-to-host copies into it run at the PCIe rate.  Rank 0 creates it; call behind a barrier on the others."""
    import torch
    path = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp", name)
    if rank == 0:
        with open(path, "wb") as f:
            f.truncate(nbytes)
    t = torch.from_file(path, shared=True, size=nbytes, dtype=torch.uint8)
    if pin and torch.cuda.is_available():
        rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), nbytes, 0)
        if int(rc) != 0:
            raise RuntimeError(f"cudaHostRegister failed: {rc}")
    return t, path


class StepPipeline:
    """The N > 1 step loop.  `encode(b)` must launch (asynchronously, on the current stream) the encode of this rank's
    batch into `outs[b]` and leave (total bytes, status) in `metas[b]` (int64[2], same device).

    step()   encode of the next step into buffer b; the exchange of the previous step runs behind it on the side stream
    drain()  exchanges whatever is still pending
    fence()  drain + wait for the device; afterwards `result()` is rank 0's gathered stream of the LAST step

    The status word travels with the byte count: a batch whose status is not zero has undefined output
    (include/mpeg1_hip.h), so nothing of it is shipped.  M1V_STATUS_SCRATCH on any rank: the ranks concerned call
    `retry(b)` — which must re-encode buffer b's batch with the worst-case scratch reserved
    (Mpeg1Encoder.reserve_scratch) and leave its new (total, status) in metas[b] — and the counts are gathered again;
    any other bit, or a second failure, raises on EVERY rank (all ranks see all status words).
    """

    STATUS_SCRATCH = 4

    def __init__(self, encode, outs, metas, world, rank, transport="xgmi", runtime=None, group=None,
                 host_buffer=None, gather_capacity=None, retry=None):
        import torch
        import torch.distributed as dist
        assert transport in ("xgmi", "host")
        self.torch, self.dist = torch, dist
        self.encode, self.outs, self.metas, self.retry = encode, outs, metas, retry
        self.world, self.rank, self.group, self.transport = world, rank, group, transport
        dev = outs[0].device
        self.rt = runtime if runtime is not None else (_CudaRuntime(dev) if dev.type == "cuda" else _HostRuntime())
        nb = len(outs)
        self.side = self.rt.side_stream()
        self.encoded = [self.rt.event() for _ in range(nb)]
        self.counted = [self.rt.event() for _ in range(nb)]
        self.drained = [None] * nb
        # everything the exchange touches is allocated here, once
        self.counts_dev = [torch.zeros(2 * world, dtype=torch.int64, device=dev) for _ in range(nb)]    # (total, status) per rank
        self.counts_host = [self.rt.pinned((2 * world,), torch.int64) for _ in range(nb)]
        self.host_buffer = host_buffer                      # transport "host": the node's shared pinned stream
        self.gathered = None                                # transport "xgmi": rank 0's device-side stream
        if transport == "xgmi" and rank == 0:
            cap = gather_capacity if gather_capacity is not None else world * outs[0].numel()
            self.gathered = torch.empty(cap, dtype=torch.uint8, device=dev)
        self.pending, self.step_no = [], 0
        self.last_counts, self.exchanges, self.retries = None, 0, 0

    # ---- one exchange, on the side stream -------------------------------------------------------------
    def _counts(self, b):
        """(total, status) of every rank for buffer b, on the host: the step's only host wait (the next encode is
        already queued on the main stream)."""
        rt = self.rt
        self.dist.all_gather_into_tensor(self.counts_dev[b], self.metas[b], group=self.group)
        self.counts_host[b].copy_(self.counts_dev[b], non_blocking=True)
        rt.record(self.counted[b], self.side)
        rt.host_wait(self.counted[b])
        pairs = [int(x) for x in self.counts_host[b]]
        return pairs[0::2], [s & 0xFFFFFFFF for s in pairs[1::2]]

    def _exchange(self, b):
        rt, dist = self.rt, self.dist
        with rt.on(self.side):
            rt.wait(self.encoded[b], self.side)
            counts, status = self._counts(b)
            if any(status):
                fatal = [(r, s) for r, s in enumerate(status) if s & ~self.STATUS_SCRATCH]
                if fatal or self.retry is None:
                    raise RuntimeError(f"encode status (rank, bits) {fatal or list(enumerate(status))}: the batch's output is undefined")
                if status[self.rank]:
                    self.retry(b)               # this rank's batch again, worst-case scratch reserved
                    self.retries += 1
                counts, status = self._counts(b)
                if any(status):
                    raise RuntimeError(f"encode status {status} after the retry")
            if self.transport == "xgmi":
                if self.rank == 0:
                    need = sum(counts)
                    if self.gathered.numel() < need:  # cannot happen with the default capacity (sum of the out buffers)
                        raise RuntimeError("gather buffer too small")
                    self.gathered[:counts[0]].copy_(self.outs[b][:counts[0]], non_blocking=True)
                works = _post_blobs(self.outs[b], counts, self.gathered, self.rank, self.group)
                for w in works:
                    w.wait()    # nccl: orders the side stream behind the transfer; gloo: blocks (CPU tests)
            else:
                off = sum(counts[:self.rank])
                if off + counts[self.rank] > self.host_buffer.numel():
                    raise RuntimeError("shared host buffer too small")
                self.host_buffer[off:off + counts[self.rank]].copy_(self.outs[b][:counts[self.rank]], non_blocking=True)
            ev = rt.event()
            rt.record(ev, self.side)
            self.drained[b] = ev
        self.last_counts = counts
        self.exchanges += 1

    def step(self):
        b = self.step_no % len(self.outs)
        self.step_no += 1
        if self.drained[b] is not None:
            self.rt.wait(self.drained[b])        # buffer b has left for rank 0 / the host: free again
        self.encode(b)
        self.rt.record(self.encoded[b])
        self.pending.append(b)
        if len(self.pending) > 1:                # the exchange lags one step behind the encode
            self._exchange(self.pending.pop(0))

    def drain(self):
        while self.pending:
            self._exchange(self.pending.pop(0))

    def fence(self):
        self.drain()
        self.rt.synchronize()
        self.dist.barrier(group=self.group)

    def result(self):
        """After fence(): the last step's gathered stream (rank 0; uint8 tensor) or None."""
        if self.rank != 0 or self.last_counts is None:
            return None
        need = sum(self.last_counts)
        return (self.gathered if self.transport == "xgmi" else self.host_buffer)[:need]
