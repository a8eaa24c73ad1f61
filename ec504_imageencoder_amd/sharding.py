"""Multi-GPU sharding of the path: frames are independent given their global index, so rank r
encodes a contiguous block of frames and the per-rank bitstreams are gathered on rank 0 — the path's only
exchange.  One process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU node; "gloo"
in the CPU tests).  No reduction anywhere.
"""


def shard_range(n_frames, world, rank):
    """Contiguous, balanced split: returns (first_global_index, count) of this rank."""
    base, extra = divmod(n_frames, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def gather_bitstreams(blob, nbytes, dst=None, group=None):
    """blob: uint8 tensor holding this rank's frame records in its first `nbytes` bytes (device = the
    backend's device).  Returns on rank 0 a uint8 tensor with all ranks' records in rank (= frame) order and
    the list of per-rank byte counts; on other ranks (None, counts).

    Protocol: all_gather of the 8-byte totals, then every rank > 0 sends its blob to rank 0 (grouped
    send/recv: with RCCL the 7 peers of an 8-GPU node arrive over 7 different xGMI links)."""
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
        ops, off = [], counts[0]
        for r in range(1, world):
            if counts[r]:
                ops.append(dist.P2POp(dist.irecv, dst[off:off + counts[r]], r, group=group))
            off += counts[r]
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return dst[:need], counts
    if counts[rank]:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, blob[:counts[rank]], 0, group=group)]):
            w.wait()
    return None, counts
