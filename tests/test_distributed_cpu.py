"""The N>1 path on CPU (gloo, world_size 2 and 3): frame sharding by global index + the gather of per-rank
bitstreams on rank 0 (ec504_imageencoder_amd/sharding.py, the code bench.py runs over RCCL).  The per-rank
producer here is the oracle (there is no GPU in this suite); the check is that the gathered stream equals the
single-process stream byte for byte — i.e. that sharding by `first_frame_index` is exact."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, W, H, result_path):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import oracle_ffi as orc
    from ec504_imageencoder_amd.sharding import gather_bitstreams, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard_range(n_frames, world, rank)
    frames = orc.synth_frames(count, W, H, seed=504, first_index=first)
    body, sizes = orc.encode_frames(frames, count, W, H, first, 12, orc.MODE_FULL) if count else (b"", [])
    blob = torch.zeros(len(body) + 64, dtype=torch.uint8)
    blob[:len(body)] = torch.frombuffer(bytearray(body), dtype=torch.uint8) if body else blob[:0]
    out, counts = gather_bitstreams(blob, len(body))
    if rank == 0:
        np.save(result_path, out.numpy())
        assert len(counts) == world and sum(counts) == out.numel()
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 7), (3, 5), (2, 1)])
def test_sharded_encode_equals_single_stream(orc, tmp_path, world, n_frames):
    W, H = 176, 144
    result = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_frames, W, H, result), nprocs=world, join=True)
    got = np.load(result).tobytes()
    frames = orc.synth_frames(n_frames, W, H, seed=504)
    want, _ = orc.encode_frames(frames, n_frames, W, H, 0, 12, orc.MODE_FULL)
    assert got == want


def test_shard_range_partitions_exactly():
    from ec504_imageencoder_amd.sharding import shard_range
    for n in (0, 1, 7, 300, 2400, 2401):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f1 == f0 + c0
