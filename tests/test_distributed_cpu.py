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


def _pipeline_worker(rank, world, port, n_per_rank, W, H, steps, transport, result_path, fail_step=-1, fail_bits=0):
    """bench.py's own N > 1 step loop (StepPipeline: step / pending / drain / fence, two output buffers, the exchange
    one step behind the encode) with the oracle as the per-rank producer.  Every step encodes DIFFERENT frames, so a
    buffer handed on too early or too late shows up as wrong bytes."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import oracle_ffi as orc
    from ec504_imageencoder_amd.sharding import StepPipeline, shared_host_buffer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cap = 64 * 1024 * n_per_rank
    outs = [torch.zeros(cap, dtype=torch.uint8) for _ in range(2)]
    metas = [torch.zeros(2, dtype=torch.int64) for _ in range(2)]
    state = {"step": 0, "held": {}, "retried": []}

    def produce(b, step, status):
        first = (step * world + rank) * n_per_rank               # global frame index of this rank's batch in this step
        frames = orc.synth_frames(n_per_rank, W, H, seed=504, first_index=first)
        body, _ = orc.encode_frames(frames, n_per_rank, W, H, first, 12, orc.MODE_FULL)
        if status:                                                # a failed batch: undefined output, wrong size
            body = bytes(len(body) // 2)
        outs[b][:len(body)] = torch.frombuffer(bytearray(body), dtype=torch.uint8)
        outs[b][len(body):len(body) + 16] = 0xEE                  # garbage behind the payload must not travel
        metas[b][0], metas[b][1] = len(body), status

    def encode(b):
        step = state["step"]
        state["held"][b] = step
        produce(b, step, fail_bits if (step == fail_step and rank == world - 1) else 0)
        state["step"] += 1

    def retry(b):                                                 # the same frames again, "with more scratch"
        state["retried"].append(state["held"][b])
        produce(b, state["held"][b], 0)

    host, path = (None, None)
    if transport == "host":
        if rank == 0:
            host, path = shared_host_buffer(world * cap, 0, f"ec504_test_{port}", pin=False)
        dist.barrier()
        if rank != 0:
            host, path = shared_host_buffer(world * cap, rank, f"ec504_test_{port}", pin=False)
    pipe = StepPipeline(encode, outs, metas, world, rank, transport=transport, host_buffer=host, retry=retry)
    if fail_bits & ~4:                                            # not recoverable: EVERY rank must raise, none may hang
        with pytest.raises(RuntimeError):
            for _ in range(steps):
                pipe.step()
            pipe.fence()
        dist.barrier()
        dist.destroy_process_group()
        return
    for _ in range(steps):
        pipe.step()
    pipe.fence()
    assert pipe.exchanges == steps and not pipe.pending
    assert state["retried"] == ([fail_step] if (fail_bits and rank == world - 1) else [])
    if rank == 0:
        np.save(result_path, pipe.result().numpy().copy())
    dist.barrier()
    dist.destroy_process_group()
    if path and rank == 0:
        os.unlink(path)


def test_step_pipeline_retries_a_batch_that_ran_out_of_scratch(orc, tmp_path):
    """A rank reports M1V_STATUS_SCRATCH once (step 3 of 5, its output half as long and zero): the pipeline must not ship
    that buffer; the rank re-encodes it and the last steps still arrive intact.  (Step 4 = the last step is the one
    compared, step 3's retry sits in the middle of the double-buffered loop.)"""
    W, H, n_per_rank, steps, world = 176, 144, 2, 5, 2
    result = str(tmp_path / "last.npy")
    mp.spawn(_pipeline_worker, args=(world, _free_port(), n_per_rank, W, H, steps, "xgmi", result, 3, 4), nprocs=world, join=True)
    first = (steps - 1) * world * n_per_rank
    frames = orc.synth_frames(world * n_per_rank, W, H, seed=504, first_index=first)
    want, _ = orc.encode_frames(frames, world * n_per_rank, W, H, first, 12, orc.MODE_FULL)
    assert np.load(result).tobytes() == want
    # the failing step is the LAST one: what rank 0 holds at the end is the retried batch itself
    mp.spawn(_pipeline_worker, args=(world, _free_port(), n_per_rank, W, H, steps, "host", result, 4, 4), nprocs=world, join=True)
    assert np.load(result).tobytes() == want


def test_step_pipeline_raises_on_every_rank_for_an_unrecoverable_status(orc, tmp_path):
    """M1V_STATUS_NOSPACE / UNENCODABLE leave the batch undefined and cannot be retried: all ranks raise."""
    mp.spawn(_pipeline_worker, args=(2, _free_port(), 2, 176, 144, 4, "xgmi", str(tmp_path / "x.npy"), 2, 2), nprocs=2, join=True)


@pytest.mark.parametrize("world,transport", [(2, "xgmi"), (2, "host"), (3, "xgmi")])
def test_step_pipeline_of_the_bench(orc, tmp_path, world, transport):
    W, H, n_per_rank, steps = 176, 144, 2, 5
    result = str(tmp_path / "last.npy")
    mp.spawn(_pipeline_worker, args=(world, _free_port(), n_per_rank, W, H, steps, transport, result), nprocs=world, join=True)
    got = np.load(result).tobytes()
    first = (steps - 1) * world * n_per_rank                     # the last step's frames, all ranks, in rank order
    frames = orc.synth_frames(world * n_per_rank, W, H, seed=504, first_index=first)
    want, _ = orc.encode_frames(frames, world * n_per_rank, W, H, first, 12, orc.MODE_FULL)
    assert got == want
