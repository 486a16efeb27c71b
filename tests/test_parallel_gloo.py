"""N > 1 path on CPU: voice sharding + the bus reduction over gloo, world_size 2.
Kernels need a GPU, so each rank's "bus" here is a host-computed stand-in with a known closed form;
what is under test is the partition and the collective."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from signals_amd.parallel import reduce_bus, shard_voices


def test_shards_partition_the_voices():
    for total, world, group in ((8192, 8, 1), (8192, 8, 64), (4096, 3, 64), (1024, 1, 1), (640, 4, 64), (10, 4, 1)):
        ranges = [shard_voices(total, world, r, group) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert all(lo % group == 0 and hi % group == 0 for lo, hi in ranges)
        sizes = [hi - lo for lo, hi in ranges]
        assert max(sizes) - min(sizes) <= group
    assert shard_voices(8192, 8, 3) == (3072, 4096)
    with pytest.raises(ValueError):
        shard_voices(100, 2, 0, 64)
    with pytest.raises(ValueError):
        shard_voices(128, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, frames, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from signals_amd.parallel import init_process_group
    r, w = init_process_group()
    assert (r, w) == (rank, world) and dist.get_backend() == 'gloo'
    lo, hi = shard_voices(total, world, rank, group=64)
    # stand-in stereo bus of this shard: voice v contributes (v+1)*(n+1) left, (v+1) right
    n = torch.arange(1, frames + 1, dtype=torch.float32)[:, None]
    v = torch.arange(lo + 1, hi + 1, dtype=torch.float32)[None, :]
    bus = torch.stack([(n * v).sum(1), v.expand(frames, -1).sum(1)], dim=1)
    mixed = reduce_bus(bus.clone())
    partial = reduce_bus(bus.clone(), dst=0)
    out[rank] = (mixed, partial)
    dist.barrier()
    dist.destroy_process_group()


def test_bus_all_reduce_world_size_2():
    world, total, frames = 2, 256, 16
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), total, frames, out), nprocs=world, join=True)
    s = total * (total + 1) / 2
    n = torch.arange(1, frames + 1, dtype=torch.float32)
    expect = torch.stack([n * s, torch.full((frames,), s)], dim=1)
    for rank in range(world):
        assert torch.equal(out[rank][0], expect)          # every rank holds the full mix
    assert torch.equal(out[0][1], expect)                 # reduce-to-0 variant
