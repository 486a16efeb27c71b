"""Voice sharding across the GPUs of one node.

Voices are the `channels` axis and nothing on the render path mixes channels before the bus, so a
graph of V voices splits into contiguous voice ranges, one process per GPU, with zero traffic while
rendering.  The only exchange step is the sum of the (frames, bus_channels) float32 bus: one RCCL
all-reduce per batch (backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests).  The
message is tiny -- K*N*2*4 B, 512 KiB for a 256-block batch -- so it is latency-bound, which is why it
is issued once per batch and not per block.  Sum order differs from a single-GPU render, so results
agree to rounding (1e-6 bar), not bitwise.
"""
from __future__ import annotations

import os
import typing

import torch
import torch.distributed as dist


def shard_voices(total: int, world: int, rank: int, group: int = 1) -> tuple[int, int]:
    """[lo, hi) voice range of `rank`: contiguous, in whole `group`s (64 for MixMatrix, whose groups must
    not straddle a shard), sizes differing by at most one group, every voice covered exactly once."""
    if total % group:
        raise ValueError(f'{total} voices are not a whole number of groups of {group}')
    if not 0 <= rank < world:
        raise ValueError(f'rank {rank} outside world of {world}')
    groups = total // group
    base, extra = divmod(groups, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo * group, hi * group


def init_process_group() -> tuple[int, int]:
    """(rank, world) from the torchrun environment; RCCL when this process has a GPU, gloo otherwise."""
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    force = os.environ.get('SIG_FORCE_DIST') == '1'          # exercise the RCCL path on a single rank (tests)
    if force and world == 1:
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('SIG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            local = int(os.environ.get('LOCAL_RANK', '0'))
            dist.init_process_group('nccl', device_id=torch.device('cuda', local % torch.cuda.device_count()))
        else:
            dist.init_process_group(backend)          # gloo: CPU tests, or a multi-rank rehearsal on one GPU
    return rank, world


def reduce_bus(bus: torch.Tensor, dst: typing.Optional[int] = None, async_op: bool = False):
    """Sum the per-shard bus in place: all-reduce (every rank gets the mix) or reduce to `dst`.
    `async_op=True` returns `(bus, work)`: the collective runs on RCCL's own stream and overlaps whatever
    the render stream does next; call `work.wait()` (None when nothing was launched) before reading `bus`."""
    work = None
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('SIG_FORCE_DIST') == '1'):
        if dst is None:
            work = dist.all_reduce(bus, op=dist.ReduceOp.SUM, async_op=async_op)
        else:
            work = dist.reduce(bus, dst=dst, op=dist.ReduceOp.SUM, async_op=async_op)
    return (bus, work) if async_op else bus


class ShardedRenderer:
    """This rank's slice of a voice-parallel graph plus the bus reduction.

    `build(lo, hi)` returns the bus node (an Emitter whose reply is (frames, bus_channels)) for voices
    [lo, hi); per-voice parameter rows are sliced once, at graph-build time."""

    def __init__(self, build: typing.Callable[[int, int], 'object'], total_voices: int, bus_channels: int,
                 rate: int = 48000, group: int = 1, timer=None, fuse: bool = True, **engine_options):
        """`engine_options`: further BatchRenderer keywords (`fuse_program`, `specialise` ...), the same on every rank"""
        from signals_amd.engine import BatchRenderer
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.lo, self.hi = shard_voices(total_voices, self.world, self.rank, group)
        self.renderer = BatchRenderer(build(self.lo, self.hi), bus_channels, rate, timer=timer, fuse=fuse, **engine_options)

    def render(self, position: int, block_frames: int, nblocks: int, dst: typing.Optional[int] = None) -> torch.Tensor:
        """the mixed bus on every rank (dst None: all-reduce) or on rank `dst` only (reduce: half the traffic)"""
        return reduce_bus(self.renderer.render(position, block_frames, nblocks), dst=dst)

    def render_async(self, position: int, block_frames: int, nblocks: int, dst: typing.Optional[int] = None):
        """(bus, work): the bus reduction of this batch overlaps the next batch's kernels; `work.wait()`
        (if not None) before the bus is read."""
        return reduce_bus(self.renderer.render(position, block_frames, nblocks), dst=dst, async_op=True)
