"""Per-emitter block cache: <=16 `BlockLoc` keys, FIFO eviction, exact hit or the
first cached block that CONTAINS the request, sliced as a view (reference
src/signals/chain/__init__.py:424-457).  It is what makes a filter's `before` window cheap and what
makes cascaded filters history-dependent (A9), so it is reproduced exactly rather than optimised away.
"""
from __future__ import annotations

import abc

import attr
import torch

from signals_amd.chain.blocks import BlockLoc, Request, Shape
from signals_amd.chain.nodes import Emitter


class NotCached(RuntimeError):
    pass


class BlockCachingEmitter(Emitter, abc.ABC):
    """Per-emitter block cache: <=16 BlockLoc keys, FIFO; exact hit, else the first cached block that
    CONTAINS the request, sliced (a view) (chain/__init__.py:424-457).  This is what makes cascaded
    filters history-dependent; it is reproduced exactly, not optimised away."""

    def __init__(self):
        super().__init__()
        self._block_cache: dict[BlockLoc, torch.Tensor] = {}
        self._max_cached_blocks = 16

    def _read_block_cache(self, request: Request) -> torch.Tensor:
        try:
            return self._block_cache[request.loc]
        except KeyError:
            for loc, block in self._block_cache.items():
                if request.loc <= loc:
                    requested_shape = request.loc.shape
                    start = request.loc.position - loc.position
                    result = block[start:start + requested_shape.frames, :requested_shape.channels]
                    shape = Shape.of_array(result)
                    assert shape == requested_shape, (shape, requested_shape)
                    return result
            raise NotCached

    def _write_block_cache(self, block: torch.Tensor, request: Request) -> None:
        loc = attr.evolve(request.loc, shape=Shape.of_array(block))
        self._block_cache[loc] = block
        if len(self._block_cache) > self._max_cached_blocks:
            self._block_cache.pop(next(iter(self._block_cache)))

    def respond(self, request: Request) -> torch.Tensor:
        try:
            result = self._read_block_cache(request)
        except NotCached:
            result = super().respond(request)
            self._write_block_cache(result, request)
        return result
