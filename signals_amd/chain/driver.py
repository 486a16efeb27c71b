"""Headless block driver: the pull side of the graph without PortAudio.

`BlockDriver` steps like the reference's `SinkDevice._callback` (reference
src/signals/chain/dev.py:161-179): each pull builds a `BlockLoc` at `frame_position`, requests it
from the `input` port, and advances by `frames`; an exception stops the stream.  The block leaves
the GPU only here (the sink edge), as a numpy array.  `render()` hands whole streams to the batched
engine when the graph qualifies and falls back to block-by-block pulls otherwise.
"""
import importlib
import typing

import attr
import attrs.validators
import numpy as np
import torch

from signals_amd import SignalFlags, runtime
from signals_amd.chain import (
    BlockLoc,
    Receiver,
    Shape,
    Signal,
    port,
    state,
)
from signals_amd.discovery import is_concrete_subclass


def load_signal(name: str) -> typing.Type[Signal]:
    """Resolve a qualified class name as written in `.sigs` patches; `signals.` names map onto this
    package (reference src/signals/chain/discovery.py:129-140)."""
    module, _, cls_name = name.rpartition('.')
    if module == 'signals' or module.startswith('signals.'):
        module = 'signals_amd' + module[len('signals'):]
    cls = getattr(importlib.import_module(module), cls_name)
    if not is_concrete_subclass(cls, Signal):
        raise TypeError(f'{name} is not a concrete Signal')
    return cls


class BlockDriver(Receiver):
    input: Receiver.BoundPort = port('input')

    @state
    class State(Receiver.State):
        channels: int = attr.ib(validator=attrs.validators.ge(1), default=1)

    def __init__(self, rate: int = 48000, blocksize: int = 256):
        super().__init__()
        self.rate = rate
        self.blocksize = blocksize
        self.frame_position = 0
        self.is_active = True
        self._engine = None
        self._engine_key = None
        self._engine_ok = True
        self.engine_options: dict = {}                        # further BatchRenderer keywords (`fuse_program`, `specialise` ...)

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.SINK_DEVICE

    def seek(self, position: int) -> None:
        self.frame_position = position * self.blocksize

    def tell(self) -> int:
        return self.frame_position // self.blocksize

    def _broadcast(self, block: torch.Tensor, frames: int) -> np.ndarray:
        out = np.empty((frames, self._state.channels), dtype=np.float32)
        out[:, :] = block.detach().cpu().numpy()             # numpy broadcast, like dev.py:178
        return out

    def pull(self, frames: typing.Optional[int] = None, eager: bool = False) -> np.ndarray:
        """One callback's worth: (frames, channels) float32, then advance.  By default the block goes through
        the batched engine as a batch of one (a graph that fuses to one launch is replayed per block without
        re-walking it); `eager=True`, or a graph the engine cannot schedule, uses the reference-shaped pull."""
        frames = self.blocksize if frames is None else frames
        try:
            block = None if eager else self._engine_render(frames, 1)
            if block is None:
                loc = BlockLoc(position=self.frame_position, rate=self.rate,
                               shape=Shape(frames=frames, channels=self._state.channels))
                block = self.input.request(loc)
            runtime.check_status()
        except Exception:
            self.is_active = False
            raise
        self.frame_position += frames
        return self._broadcast(block, frames)

    def _engine_render(self, frames: int, nblocks: int) -> typing.Optional[torch.Tensor]:
        """the input rendered by the batched engine, or None when the graph needs the eager path"""
        from signals_amd import chain, engine
        if not self.input or frames < 2:
            return None
        key = (chain.graph_clock.version, self.input.sig, self._state.channels, self.rate)
        if self._engine_key != key:
            # pull() copies every block to the host before asking for the next: graph-owned buffers are safe
            self._engine = engine.BatchRenderer(self.input.sig, self._state.channels, self.rate, graph_replay=True,
                                                **self.engine_options)
            self._engine_key = key
            self._engine_ok = True
        if not self._engine_ok:
            return None
        try:
            return self._engine.render(self.frame_position, frames, nblocks)
        except engine.NotBatchable:
            self._engine.reset()
            self._engine_ok = False         # until the graph changes
            return None

    def render(self, nblocks: int, frames: typing.Optional[int] = None, batched: bool = True) -> np.ndarray:
        """`nblocks` consecutive blocks as one (nblocks*frames, channels) array"""
        frames = self.blocksize if frames is None else frames
        if batched:
            block = self._engine_render(frames, nblocks)
            if block is not None:
                runtime.check_status()
                self.frame_position += frames * nblocks
                return self._broadcast(block, frames * nblocks)
        return np.concatenate([self.pull(frames, eager=not batched) for _ in range(nblocks)])
