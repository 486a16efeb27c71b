"""Block addressing and the chain layer's error types.

`Shape` is the `(frames, channels)` pair with the reference's broadcast-compatible ordering
(reference src/signals/chain/__init__.py:25-84); `BlockLoc` is the integer address of a block and
the block-cache key (:107-159); `Request` carries it through the pull protocol (:162-166).
Pure host-side integer logic -- nothing here touches the GPU.
"""
from __future__ import annotations

import enum
import functools
import typing

import attr
import numpy as np

from signals_amd import PortName, SignalsError


class ChainLayerError(SignalsError):
    pass


class Shape(typing.NamedTuple):
    """`(frames, channels)`.  `a <= b`: every dim of `a` is 1 or equals `b`'s -- a reply may be
    broadcast-compatible with the request (chain/__init__.py:25-84).

    >>> s = Shape(frames=10, channels=2)
    >>> s == (10, 2), s <= (10, 2), s >= (10, 2)
    (True, True, True)
    >>> (1, 1) <= Shape(frames=s.frames, channels=1) <= s
    True
    >>> (1, 1) <= Shape(frames=1, channels=s.channels) <= s
    True
    >>> (0, 0) <= s, Shape(frames=3, channels=2) <= s, Shape(frames=10, channels=0) <= s
    (False, False, False)
    """
    frames: int
    channels: int

    @classmethod
    def unit(cls) -> 'Shape':
        return cls(frames=1, channels=1)

    def __le__(self, other) -> bool:
        return (self[0] in (1, other[0])) and (self[1] in (1, other[1]))

    def __ge__(self, other) -> bool:
        return (other[0] in (1, self[0])) and (other[1] in (1, self[1]))

    @classmethod
    def of_array(cls, array) -> 'Shape':
        """Shape of a 2-D tensor/array; any other rank raises TypeError like the reference
        (chain/__init__.py:66-84).

        >>> Shape.of_array(np.array([[1, 2, 3]]))
        Shape(frames=1, channels=3)
        >>> Shape.of_array(np.zeros((3, 1)))
        Shape(frames=3, channels=1)
        """
        return cls(*(int(d) for d in array.shape))


class BadShape(ChainLayerError):

    def __init__(self, source: 'Signal', shape: tuple, constraint: tuple):
        super().__init__(f'Invalid response from {source.cls_name()!r}): '
                         f'Block with shape {tuple(shape)} incompatible with requested shape {tuple(constraint)}')


class BadStateSchema(ChainLayerError):

    def __init__(self, sig: 'Signal', state: 'Signal.State'):
        super().__init__(f'Signal {sig.cls_name()!r} cannot accept state of type {state.cls_name()!r}')


class BadStateValue(ChainLayerError):

    def __init__(self, state: 'Signal.State', key: str, value: typing.Any, reason: typing.Any = None):
        reason = '' if reason is None else f': ({reason})'
        super().__init__(f'Value {value!r} is invalid for property {key!r} in schema {state.cls_name()!r}{reason}')


@attr.s(auto_attribs=True, frozen=True, kw_only=True, order=False)
class BlockLoc:
    """Address of a block: absolute frame `position`, sample `rate`, `shape`.  Hashable: it is the
    block-cache key.  Integer arithmetic only."""
    position: int
    rate: int
    shape: Shape

    @property
    def end_position(self) -> int:
        return self.position + self.shape[0]

    @property
    def timestamp(self) -> float:
        return self.position / self.rate

    @functools.cached_property
    def frame_range(self) -> np.ndarray:
        """int64 column of absolute frame indices, read-only (chain/__init__.py:121-125).  The kernels
        regenerate the same integers from `position`; this host copy is for user nodes and tests."""
        frames = np.arange(self.position, self.end_position, dtype=np.int64).reshape(-1, 1)
        frames.flags.writeable = False
        return frames

    def resize(self, new_frames: int) -> 'BlockLoc':
        if new_frames == self.shape.frames:
            return self
        return attr.evolve(self, shape=Shape(frames=new_frames, channels=self.shape.channels))

    def reslice(self, new_channels: int) -> 'BlockLoc':
        if new_channels == self.shape.channels:
            return self
        return attr.evolve(self, shape=Shape(frames=self.shape.frames, channels=new_channels))

    def __le__(self, other: 'BlockLoc') -> bool:
        """containment: same rate, frame span inside `other`'s, no more channels"""
        return (
            self.rate == other.rate
            and self.position >= other.position
            and self.end_position <= other.end_position
            and self.shape.channels <= other.shape.channels
        )

    def before(self, frames: int) -> 'BlockLoc':
        return attr.evolve(self,
                           position=max(self.position - frames, 0),
                           shape=Shape(frames=min(frames, self.position), channels=self.shape.channels))

    def after(self, frames: int) -> 'BlockLoc':
        return attr.evolve(self,
                           position=self.end_position,
                           shape=Shape(frames=frames, channels=self.shape.channels))


@attr.s(auto_attribs=True, frozen=True, kw_only=True)
class Request:
    requestor: 'Receiver'
    port: PortName
    loc: BlockLoc


class RequestRate(enum.Enum):
    UNKNOWN = enum.auto()
    BLOCK = enum.auto()
    FRAME = enum.auto()
    UNUSED_FRAME = enum.auto()
