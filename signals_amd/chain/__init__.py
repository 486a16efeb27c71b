"""Node API and pull protocol -- the drop-in boundary.

Same names, ports, state attributes, error types and `(frames, channels)` semantics as the
reference's `signals.chain` (reference src/signals/chain/__init__.py:21-457), written fresh with
`torch.Tensor` buffers resident on the MI355X in place of numpy arrays:

  * a reply with one row (block rate, `forward_at_block_rate`) is a float64 control row;
  * a reply with more rows is float32 audio (arithmetic inside the kernels is float64 where the
    1e-6 bar needs it -- phase, filter coefficients and state);
  * replies are borrowed, read-only views: cache hits return the cached tensor or a slice of it
    (chain/__init__.py:431-442), `Fixed` returns its resident upload.

Nothing here computes samples; `_eval` bodies in osc/fx/shape/ext call the HIP kernels.
"""
from __future__ import annotations

import abc
import collections
import enum
import functools
import typing

import attr
import attrs.validators
import numpy as np
import torch

import signals_amd.discovery
from signals_amd import PortName, SignalFlags, SignalsError, runtime

CTRL_DTYPE = torch.float64      # one-row replies
AUDIO_DTYPE = torch.float32     # multi-row replies


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
        >>> Shape.of_array(torch.zeros(3, 1))
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


class _Port(property):
    pass


class RequestRate(enum.Enum):
    UNKNOWN = enum.auto()
    BLOCK = enum.auto()
    FRAME = enum.auto()
    UNUSED_FRAME = enum.auto()


state = attr.s(auto_attribs=True, frozen=False, kw_only=True)


def _graph_changed() -> None:
    """Port or state mutation invalidates compiled batch plans (signals_amd.chain.driver)."""
    global graph_version
    graph_version += 1


graph_version = 0


class Signal(abc.ABC, signals_amd.discovery.Named):
    @state
    class State(signals_amd.discovery.Named):
        pass

    def __init__(self):
        self._state = self.State()

    @classmethod
    @abc.abstractmethod
    def flags(cls) -> SignalFlags:
        return SignalFlags(0)

    @classmethod
    def state_attrs(cls) -> typing.AbstractSet[str]:
        return attr.fields_dict(cls.State).keys()

    def get_state(self) -> 'Signal.State':
        return self._state

    def set_state(self, new_state: 'Signal.State') -> None:
        if not isinstance(new_state, self.State):
            raise BadStateSchema(self, new_state)
        self._state = new_state
        _graph_changed()

    def destroy(self) -> None:
        pass


class Emitter(Signal, abc.ABC):
    @state
    class State(Signal.State):
        enabled: bool = attr.ib(validator=attrs.validators.instance_of(bool), default=True)

    def __init__(self):
        super().__init__()
        self._outputs: set[tuple[PortName, 'Receiver']] = set()
        self._last_request: typing.Optional[Request] = None

    @property
    def outputs_with_ports(self) -> typing.AbstractSet[tuple[PortName, 'Receiver']]:
        return self._outputs

    @property
    def rate(self) -> RequestRate:
        if self._last_request is None:
            return RequestRate.UNKNOWN
        frames = self._last_request.loc.shape.frames
        if frames <= 0:
            return RequestRate.UNKNOWN
        return RequestRate.BLOCK if frames == 1 else RequestRate.FRAME

    @property
    @abc.abstractmethod
    def channels(self) -> int:
        raise NotImplementedError

    @abc.abstractmethod
    def _eval(self, request: Request) -> torch.Tensor:
        raise NotImplementedError

    @classmethod
    def empty_result(cls) -> torch.Tensor:
        """zeros((1,1)) float64: what an unplugged port or a disabled emitter answers
        (chain/__init__.py:250-254, :297-298)."""
        return torch.zeros(Shape.unit(), dtype=CTRL_DTYPE, device=runtime.device())

    def _get_result(self, request: Request) -> torch.Tensor:
        return self._eval(request) if self._state.enabled else self.empty_result()

    def respond(self, request: Request) -> torch.Tensor:
        self._last_request = request
        return self._get_result(request)

    def destroy(self) -> None:
        super().destroy()
        for port_name, receiver in tuple(self.outputs_with_ports):
            delattr(receiver, port_name)


class Receiver(Signal, abc.ABC):
    class BoundPort:

        def __init__(self, parent: 'Receiver', name: PortName, emitter: 'Emitter' = None):
            self.name = name
            self.parent = parent
            self.sig = emitter

        def expel(self) -> None:
            self.sig._outputs.remove((self.name, self.parent))
            self.sig = None
            _graph_changed()

        def assign(self, input_: 'Emitter') -> None:
            if self.sig is not None:
                self.expel()
            self.sig = input_
            self.sig._outputs.add((self.name, self.parent))
            _graph_changed()

        def __bool__(self):
            return self.sig is not None

        def _make_request(self, loc: BlockLoc) -> Request:
            return Request(requestor=self.parent, port=self.name, loc=loc)

        def _do_request(self, request: Request) -> torch.Tensor:
            block = self.sig.respond(request)
            # NB: compare as Shape -- torch.Size <= Shape would be a lexicographic tuple compare
            if not (Shape.of_array(block) <= request.loc.shape):
                raise BadShape(self.sig, block.shape, request.loc.shape)
            return block

        def request(self, loc: BlockLoc) -> torch.Tensor:
            if self.sig is None:
                return Emitter.empty_result()
            return self._do_request(self._make_request(loc))

        def forward(self, request: Request) -> torch.Tensor:
            return self.request(request.loc)

        def forward_at_block_rate(self, request: Request) -> torch.Tensor:
            return self.request(request.loc.resize(1))

        def forward_with_context(self, request: Request, context_frames: int) -> torch.Tensor:
            """[<=context before | block | context after] concatenated along frames
            (chain/__init__.py:308-315).  The `after` request is issued like the reference does:
            it is what fills upstream caches for the next block."""
            blocks = []
            loc = request.loc
            if loc.position > 0:
                blocks.append(self.request(loc.before(context_frames)))
            blocks.append(self.forward(request))
            blocks.append(self.request(loc.after(context_frames)))
            return concatenate(blocks)

        @property
        def channels(self) -> typing.Optional[int]:
            return None if self.sig is None else self.sig.channels

    def __init__(self):
        super().__init__()
        self._ports = {
            port: self.BoundPort(parent=self, name=port)
            for port in self.port_names()
        }

    @classmethod
    def port_names(cls) -> list[PortName]:
        return [k for k in dir(cls) if isinstance(getattr(cls, k), _Port)]

    @property
    def inputs_by_port(self) -> dict[PortName, 'Emitter']:
        return {port.name: port.sig for port in self._ports.values() if port}

    def upstream(self) -> typing.Sequence['Emitter']:
        """Receivers feeding this node, dependencies first, self last (chain/__init__.py:347-358)."""
        return self._upstream(set())

    def _upstream(self, visited: set) -> collections.deque:
        result = collections.deque()
        for input_ in self.inputs_by_port.values():
            if input_ not in visited and isinstance(input_, Receiver):
                result.extend(input_._upstream(visited))
                visited.update(result)
        assert self not in visited, 'Cycle detected'
        result.append(self)
        return result

    def destroy(self) -> None:
        super().destroy()
        for port_name, bound_port in tuple(self._ports.items()):
            if bound_port:
                delattr(self, port_name)


def port(name: PortName) -> _Port:
    """Class-level input port: `node.name = emitter` connects, `del node.name` disconnects,
    `node.name` is the BoundPort (chain/__init__.py:367-377)."""

    def fget(self: Receiver) -> Receiver.BoundPort:
        return self._ports[name]

    def fdel(self: Receiver) -> None:
        self._ports[name].expel()

    def fset(self: Receiver, input_: Emitter) -> None:
        self._ports[name].assign(input_)

    return _Port(fget=fget, fset=fset, fdel=fdel)


def concatenate(blocks: typing.Sequence[torch.Tensor]) -> torch.Tensor:
    """np.concatenate along frames: channel counts must match exactly (no broadcasting); a window
    that mixes control (f64) and audio (f32) rows is promoted to f64."""
    widths = {int(b.shape[1]) for b in blocks}
    if len(widths) > 1:
        raise ValueError('all the input array dimensions except for the concatenation axis must match exactly, '
                         f'got channel counts {sorted(widths)}')
    dtypes = {b.dtype for b in blocks}
    if len(dtypes) > 1:
        blocks = [b.to(CTRL_DTYPE) for b in blocks]
    return torch.cat(list(blocks), dim=0)


def result_dtype(frames: int) -> torch.dtype:
    return CTRL_DTYPE if frames == 1 else AUDIO_DTYPE


def as_control(row: torch.Tensor) -> torch.Tensor:
    """A block-rate reply as the contiguous float64 (1, C) row the kernels take."""
    if row.dtype != CTRL_DTYPE:
        row = row.to(CTRL_DTYPE)
    return row if row.is_contiguous() else row.contiguous()


def broadcast_shape(*shapes) -> tuple[int, int]:
    """numpy broadcasting of 2-D shapes; ValueError on mismatch like numpy."""
    out = [1, 1]
    for s in shapes:
        for d in (0, 1):
            if s[d] != 1:
                if out[d] not in (1, s[d]):
                    raise ValueError('operands could not be broadcast together with shapes '
                                     + ' '.join(str(tuple(x)) for x in shapes))
                out[d] = int(s[d])
    return out[0], out[1]


class ExplicitChannels(Signal, abc.ABC):
    @state
    class State(Signal.State):
        channels: int = attr.ib(validator=attrs.validators.ge(1), default=1)


class ExplicitChannelsEmitter(ExplicitChannels, Emitter, abc.ABC):
    @state
    class State(ExplicitChannels.State, Emitter.State):
        pass

    @property
    def channels(self) -> int:
        return self._state.channels


class ImplicitChannels(Receiver, Emitter, abc.ABC):

    @property
    def channels(self) -> int:
        """the one non-1 input width (chain/__init__.py:396-406)"""
        widths = {input_.channels for input_ in self.inputs_by_port.values()}
        if len(widths) > 1:
            widths.discard(1)
        if len(widths) != 1:
            raise ValueError(f'expected exactly one input width, got {sorted(widths)}')
        return next(iter(widths))


class PassThroughResult(ImplicitChannels, abc.ABC):
    input: Receiver.BoundPort = port('input')

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.PASSTHRU

    def _get_result(self, request: Request) -> torch.Tensor:
        return super()._get_result(request) if self._state.enabled else self.input.forward(request)


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
