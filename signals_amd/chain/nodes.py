"""Node base types and the pull protocol: `Signal` + attrs `State`,
`Emitter.respond/_get_result/_eval`, `Receiver` with `BoundPort` request/forward helpers, the
`port()` property factory and the channel-count mixins (reference
src/signals/chain/__init__.py:169-417), with `torch.Tensor` replies resident on the render device.
"""
from __future__ import annotations

import abc
import collections
import ctypes
import typing

import attr
import attrs.validators
import numpy as np
import torch

import signals_amd.discovery
from signals_amd import PortName, SignalFlags, runtime
from signals_amd.chain.blocks import (
    BadShape,
    BadStateSchema,
    BlockLoc,
    Request,
    RequestRate,
    Shape,
)

CTRL_DTYPE = torch.float64      # one-row replies
AUDIO_DTYPE = torch.float32     # multi-row replies


_SMALL_BYTES = 1 << 15      # snapshots up to this size are kept as bytes (one memcpy + memcmp, ~0.4 us for 1024 doubles)
_libc = ctypes.CDLL(None)
_libc.memcmp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
_libc.memcmp.restype = ctypes.c_int


class HostSnapshot:
    """What a host array (a `Fixed` value, bus gains, a mix matrix) held when it was uploaded, so that an in-place edit
    of ANY size is seen at the next reply
    (the reference's `Fixed._eval` returns the live array, fixed.py:38-39, so it always sees them).  Bitwise
    comparison: identical NaNs compare equal (no re-upload per reply), -0.0 differs from 0.0 (a harmless re-upload)."""
    __slots__ = ('layout', 'data', 'ptr')

    def __init__(self, value: np.ndarray):
        self.layout = (value.shape, value.dtype, value.strides)
        if value.nbytes <= _SMALL_BYTES:
            self.data, self.ptr = value.tobytes(), 0
        else:
            self.data = np.ascontiguousarray(value).copy()
            self.ptr = self.data.ctypes.data

    def matches(self, value: np.ndarray) -> bool:
        if (value.shape, value.dtype, value.strides) != self.layout:
            return False
        if self.ptr == 0:
            return value.tobytes() == self.data
        if value.flags.c_contiguous:
            return _libc.memcmp(value.ctypes.data, self.ptr, value.nbytes) == 0
        return np.ascontiguousarray(value).tobytes() == self.data.tobytes()


class _Port(property):
    pass


state = attr.s(auto_attribs=True, frozen=False, kw_only=True)


class GraphClock:
    """Monotonic counter bumped by every port or state mutation; compiled batch plans
    (signals_amd.chain.driver) are keyed on it and re-planned at the next block boundary."""

    def __init__(self):
        self.version = 0

    def tick(self) -> None:
        self.version += 1


graph_clock = GraphClock()
_graph_changed = graph_clock.tick


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
        return adopt_reply(self._get_result(request))

    def destroy(self) -> None:
        super().destroy()
        for port_name, receiver in tuple(self.outputs_with_ports):
            delattr(receiver, port_name)


class BoundPort:
    """One input port of one receiver instance: holds the connected emitter and issues requests to it.
    (`Receiver.BoundPort` in the reference, chain/__init__.py:267-322.)"""

    def __init__(self, parent: 'Receiver', name: PortName, emitter: 'Emitter' = None):
        self.name = name
        self.parent = parent
        self.sig = emitter

    def __bool__(self):
        return self.sig is not None

    @property
    def channels(self) -> typing.Optional[int]:
        return None if self.sig is None else self.sig.channels

    # -- wiring
    def assign(self, input_: 'Emitter') -> None:
        if self.sig is not None:
            self.expel()
        self.sig = input_
        input_._outputs.add((self.name, self.parent))
        _graph_changed()

    def expel(self) -> None:
        self.sig._outputs.remove((self.name, self.parent))
        self.sig = None
        _graph_changed()

    # -- pulling
    def request(self, loc: BlockLoc) -> torch.Tensor:
        """The reply for `loc`, shape-checked; an unplugged port answers zeros((1,1))."""
        if self.sig is None:
            return np.zeros(Shape.unit()) if wants_host_arrays(self.parent) else Emitter.empty_result()
        return self._do_request(self._make_request(loc))

    def _make_request(self, loc: BlockLoc) -> Request:
        return Request(requestor=self.parent, port=self.name, loc=loc)

    def _do_request(self, request: Request):
        block = adopt_reply(self.sig.respond(request))      # (a plugin may override respond() itself)
        # compare as Shape: torch.Size <= Shape would be a lexicographic tuple compare
        if not (Shape.of_array(block) <= request.loc.shape):
            raise BadShape(self.sig, block.shape, request.loc.shape)
        if wants_host_arrays(self.parent):
            return block.to(CTRL_DTYPE).cpu().numpy()       # what the reference hands its nodes: float64 (frames, channels)
        return block

    def forward(self, request: Request) -> torch.Tensor:
        return self.request(request.loc)

    def forward_at_block_rate(self, request: Request) -> torch.Tensor:
        """same position, one frame: how control inputs (hertz, cutoff, gain ...) are read"""
        return self.request(request.loc.resize(1))

    def forward_with_context(self, request: Request, context_frames: int) -> torch.Tensor:
        """[<=context before | block | context after] concatenated along frames
        (chain/__init__.py:308-315).  The `after` request is issued like the reference does:
        it is what fills upstream caches for the next block."""
        loc = request.loc
        window = [self.request(loc.before(context_frames))] if loc.position > 0 else []
        window.append(self.forward(request))
        window.append(self.request(loc.after(context_frames)))
        return concatenate(window)


class Receiver(Signal, abc.ABC):
    BoundPort = BoundPort        # the reference nests the class; both spellings resolve
    # Plugin interop (reference chain/__init__.py:245-247: every `_eval` takes and returns numpy arrays).  True: this
    # node's ports hand it float64 numpy arrays (one device-to-host copy per request) instead of device tensors, so a
    # node written for the reference (`np.tanh(self.input.forward(request))`) runs unchanged inside a GPU graph.
    # None: decided by `host_plugins(...)` -- nodes defined outside this package get numpy once a script has called
    # `signals_amd.install_as_signals()`, i.e. declared itself written against the reference's API.
    HOST_ARRAYS: typing.Optional[bool] = None

    def __init__(self):
        super().__init__()
        self._ports = {name: BoundPort(parent=self, name=name) for name in self.port_names()}

    @classmethod
    def port_names(cls) -> list[PortName]:
        return [k for k in dir(cls) if isinstance(getattr(cls, k), _Port)]

    @property
    def inputs_by_port(self) -> dict[PortName, 'Emitter']:
        return {bound.name: bound.sig for bound in self._ports.values() if bound}

    def upstream(self) -> typing.Sequence['Emitter']:
        """Receivers feeding this node, dependencies first, self last (chain/__init__.py:347-358).
        A cycle raises AssertionError('Cycle detected') -- the reference asserts the same thing but
        recurses without bound on a direct loop; here the walk keeps its path and always terminates."""
        return self._upstream(set(), set())

    def _upstream(self, visited: set, path: set) -> collections.deque:
        assert self not in path, 'Cycle detected'
        path.add(self)
        result = collections.deque()
        for input_ in self.inputs_by_port.values():
            if input_ not in visited and isinstance(input_, Receiver):
                result.extend(input_._upstream(visited, path))
                visited.update(result)
        path.discard(self)
        assert self not in visited, 'Cycle detected'
        result.append(self)
        return result

    def destroy(self) -> None:
        super().destroy()
        for name, bound in tuple(self._ports.items()):
            if bound:
                delattr(self, name)


def port(name: PortName) -> _Port:
    """Class-level input port: `node.name = emitter` connects, `del node.name` disconnects,
    `node.name` is the BoundPort (chain/__init__.py:367-377)."""

    def fget(self: Receiver) -> BoundPort:
        return self._ports[name]

    def fdel(self: Receiver) -> None:
        self._ports[name].expel()

    def fset(self: Receiver, input_: Emitter) -> None:
        self._ports[name].assign(input_)

    return _Port(fget=fget, fset=fset, fdel=fdel)


_host_plugins = False


def host_plugins(enabled: bool) -> None:
    """nodes defined outside `signals_amd` (plugins written against the reference) receive numpy arrays at their ports"""
    global _host_plugins
    _host_plugins = bool(enabled)


def wants_host_arrays(receiver) -> bool:
    flag = getattr(receiver, 'HOST_ARRAYS', None)
    if flag is not None:
        return bool(flag)
    return _host_plugins and not type(receiver).__module__.startswith('signals_amd')


def adopt_reply(block) -> torch.Tensor:
    """A reply as a tensor on the render device.  Built-in nodes answer device tensors (returned as they are); a node
    written against the reference's API answers a numpy array (chain/__init__.py:245-247), which is uploaded here
    with this package's dtype rule -- float64 for a one-row (block-rate) reply, float32 audio otherwise.  Anything
    that is not 2-D raises TypeError at the shape check, like the reference's Shape.of_array."""
    if isinstance(block, torch.Tensor):
        if block.device != runtime.device():
            block = block.to(runtime.device())
        return block
    host = np.asarray(block)
    if host.ndim != 2:
        raise TypeError(f'a reply must be a 2-D (frames, channels) array, got shape {host.shape}')
    host = np.ascontiguousarray(host, dtype=np.float64)
    return torch.from_numpy(host).to(device=runtime.device(), dtype=result_dtype(host.shape[0]))


def concatenate(blocks: typing.Sequence[torch.Tensor]) -> torch.Tensor:
    """np.concatenate along frames: channel counts must match exactly (no broadcasting); a window
    that mixes control (f64) and audio (f32) rows is promoted to f64."""
    widths = {int(b.shape[1]) for b in blocks}
    if len(widths) > 1:
        raise ValueError('all the input array dimensions except for the concatenation axis must match exactly, '
                         f'got channel counts {sorted(widths)}')
    if any(isinstance(b, np.ndarray) for b in blocks):      # a HOST_ARRAYS receiver (a reference-style plugin): its ports hand it numpy
        return np.concatenate([np.asarray(b) for b in blocks])
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
