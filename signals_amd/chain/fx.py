"""Effects and critical-frequency filters (reference src/signals/chain/fx.py:23-163).

Element-wise nodes call `sig_elementwise`; LowPass/HighPass call `sig_biquad_coldstart`, which
reproduces the reference's non-streaming semantics: every block is filtered from zero state over
[<=100 frames before | block] (the `after` window cannot reach the kept samples: sosfilt is causal)
with a Butterworth biquad designed per channel in the kernel.
"""
import abc
import enum

import torch

from signals_amd import SignalFlags, _native, runtime
from signals_amd.chain import (
    BlockCachingEmitter,
    ImplicitChannels,
    Receiver,
    Request,
    Shape,
    as_control,
    broadcast_shape,
    port,
    result_dtype,
)


class Effect(BlockCachingEmitter, ImplicitChannels, abc.ABC):

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.EFFECT


def _apply(op: str, a: torch.Tensor, b: torch.Tensor, c: torch.Tensor = None) -> torch.Tensor:
    shapes = [a.shape, b.shape] + ([c.shape] if c is not None else [])
    rows, cols = broadcast_shape(*shapes)
    out = torch.empty((rows, cols), dtype=result_dtype(rows), device=a.device)
    return _native.elementwise(op, a, b, c, out)


class BinaryEffect(Effect, abc.ABC):
    left: Receiver.BoundPort = port('left')
    right: Receiver.BoundPort = port('right')


class Mix(BinaryEffect):
    """`mix * left + (1 - mix) * right`, mix at block rate (fx.py:35-40)"""
    mix: Receiver.BoundPort = port('mix')

    def _eval(self, request: Request) -> torch.Tensor:
        mix = as_control(self.mix.forward_at_block_rate(request))
        return _apply('Mix', self.left.forward(request), self.right.forward(request), mix)


class RingMod(BinaryEffect):
    """`left * right`, both at frame rate (fx.py:43-46)"""

    def _eval(self, request: Request) -> torch.Tensor:
        return _apply('RingMod', self.left.forward(request), self.right.forward(request))


class Gain(BinaryEffect):
    """`left * right`, right at block rate (fx.py:49-52)"""

    def _eval(self, request: Request) -> torch.Tensor:
        left = self.left.forward(request)
        return _apply('Gain', left, as_control(self.right.forward_at_block_rate(request)))


class Amp(BinaryEffect):
    """`copysign(left ** right, left)`, right at block rate; NaN for negative input with a fractional
    exponent, like numpy (fx.py:55-60)"""

    def _eval(self, request: Request) -> torch.Tensor:
        input_ = self.left.forward(request)
        return _apply('Amp', input_, as_control(self.right.forward_at_block_rate(request)))


class CritFilter(Effect, abc.ABC):
    input: Receiver.BoundPort = port('input')

    order = 2

    class Type(str, enum.Enum):
        low_pass = 'lp'
        high_pass = 'hp'
        band_pass = 'bp'
        band_stop = 'bs'

        def __str__(self) -> str:
            return str(self.value)

        @property
        def is_band(self) -> bool:
            return self.value.startswith('b')

    def __init__(self):
        super().__init__()
        self._status = None

    @abc.abstractmethod
    def type(self) -> 'CritFilter.Type':
        raise NotImplementedError

    def context_frames(self) -> int:
        return 100

    def _filter(self, request: Request, crit_1: torch.Tensor, crit_2: torch.Tensor = None) -> torch.Tensor:
        assert Shape.of_array(crit_1).frames == 1
        if crit_2 is not None:
            assert Shape.of_array(crit_2).frames == 1
        context_frames = self.context_frames()
        window = self.input.forward_with_context(request, context_frames)
        shape = request.loc.shape
        if window.shape[1] < shape.channels:
            raise IndexError(f'index {window.shape[1]} is out of bounds for axis 1 with size {window.shape[1]}')
        if crit_1.shape[1] < shape.channels:
            raise IndexError(f'index {crit_1.shape[1]} is out of bounds for axis 1 with size {crit_1.shape[1]}')
        history = window.shape[0] - shape.frames - context_frames     # rows the `before` request returned
        if history != min(context_frames, request.loc.position):
            raise ValueError(f'could not broadcast input array from shape ({window.shape[0] - context_frames},) '
                             f'into shape ({shape.frames},)')
        dtype = result_dtype(shape.frames)
        buf = window[:history + shape.frames, :shape.channels]
        if buf.dtype != dtype:
            buf = buf.to(dtype)
        cutoff = crit_1[:, :shape.channels]
        if not cutoff.is_contiguous():
            cutoff = cutoff.contiguous()
        if self._status is None:
            self._status = runtime.StatusWord(self.cls_name())
        result = torch.empty(tuple(shape), dtype=dtype, device=buf.device)
        if crit_2 is not None:
            # The reference raises TypeError here for every band filter (it star-unpacks a scalar, fx.py:99).
            # Its evident intent -- butter(N=2, Wn=[low, high], 'bp'|'bs') + sosfilt over the same window --
            # is what runs instead, pinned against scipy.
            if crit_2.shape[1] < shape.channels:
                raise IndexError(f'index {crit_2.shape[1]} is out of bounds for axis 1 with size {crit_2.shape[1]}')
            high = crit_2[:, :shape.channels]
            if not high.is_contiguous():
                high = high.contiguous()
            return _native.band_coldstart(str(self.type()), request.loc.rate, request.loc.position, shape.frames, 1,
                                          context_frames, cutoff, high, buf, history, result,
                                          status=self._status.tensor)
        return _native.biquad_coldstart(str(self.type()), request.loc.rate, request.loc.position,
                                        shape.frames, 1, context_frames, cutoff, buf, history, result,
                                        status=self._status.tensor)


class SingleCritFilter(CritFilter, abc.ABC):
    cutoff: Receiver.BoundPort = port('cutoff')

    def _eval(self, request: Request) -> torch.Tensor:
        hertz = as_control(self.cutoff.forward_at_block_rate(request))
        return self._filter(request, hertz)


class DoubleCritFilter(CritFilter, abc.ABC):
    low: Receiver.BoundPort = port('low')
    high: Receiver.BoundPort = port('high')

    def _eval(self, request: Request) -> torch.Tensor:
        low = as_control(self.low.forward_at_block_rate(request))
        high = as_control(self.high.forward_at_block_rate(request))
        return self._filter(request, low, high)


class LowPass(SingleCritFilter):

    def type(self) -> CritFilter.Type:
        return self.Type.low_pass


class HighPass(SingleCritFilter):

    def type(self) -> CritFilter.Type:
        return self.Type.high_pass


class BandPass(DoubleCritFilter):
    """4th-order Butterworth band-pass between the `low` and `high` ports (two biquad sections).  The
    reference class raises TypeError on every block (fx.py:99); this is its intent,
    `butter(2, [low, high], 'bp', output='sos')` + `sosfilt`, per SURVEY.md §8f-4."""

    def type(self) -> CritFilter.Type:
        return self.Type.band_pass


class BandStop(DoubleCritFilter):
    """4th-order Butterworth band-stop; see `BandPass`."""

    def type(self) -> CritFilter.Type:
        return self.Type.band_stop
