"""Oscillators (reference src/signals/chain/osc.py:18-62): closed-form in absolute position --
`t = frame_range / rate * hertz + phase` in float64, then the waveform -- evaluated by the
`sig_osc_bank` HIP kernel.  No carried phase: any block renders at any position."""
import abc

import torch

from signals_amd import SignalFlags, _native
from signals_amd.chain import (
    BlockCachingEmitter,
    ImplicitChannels,
    Request,
    as_control,
    broadcast_shape,
    port,
    result_dtype,
)


class Osc(BlockCachingEmitter, ImplicitChannels, abc.ABC):
    hertz = port('hertz')
    phase = port('phase')

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.GENERATOR

    @classmethod
    @abc.abstractmethod
    def kind(cls) -> str:
        """kernel selector: 'Sine' | 'Square' | 'Sawtooth' | 'Triangle'"""
        raise NotImplementedError

    def _eval(self, request: Request) -> torch.Tensor:
        # phase: cycles; hertz: cycles/second -- both at block rate (osc.py:28-30)
        phase = as_control(self.phase.forward_at_block_rate(request))
        hertz = as_control(self.hertz.forward_at_block_rate(request))
        loc = request.loc
        # (N,1) * (1,Ch) + (1,Cp) under numpy broadcasting
        frames, voices = broadcast_shape((loc.shape.frames, 1), hertz.shape, phase.shape)
        out = torch.empty((frames, voices), dtype=result_dtype(frames), device=hertz.device)
        return _native.osc_bank(self.kind(), loc.position, loc.rate, hertz, phase, out)


class Sine(Osc):

    @classmethod
    def kind(cls) -> str:
        return 'Sine'


class Square(Osc):

    @classmethod
    def kind(cls) -> str:
        return 'Square'


class Sawtooth(Osc):

    @classmethod
    def kind(cls) -> str:
        return 'Sawtooth'


class Triangle(Osc):

    @classmethod
    def kind(cls) -> str:
        return 'Triangle'
