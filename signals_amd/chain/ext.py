"""Build-defined nodes under the same plugin API.  The reference names a
voice sum (`shape.Flatten`) but it sums over frames and crashes; these nodes are what BASELINE's
configurations need and are pinned against `oracle/chain_ref.py` ("parity unpinned" by the
reference itself)."""
import typing

import attr
import numpy as np
import torch

from signals_amd import SignalFlags, _native, runtime
from signals_amd.chain import (
    BadStateValue,
    BlockCachingEmitter,
    HostSnapshot,
    ImplicitChannels,
    PassThroughResult,
    Receiver,
    Request,
    as_control,
    broadcast_shape,
    port,
    result_dtype,
    state,
)


def _validate_gains(instance, attribute, new_value):
    if new_value is not None and not (isinstance(new_value, np.ndarray) and new_value.ndim == 2):
        raise BadStateValue(instance, attribute.name, new_value, 'must be None or a 2D array (bus_channels, voices)')


class SumBus(BlockCachingEmitter, Receiver):
    """Voice sum bus: `out[n, c] = sum_v gains[c, v] * input[n, v]`; with `gains=None` a mono sum
    `out[n, 0] = sum_v input[n, v]`.  float64 accumulation in a fixed order (sig_sum_bus)."""
    input: Receiver.BoundPort = port('input')

    @state
    class State(BlockCachingEmitter.State):
        gains: typing.Optional[np.ndarray] = attr.ib(default=None, validator=_validate_gains,
                                                     on_setattr=attr.setters.validate)

    def __init__(self):
        super().__init__()
        self._resident = None

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.EFFECT

    @property
    def channels(self) -> int:
        gains = self._state.gains
        return 1 if gains is None else int(gains.shape[0])

    def resident_gains(self) -> typing.Optional[torch.Tensor]:
        gains = self._state.gains
        if gains is None:
            return None
        held = self._resident
        if held is None or held[0] is not gains or not held[1].matches(gains):
            host = np.ascontiguousarray(gains, dtype=np.float64)
            self._resident = held = (gains, HostSnapshot(gains), torch.from_numpy(host.copy()).to(runtime.device()))
        return held[2]

    def _eval(self, request: Request) -> torch.Tensor:
        voices = self.input.channels
        x = self.input.request(request.loc.reslice(voices))
        out = torch.empty((x.shape[0], self.channels), dtype=result_dtype(x.shape[0]), device=x.device)
        return _native.sum_bus(x, self.resident_gains(), out)


class ADSR(BlockCachingEmitter, ImplicitChannels):
    """Position-pure piecewise-linear envelope per voice at frame rate; multiply it into a signal with
    `RingMod` (frame-rate product; `Gain.right` is block-rate, fx.py:52).  Ports (all block-rate, seconds
    except `sustain`, a level): attack, decay, sustain, release, gate_on, gate_off.  Definition:
    oracle/chain_ref.py:adsr; kernel: sig_adsr."""
    attack: Receiver.BoundPort = port('attack')
    decay: Receiver.BoundPort = port('decay')
    sustain: Receiver.BoundPort = port('sustain')
    release: Receiver.BoundPort = port('release')
    gate_on: Receiver.BoundPort = port('gate_on')
    gate_off: Receiver.BoundPort = port('gate_off')

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.GENERATOR | SignalFlags.EPOCH

    def control_rows(self, fetch) -> dict:
        return {name: as_control(fetch(getattr(self, name))) for name in _native.ADSR_PARAMS}

    def _eval(self, request: Request) -> torch.Tensor:
        rows = self.control_rows(lambda bound: bound.forward_at_block_rate(request))
        frames, voices = broadcast_shape((request.loc.shape.frames, 1), *(r.shape for r in rows.values()))
        out = torch.empty((frames, voices), dtype=result_dtype(frames), device=runtime.device())
        return _native.adsr(request.loc.position, request.loc.rate, rows, out)


def _validate_matrix(instance, attribute, new_value):
    if not (isinstance(new_value, np.ndarray) and new_value.shape == (64, 64)):
        raise BadStateValue(instance, attribute.name, new_value, 'must be a (64, 64) array')


class MixMatrix(BlockCachingEmitter, Receiver):
    """Dense 64x64 mix of every group of 64 consecutive voices:
    `out[n, 64g:64g+64] = input[n, 64g:64g+64] @ matrix` (float32, exact-f32 MFMA; sig_mix_matrix).
    Groups must not straddle a GPU shard."""
    input: Receiver.BoundPort = port('input')

    @state
    class State(BlockCachingEmitter.State):
        matrix: np.ndarray = attr.ib(factory=lambda: np.eye(64), validator=_validate_matrix,
                                     on_setattr=attr.setters.validate)

    def __init__(self):
        super().__init__()
        self._resident = None

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.EFFECT

    @property
    def channels(self) -> int:
        return self.input.channels

    def resident_matrix(self) -> torch.Tensor:
        matrix = self._state.matrix
        held = self._resident
        if held is None or held[0] is not matrix or not held[1].matches(matrix):
            host = np.ascontiguousarray(matrix, dtype=np.float32)
            self._resident = held = (matrix, HostSnapshot(matrix), torch.from_numpy(host.copy()).to(runtime.device()))
        return held[2]

    def _eval(self, request: Request) -> torch.Tensor:
        x = self.input.forward(request)
        if x.shape[1] % 64:
            raise ValueError(f'MixMatrix needs a multiple of 64 voices, got {x.shape[1]}')
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.to(torch.float32).contiguous()
        return _native.mix_matrix(x, self.resident_matrix(), torch.empty_like(x))


class Tap(PassThroughResult):
    """Stand-in for the reference's side-effect taps (vis.Wave / vis.Spec / files.FileWriter): forwards its
    input unchanged, enabled or not; the original class name and state are kept for round-tripping."""

    def __init__(self, cls_name: str = '', **state):
        super().__init__()
        self.original_cls_name = cls_name
        self.original_state = state

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.VIS

    def _eval(self, request: Request) -> torch.Tensor:
        return self.input.forward(request)
