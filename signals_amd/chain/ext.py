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
    Receiver,
    Request,
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
        if held is None or held[0] is not gains or not np.array_equal(held[1], gains):
            host = np.ascontiguousarray(gains, dtype=np.float64)
            self._resident = held = (gains, host.copy(), torch.from_numpy(host.copy()).to(runtime.device()))
        return held[2]

    def _eval(self, request: Request) -> torch.Tensor:
        voices = self.input.channels
        x = self.input.request(request.loc.reslice(voices))
        out = torch.empty((x.shape[0], self.channels), dtype=result_dtype(x.shape[0]), device=x.device)
        return _native.sum_bus(x, self.resident_gains(), out)
