"""Noise generators (reference src/signals/chain/noise.py:13-23).  `White` is uniform on [0, 1) like
`np.random.rand`; the reference draws from numpy's global unseeded RNG, so parity is statistical
only.  Here a sample is a hash of (seed, frame, channel): reproducible and
position-pure, `seed` being an extra state attribute (default 0)."""
import abc

import attr
import attrs.validators
import torch

from signals_amd import SignalFlags, _native, runtime
from signals_amd.chain import (
    BlockCachingEmitter,
    ExplicitChannelsEmitter,
    Request,
    result_dtype,
    state,
)


class Noise(ExplicitChannelsEmitter, BlockCachingEmitter, abc.ABC):

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.GENERATOR


class White(Noise):
    @state
    class State(Noise.State):
        seed: int = attr.ib(validator=attrs.validators.ge(0), default=0)

    def _eval(self, request: Request) -> torch.Tensor:
        frames, channels = request.loc.shape
        out = torch.empty((frames, channels), dtype=result_dtype(frames), device=runtime.device())
        return _native.white_noise(self._state.seed, request.loc.position, out)
