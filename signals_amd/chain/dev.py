"""`signals.chain.dev` names without PortAudio (reference src/signals/chain/dev.py:29-179): `SinkDevice` is the
headless `BlockDriver` -- the same stepping as `SinkDevice._callback` (position += frames per pull, exception stops
the stream, `seek`/`tell` in blocks) with the block handed back as a numpy array instead of written to a sound card."""
import typing

import attr

from signals_amd.chain import ChainLayerError
from signals_amd.chain.driver import BlockDriver


class BadPlaybackState(ChainLayerError):
    pass


@attr.s(auto_attribs=True, frozen=True, kw_only=True)
class DeviceInfo:
    """the fields of a PortAudio device record the reference keeps (dev.py:34-77); 'default' describes the headless sink"""
    name: str = 'default'
    index: int = 0
    hostapi: int = 0
    max_input_channels: int = 0
    max_output_channels: int = 2
    default_samplerate: float = 48000.0

    @property
    def is_source(self) -> bool:
        return self.max_input_channels > 0

    @property
    def is_sink(self) -> bool:
        return self.max_output_channels > 0


class SinkDevice(BlockDriver):

    def __init__(self, info: typing.Optional[DeviceInfo] = None, blocksize: int = 256):
        self.info = info or DeviceInfo()
        super().__init__(rate=int(self.info.default_samplerate), blocksize=blocksize)
        self._started = False

    def start(self) -> None:
        if self._started:
            raise BadPlaybackState('already started')
        self._started = True
        self.is_active = True

    def stop(self) -> None:
        if not self._started:
            raise BadPlaybackState('not started')
        self._started = False
