"""`signals.chain.dev` without PortAudio (reference src/signals/chain/dev.py:29-179).  `SinkDevice` keeps the
reference's life cycle -- `open / close / start / stop / is_open / is_active / seek / tell`, the `_callback` that requests
one block at `frame_position`, copies it into `outdata` and advances, and "an exception stops the stream" (dev.py:167-179)
-- around a NULL audio device: a worker thread that calls `_callback` once per block period, like PortAudio's callback
thread would, and drops the samples (or hands them to `on_block`).  Blocks are rendered through `BlockDriver.pull`, i.e.
by the batched engine as batches of one (one fused launch per block where the graph allows)."""
import sys
import threading
import time
import traceback
import typing

import attr
import attrs.validators
import numpy as np

from signals_amd import runtime
from signals_amd.chain import ChainLayerError, state
from signals_amd.chain.driver import BlockDriver


class BadPlaybackState(ChainLayerError):
    pass


@attr.s(auto_attribs=True, frozen=True, kw_only=True, order=False)
class DeviceInfo:
    """the fields of a PortAudio device record the reference keeps (dev.py:34-77); `default` describes the null sink"""
    name: str = 'default'
    index: int = 0
    hostapi: int = 0
    max_input_channels: int = 2
    max_output_channels: int = 2
    default_low_input_latency: float = 0.0
    default_low_output_latency: float = 0.0
    default_high_input_latency: float = 0.0
    default_high_output_latency: float = 0.0
    default_samplerate: float = 48000.0

    @property
    def is_source(self) -> bool:
        return self.max_input_channels > 0

    @property
    def is_sink(self) -> bool:
        return self.max_output_channels > 0

    def __lt__(self, other: 'DeviceInfo') -> bool:
        return self.index < other.index

    def __str__(self) -> str:
        return f'{self.index} {self.name} ({self.max_input_channels} in, {self.max_output_channels} out)'


class CallbackStop(Exception):
    """raised by a callback to end the stream (sounddevice.CallbackStop in the reference, dev.py:176)"""


class NullOutputStream:
    """The part of `sounddevice.OutputStream` the reference's SinkDevice uses, with no sound card behind it: `start()` runs
    `callback(outdata, frames, time, status)` on a thread of its own once per block period until `stop()`, `close()` or a
    `CallbackStop`.  `realtime=False` calls back to back (offline rendering through the device API)."""

    def __init__(self, callback, channels: int, samplerate: float = 48000.0, blocksize: int = 256, realtime: bool = True):
        self.callback, self.channels, self.samplerate, self.blocksize, self.realtime = callback, channels, samplerate, blocksize, realtime
        self._thread: typing.Optional[threading.Thread] = None
        self._run = threading.Event()
        self.closed = False

    @property
    def active(self) -> bool:
        return self._thread is not None and self._thread.is_alive() and self._run.is_set()

    def _loop(self) -> None:
        if runtime.device().type == 'cuda':
            import torch
            torch.cuda.set_device(runtime.device())          # the current device is per thread
        outdata = np.zeros((self.blocksize, self.channels), dtype=np.float32)
        period = self.blocksize / self.samplerate
        deadline = time.perf_counter()
        while self._run.is_set():
            try:
                self.callback(outdata, self.blocksize, None, None)
            except CallbackStop:
                break
            if self.realtime:
                deadline += period
                delay = deadline - time.perf_counter()
                if delay > 0:
                    time.sleep(delay)
                else:
                    deadline = time.perf_counter()             # fell behind: no burst to catch up
        self._run.clear()

    def start(self) -> None:
        if self.closed:
            raise BadPlaybackState('The output stream is closed')
        if not self.active:
            self._run.set()
            self._thread = threading.Thread(target=self._loop, name='signals-null-sink', daemon=True)
            self._thread.start()

    def stop(self) -> None:
        self._run.clear()
        if self._thread is not None and self._thread is not threading.current_thread():
            self._thread.join()
        self._thread = None

    def close(self) -> None:
        self.stop()
        self.closed = True


class SinkDevice(BlockDriver):

    def __init__(self, info: typing.Optional[DeviceInfo] = None, blocksize: int = 256, realtime: bool = True):
        info = info or DeviceInfo()

        @state
        class State(BlockDriver.State):
            channels: int = attr.ib(default=1, validator=attrs.validators.in_(
                range(1, max(info.max_input_channels, info.max_output_channels) + 1)))

        self.State = State
        self.info = info
        self.realtime = realtime
        self.on_block: typing.Optional[typing.Callable[[np.ndarray], None]] = None     # gets a copy of every played block
        self._stream: typing.Optional[NullOutputStream] = None
        super().__init__(rate=int(info.default_samplerate), blocksize=blocksize)

    def log(self, msg: typing.Any) -> None:
        print(msg, file=sys.stderr)

    def set_state(self, new_state) -> None:
        super().set_state(new_state)
        if self.is_open and self._stream.channels != new_state.channels:      # dev.py:109-117: re-open with the new width
            active = self.is_active
            self.close()
            if active:
                self.start()
            else:
                self.open()

    def destroy(self) -> None:
        if self.is_open:
            self.close()
        super().destroy()

    @property
    def is_open(self) -> bool:
        return self._stream is not None

    @property
    def is_active(self) -> bool:
        return self.is_open and self._stream.active

    @is_active.setter
    def is_active(self, value: bool) -> None:        # BlockDriver.pull marks a failed stream inactive: here the stream itself knows
        pass

    def open(self) -> None:
        if self.is_open:
            raise BadPlaybackState('The output stream is already open')
        self._stream = NullOutputStream(self._callback, self._state.channels, self.info.default_samplerate, self.blocksize,
                                        self.realtime)

    def close(self) -> None:
        if self.is_open:
            self._stream.close()
            self._stream = None
        else:
            raise BadPlaybackState('The output stream is not open')

    def start(self) -> None:
        if not self.is_open:
            self.open()
        self._stream.start()

    def stop(self) -> None:
        if self.is_active:
            self._stream.stop()
        else:
            raise BadPlaybackState('The output stream is not active')

    def _callback(self, outdata: np.ndarray, frames: int, time_: typing.Any, status: typing.Any) -> None:
        """dev.py:167-179: one block at frame_position into outdata, advance; an exception ends the stream"""
        if status:
            self.log(status)
        try:
            block = self.pull(frames)                         # requests BlockLoc(frame_position, frames x channels), advances
        except Exception:
            self.log(traceback.format_exc())
            raise CallbackStop
        outdata[:, :block.shape[1]] = block
        if self.on_block is not None:
            self.on_block(outdata.copy())
