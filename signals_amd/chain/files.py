"""Sound-file taps (reference src/signals/chain/files.py:23-102): `FileWriter` passes its input through and
records it, `FileReader` plays a file into the graph.  These are the two places audio crosses between HBM
and the host on this path (device->host copy at the writer, host->device at the reader); SURVEY.md §8f-3.

The reference uses `soundfile` (libsndfile), which is not in this image; the container format here is a
plain RIFF/WAVE file written and read with the standard library only: `subtype` 'PCM_16' (soundfile's
default for .wav; samples = round(clip(x, -1, 1) * 32767)) or 'FLOAT' (32-bit float, lossless for this
engine's float32 buffers).  Like the reference, every block is written/read AT its request position
(files.py:53-55), so out-of-order and repeated blocks land where they belong.
"""
import abc
import pathlib
import struct
import typing

import attr
import numpy as np
import torch

from signals_amd import SignalFlags, runtime
from signals_amd.chain import (
    BadStateValue,
    Emitter,
    PassThroughResult,
    Request,
    result_dtype,
    state,
)

_SUBTYPES = {'PCM_16': (1, 2), 'FLOAT': (3, 4)}          # WAVE format tag, bytes per sample
_HEADER = 44


def _validate_subtype(instance, attribute, value):
    if value not in _SUBTYPES:
        raise BadStateValue(instance, attribute.name, value, f'one of {sorted(_SUBTYPES)}')


class _WaveFile:
    """Minimal random-access RIFF/WAVE: canonical 44-byte header, one data chunk."""

    def __init__(self, path: pathlib.Path, mode: str, samplerate: int = 0, channels: int = 0, subtype: str = 'PCM_16'):
        self.mode = mode
        self.path = path
        if mode == 'w':
            self.samplerate, self.channels, self.subtype = samplerate, channels, subtype
            self.frames = 0
            self._fh = open(path, 'w+b')
            self._write_header()
        else:
            self._fh = open(path, 'rb')
            head = self._fh.read(_HEADER)
            if len(head) < _HEADER or head[:4] != b'RIFF' or head[8:12] != b'WAVE' or head[12:16] != b'fmt ' \
                    or head[36:40] != b'data':
                raise ValueError(f'{path}: not a canonical RIFF/WAVE file')
            tag, self.channels, self.samplerate, _, _, bits = struct.unpack('<HHIIHH', head[20:36])
            self.subtype = {(1, 16): 'PCM_16', (3, 32): 'FLOAT'}.get((tag, bits))
            if self.subtype is None:
                raise ValueError(f'{path}: unsupported WAVE encoding tag={tag} bits={bits}')
            self.frames = struct.unpack('<I', head[40:44])[0] // (self.channels * _SUBTYPES[self.subtype][1])

    @property
    def _frame_bytes(self) -> int:
        return self.channels * _SUBTYPES[self.subtype][1]

    def _write_header(self) -> None:
        tag, width = _SUBTYPES[self.subtype]
        data = self.frames * self._frame_bytes
        self._fh.seek(0)
        self._fh.write(b'RIFF' + struct.pack('<I', 36 + data) + b'WAVEfmt ' +
                       struct.pack('<IHHIIHH', 16, tag, self.channels, self.samplerate,
                                   self.samplerate * self._frame_bytes, self._frame_bytes, 8 * width) +
                       b'data' + struct.pack('<I', data))

    def write(self, position: int, block: np.ndarray) -> None:
        block = np.broadcast_to(block, (block.shape[0], self.channels))
        if self.subtype == 'PCM_16':
            raw = np.rint(np.clip(block, -1.0, 1.0) * 32767.0).astype('<i2')
        else:
            raw = block.astype('<f4')
        if position > self.frames:                                  # gap: silence, like a sparse seek
            self._fh.seek(_HEADER + self.frames * self._frame_bytes)
            self._fh.write(bytes((position - self.frames) * self._frame_bytes))
        self._fh.seek(_HEADER + position * self._frame_bytes)
        self._fh.write(raw.tobytes())
        self.frames = max(self.frames, position + block.shape[0])
        self._write_header()
        self._fh.flush()

    def read(self, position: int, frames: int) -> np.ndarray:
        have = max(0, min(frames, self.frames - position))
        self._fh.seek(_HEADER + position * self._frame_bytes)
        raw = self._fh.read(have * self._frame_bytes)
        if self.subtype == 'PCM_16':
            data = np.frombuffer(raw, dtype='<i2').astype(np.float64) / 32768.0     # libsndfile's normalisation
        else:
            data = np.frombuffer(raw, dtype='<f4').astype(np.float64)
        return data.reshape(have, self.channels)                   # short read at end of file, like soundfile

    def close(self) -> None:
        self._fh.close()


class SoundFileBase(Emitter, abc.ABC):

    def __init__(self):
        super().__init__()
        self._buffer: typing.Optional[_WaveFile] = None

    @state
    class State(Emitter.State):
        path: str = attr.ib(default='/dev/null')
        subtype: str = attr.ib(default='PCM_16', validator=_validate_subtype)

    @property
    def _file_path(self) -> pathlib.Path:
        return pathlib.Path(self._state.path)

    def _open(self, mode: str, rate: int, channels: int) -> _WaveFile:
        buf = self._buffer
        if buf is not None and (buf.mode != mode or buf.samplerate != rate or buf.path != self._file_path):
            self._close()
            buf = None
        if buf is None:
            buf = self._buffer = _WaveFile(self._file_path, mode, rate, channels, self._state.subtype)
            if mode == 'r' and buf.samplerate != rate:
                raise ValueError(f'{self._file_path}: file rate {buf.samplerate} != requested {rate}')
        return buf

    def _close(self) -> None:
        if self._buffer is not None:
            self._buffer.close()
            self._buffer = None

    def destroy(self) -> None:
        self._close()
        super().destroy()


class FileReader(SoundFileBase):

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.GENERATOR

    @property
    def channels(self) -> int:
        return self._buffer.channels

    def read_rows(self, position: int, frames: int, rate: int, channels: int) -> torch.Tensor:
        block = self._open('r', rate, channels).read(position, frames)
        return torch.from_numpy(np.ascontiguousarray(block)).to(device=runtime.device(), dtype=result_dtype(frames))

    def _eval(self, request: Request) -> torch.Tensor:
        loc = request.loc
        return self.read_rows(loc.position, loc.shape.frames, loc.rate, loc.shape.channels)


class FileWriter(SoundFileBase, PassThroughResult):

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags() | SignalFlags.RECORDER

    def write_rows(self, position: int, rate: int, channels: int, block: torch.Tensor) -> None:
        """device -> host copy at the tap; synchronises the render stream"""
        self._open('w', rate, channels).write(position, block.detach().to('cpu', torch.float64).numpy())

    def _eval(self, request: Request) -> torch.Tensor:
        result = self.input.forward(request)
        self.write_rows(request.loc.position, request.loc.rate, request.loc.shape.channels, result)
        return result
