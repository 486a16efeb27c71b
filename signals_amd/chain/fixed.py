"""`Fixed`: constant source -- how per-voice parameters (hertz, cutoff, gain) enter a graph
(reference src/signals/chain/fixed.py:21-39).  The state value stays a 2-D numpy array so scripts
(`get_state().value = np.array([[330]])`) and `.sigs` values (`value=[[220]]`, int64) work as they
are; the device copy is uploaded once and re-uploaded only when the array changes -- assignment of another
array or an in-place edit, which is detected by comparing the array's bytes with a snapshot on every reply
(arrays of every size: ~0.4 us for 1024 doubles, ~16 us for 70 000, proportional beyond)."""
import ctypes

import attr
import numpy as np
import torch

from signals_amd import SignalFlags, runtime
from signals_amd.chain import (
    BadStateValue,
    Emitter,
    Request,
    Shape,
    result_dtype,
    state,
)

_SMALL_BYTES = 1 << 15      # snapshots up to this size are kept as bytes (one memcpy + memcmp, ~0.4 us for 1024 doubles)
_libc = ctypes.CDLL(None)
_libc.memcmp.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
_libc.memcmp.restype = ctypes.c_int


class _Snapshot:
    """What a `Fixed` array held when it was uploaded, so that an in-place edit of ANY size is seen at the next reply
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


def _validate_array(instance, attribute, new_value):
    if not (isinstance(new_value, np.ndarray) and new_value.ndim == 2):
        raise BadStateValue(instance, attribute.name, new_value, 'must be a 2D array')


def _unit_zero() -> np.ndarray:
    return np.zeros(Shape.unit())


class Fixed(Emitter):
    @state
    class State(Emitter.State):
        value: np.ndarray = attr.ib(
            factory=_unit_zero,
            validator=_validate_array,
            on_setattr=attr.setters.validate,
        )

    def __init__(self):
        super().__init__()
        self._resident = None       # (source array, host snapshot, device tensor)

    @classmethod
    def flags(cls) -> SignalFlags:
        return super().flags()

    @property
    def channels(self) -> int:
        return Shape.of_array(self._state.value).channels

    def resident(self) -> torch.Tensor:
        """Device copy of `value`: float64 for one-row (control) values -- integers from `.sigs` files
        are exact below 2**53, like numpy's int64 -> float64 promotion in osc.py:32 -- float32 for
        multi-row (audio) values."""
        value = self._state.value
        held = self._resident
        if held is not None and held[0] is value and held[1].matches(value):
            return held[2]
        dtype = result_dtype(value.shape[0])
        host = np.ascontiguousarray(value, dtype=np.float64)
        tensor = torch.from_numpy(host.copy()).to(device=runtime.device(), dtype=dtype)
        self._resident = (value, _Snapshot(value), tensor)
        return tensor

    def _eval(self, request: Request) -> torch.Tensor:
        return self.resident()
