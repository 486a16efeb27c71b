"""Process-wide render context: which GPU this process renders on, and the lazily-checked device
status words kernels OR error bits into (include/signals_amd.h SIG_STATUS_*).

One process per GPU: the device is `cuda:LOCAL_RANK` (torchrun) unless `set_device` says otherwise.
Without a GPU the node API still imports and wires graphs (CPU tensors for plumbing such as `Fixed`),
but every kernel call raises `signals_amd._native.NativeError` -- there is no CPU compute path.
"""
from __future__ import annotations

import os
import weakref

import torch

_device: torch.device | None = None
_status_words: 'weakref.WeakSet' = weakref.WeakSet()


def device() -> torch.device:
    global _device
    if _device is None:
        if torch.cuda.is_available():
            _device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
            torch.cuda.set_device(_device)
        else:
            _device = torch.device('cpu')
    return _device


def set_device(dev) -> None:
    global _device
    _device = torch.device(dev)
    if _device.type == 'cuda':
        torch.cuda.set_device(_device)


class StatusWord:
    """One int32 on the device that kernels OR error bits into; read back only at sync points."""

    def __init__(self, owner_name: str):
        self.owner_name = owner_name
        self.tensor = torch.zeros(1, dtype=torch.int32, device=device())
        _status_words.add(self)

    def poll(self) -> int:
        bits = int(self.tensor.item())
        if bits:
            self.tensor.zero_()
        return bits


def check_status() -> None:
    """Raise what the reference would have raised inside the block (scipy's ValueError for a
    critical frequency outside (0, 1), fx.py:99-121).  Costs one device sync per live status word:
    call at the sink edge, not per node."""
    from signals_amd._native import STATUS_BAD_CUTOFF
    for word in list(_status_words):
        bits = word.poll()
        if bits & STATUS_BAD_CUTOFF:
            raise ValueError(f'{word.owner_name}: Digital filter critical frequencies must be 0 < Wn < 1')
