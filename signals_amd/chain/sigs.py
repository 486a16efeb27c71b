"""Loader for `.sigs` patch files: the `add` / `con` / `sink` subset of the reference's command
language (reference src/signals/map/control.py:291-330 `+`, :422-465 `>`; value syntax
map/__init__.py:104-148; coordinates :54-101), enough to load its two fixture patches
(src/signals/vis_test.sigs, lowpass_test.sigs) verbatim:

    sink 7a default                                   -> a headless BlockDriver at 7a
    + 1a signals.chain.fixed.Fixed enabled=true value=[[440]]
    > 1a 2a.hertz                                     -> node at 2a: .hertz = node at 1a

Editing, undo/redo and the GUI are out of scope.  The plot taps
(`signals.chain.vis.Wave/Spec`) load as their headless pass-through namesakes in `signals_amd.chain.vis`;
`signals.chain.files.FileWriter/FileReader` resolve to the WAV taps in `signals_amd.chain.files`.
"""
from __future__ import annotations

import copy
import json
import pathlib
import re
import shlex
import typing

import numpy as np

from signals_amd import SignalsError
from signals_amd.chain import Signal
from signals_amd.chain.driver import BlockDriver, load_signal

_COORD = re.compile(r'(\d+)([a-z]+)')


class PatchError(SignalsError):
    pass


def parse_coordinates(text: str) -> tuple[int, int]:
    """'1a' -> (1, 1), '1aa' -> (1, 27), '1234aul' -> (1234, 1234)  (map/__init__.py:80-101)"""
    match = _COORD.fullmatch(text)
    if not match:
        raise PatchError(f'bad coordinates {text!r}')
    row, letters = match.groups()
    col = 0
    for ch in letters:
        col = col * 26 + (ord(ch) - ord('a') + 1)
    if int(row) < 1:
        raise PatchError(f'bad coordinates {text!r}')
    return int(row), col


def parse_value(text: str):
    """json where it parses, lists as numpy arrays, anything else the raw string
    (map/__init__.py:131-139): `[[220]]` -> int64 array, `true` -> True, `/tmp/x.wav` -> str."""
    try:
        value = json.loads(text)
    except ValueError:
        return text
    return np.array(value) if isinstance(value, list) else value


class Patch:
    """Nodes by coordinates, plus the sinks (BlockDrivers) the patch declared."""

    def __init__(self, rate: int = 48000, blocksize: int = 256):
        self.rate, self.blocksize = rate, blocksize
        self.nodes: dict[tuple[int, int], Signal] = {}
        self.sinks: dict[tuple[int, int], BlockDriver] = {}

    def __getitem__(self, at: str) -> Signal:
        return self.nodes[parse_coordinates(at)]

    # -- commands
    def add(self, at: str, cls_name: str, state: dict) -> Signal:
        key = parse_coordinates(at)
        if key in self.nodes:
            raise PatchError(f'{at} is occupied')
        node = load_signal(cls_name)()
        new_state = copy.copy(node.get_state())
        for k, v in state.items():
            if k not in node.state_attrs():
                raise PatchError(f'{cls_name} at {at} has no property {k!r}')
            setattr(new_state, k, v)
        node.set_state(new_state)
        self.nodes[key] = node
        return node

    def sink(self, at: str, device: str = 'default') -> BlockDriver:
        key = parse_coordinates(at)
        driver = BlockDriver(rate=self.rate, blocksize=self.blocksize)
        self.nodes[key] = self.sinks[key] = driver
        return driver

    def connect(self, input_at: str, output: str) -> None:
        at, _, port_name = output.partition('.')
        src, dst = self[input_at], self[at]
        if port_name not in getattr(dst, 'port_names', lambda: [])():
            raise PatchError(f'{type(dst).__name__} at {at} has no port {port_name!r}')
        setattr(dst, port_name, src)

    def exec_line(self, line: str) -> None:
        words = shlex.split(line, comments=True)
        if not words:
            return
        cmd, args = words[0], words[1:]
        if cmd in ('+', 'add'):
            self.add(args[0], args[1], dict((k, parse_value(v)) for k, _, v in (a.partition('=') for a in args[2:])))
        elif cmd in ('>', 'con'):
            self.connect(*args)
        elif cmd == 'sink':
            self.sink(*args)
        else:
            raise PatchError(f'unsupported command {cmd!r} (only +, >, sink are loaded)')


def loads(text: str, **kw) -> Patch:
    patch = Patch(**kw)
    for line in text.splitlines():
        patch.exec_line(line)
    return patch


def load(path: typing.Union[str, pathlib.Path], **kw) -> Patch:
    return loads(pathlib.Path(path).read_text(), **kw)
