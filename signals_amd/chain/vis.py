"""Headless stand-ins for the reference's plot taps (reference src/signals/chain/vis.py:19-90): same class names,
state attributes and pass-through behaviour, so patches and scripts that name `signals.chain.vis.Wave` load and
render unchanged; nothing is drawn (matplotlib/Qt are out of scope) and no block is queued for a UI."""
import abc

import attr

from signals_amd.chain import state
from signals_amd.chain.ext import Tap


class Vis(Tap, abc.ABC):
    """pass-through with the VIS flag (vis.py:19-64); the engine shares its input's buffer"""


class Wave(Vis):
    @state
    class State(Vis.State):
        min_amp: float = attr.ib(default=-1.)
        max_amp: float = attr.ib(default=+1.)


class Spec(Vis):
    @state
    class State(Vis.State):
        min_freq: float = attr.ib(default=0)
        max_freq: float = attr.ib(default=22000)
        bands: int = attr.ib(default=80)
