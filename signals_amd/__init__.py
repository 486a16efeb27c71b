"""signals_amd -- MI355X-native block-render engine behind the `signals` node API.

Package root: the three names the render path needs from the reference's package root
(reference src/signals/__init__.py:15-64) -- `PortName`, `SignalsError`, `SignalFlags` -- without
the Qt application shell that lives beside them there (out of scope).

`install_as_signals()` aliases this package as `signals` in `sys.modules`, so graph scripts and
`.sigs` patches that name `signals.chain.osc.Sine` resolve to the classes here unchanged.
"""
import enum
import sys

PortName = str

__all__ = ['PortName', 'SignalsError', 'SignalFlags', 'install_as_signals']


class SignalsError(Exception):

    def __str__(self) -> str:
        return ' '.join((type(self).__name__, *map(str, self.args)))


class SignalFlags(enum.Flag):
    """Node classification bits (reference src/signals/__init__.py:27-58); same names and
    compositions, so `flags()` results compare equal by name."""
    CYCLIC = enum.auto()
    SINK_DEVICE = enum.auto()
    SOURCE_DEVICE = enum.auto()
    DEVICE = SINK_DEVICE | SOURCE_DEVICE
    GENERATOR = enum.auto()
    EFFECT = enum.auto()
    AUDIO = GENERATOR | EFFECT | SOURCE_DEVICE
    EPOCH = enum.auto()
    RECORDER = enum.auto()
    VIS = enum.auto()
    PASSTHRU = enum.auto()
    SIDE_EFFECT = VIS | RECORDER | PASSTHRU


def install_as_signals(host_plugins: bool = True) -> None:
    """Make `import signals.chain.osc` (and friends) resolve to this package.  A script that calls this declares
    itself written against the reference's API, so with `host_plugins` (default) node classes defined outside this
    package are handed float64 numpy arrays at their ports, as the reference would (chain/__init__.py:245-247);
    their numpy replies are accepted at every port regardless."""
    import importlib
    names = ['', '.discovery', '.chain', '.chain.osc', '.chain.fx', '.chain.fixed', '.chain.noise',
             '.chain.shape', '.chain.ext', '.chain.files', '.chain.driver', '.chain.sigs',
             '.chain.vis', '.chain.dev', '.chain.discovery']
    for suffix in names:
        mod = importlib.import_module(__name__ + suffix)
        sys.modules['signals' + suffix] = mod
    # the command language (signals.map.control) is out of scope; scripts import it next to the chain modules
    # (scripts/edited_sine.py:9), so the name resolves -- to the `.sigs` loader, its add / con / sink subset
    sigs = importlib.import_module(__name__ + '.chain.sigs')
    sys.modules.setdefault('signals.map', sigs)
    sys.modules.setdefault('signals.map.control', sigs)
    from signals_amd.chain import nodes
    nodes.host_plugins(host_plugins)
