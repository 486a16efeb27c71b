"""`signals.chain.discovery` (reference src/signals/chain/discovery.py:19-140): `load_signal` resolves the qualified
class names `.sigs` patches and scripts use, with the reference's error types; `Library.scan` lists the node classes
of this package and of user plugin directories -- any concrete `Signal` subclass defined in a scanned module is a
node (:71-93).  The device `Rack` (:96-126) is not provided: audio devices are out of scope."""
import abc
import importlib
import importlib.util
import pathlib
import sys
import typing

import signals_amd.chain
from signals_amd import SignalFlags, SignalsError
from signals_amd.discovery import is_concrete_subclass


class DiscoveryError(SignalsError):
    pass


class BadSignal(DiscoveryError, abc.ABC):
    pass


class BadSyntax(BadSignal):

    def __init__(self, cls_qualname: str):
        super().__init__(f'{cls_qualname!r} is not a valid signal name')


class BadPath(BadSignal):

    def __init__(self, cls_qualname: str, reason: str):
        super().__init__(f'Failed to load {cls_qualname!r}: {reason}')


class InvalidObject(BadSignal):

    def __init__(self, cls_qualname: str, o: object):
        super().__init__(f'Python object {cls_qualname}={o!r} is not a signal')


def load_signal(qualname: str) -> typing.Type[signals_amd.chain.Signal]:
    """`signals.chain.osc.Sine` (or any importable `package.module.Class`) -> the class; `signals.` names resolve to
    this package whether or not `install_as_signals()` was called"""
    module, dot, cls_name = qualname.rpartition('.')
    if not dot or not all(part.isidentifier() for part in qualname.split('.')):
        raise BadSyntax(qualname)
    if module == 'signals' or module.startswith('signals.'):
        module = 'signals_amd' + module[len('signals'):]
    try:
        cls = getattr(importlib.import_module(module), cls_name)
    except (AttributeError, ImportError) as e:
        raise BadPath(qualname, str(e.args[0]) if e.args else repr(e))
    if not is_concrete_subclass(cls, signals_amd.chain.Signal):
        raise InvalidObject(qualname, cls)
    return cls


class Library:
    """names of the node classes found in this package's `chain/` and in the given plugin files / package directories"""

    def __init__(self, paths: typing.Iterable[pathlib.Path] = ()):
        self.paths = [pathlib.Path(signals_amd.chain.__file__).parent, *map(pathlib.Path, paths)]
        self.names: list[str] = []

    @staticmethod
    def _modules(path: pathlib.Path):
        if path.is_dir():
            files = sorted(path.glob('*.py'))
        elif path.is_file():
            files = [path]
        else:
            raise FileNotFoundError(path)
        own = pathlib.Path(signals_amd.chain.__file__).parent
        for file in files:
            if file.parent == own:
                name = 'signals_amd.chain' + ('' if file.stem == '__init__' else '.' + file.stem)
                yield importlib.import_module(name)
                continue
            name = file.stem if file.stem != '__init__' else file.parent.name
            if name in sys.modules and getattr(sys.modules[name], '__file__', None) == str(file):
                yield sys.modules[name]
                continue
            spec = importlib.util.spec_from_file_location(name, file)
            module = importlib.util.module_from_spec(spec)
            sys.modules[name] = module
            spec.loader.exec_module(module)
            yield module

    def scan(self) -> None:
        names = []
        for path in self.paths:
            for module in self._modules(path):
                for k, v in vars(module).items():
                    if (not k.startswith('_') and getattr(v, '__module__', None) == module.__name__
                            and is_concrete_subclass(v, signals_amd.chain.Signal)
                            and not (v.flags() & SignalFlags.DEVICE)):
                        names.append(v.cls_name())
        self.names[:] = names
