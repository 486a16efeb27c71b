"""Naming helpers the node API relies on (reference src/signals/discovery.py:11-12, :59-71).
Module scanning (`Library.scan`) is out of scope; `load_signal` for `.sigs` names lives in
`signals_amd.chain.driver`."""
import inspect
import typing


def is_concrete_subclass(o: typing.Any, superclass: type, *, allow_abstract: bool = False) -> bool:
    return isinstance(o, type) and issubclass(o, superclass) and (allow_abstract or not inspect.isabstract(o))


def qualname(type_: type) -> str:
    """Qualified class name as written in `.sigs` patches.  Classes of this package report the
    reference's package name so dumps stay loadable by the reference."""
    module = type_.__module__
    if module == 'signals_amd' or module.startswith('signals_amd.'):
        module = 'signals' + module[len('signals_amd'):]
    return f'{module}.{type_.__qualname__}'


class Named:

    @classmethod
    def cls_name(cls) -> str:
        return qualname(cls)
