"""The C-ABI library loads on a CPU-only host and exports every symbol include/signals_amd.h declares;
argument validation returns hipErrorInvalidValue without touching a device."""
import ctypes
import pathlib
import re

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope='module')
def lib():
    from signals_amd import _native
    if not _native.LIB_PATH.exists():
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(str(_native.LIB_PATH))


def declared_symbols():
    text = (ROOT / 'include' / 'signals_amd.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|int64_t)\s+(sig_\w+)\s*\(', text)))


def test_header_and_binding_agree():
    from signals_amd import _native
    assert declared_symbols() == sorted(_native.EXPORTS)


def test_every_declared_symbol_is_exported(lib):
    for name in declared_symbols():
        assert getattr(lib, name) is not None, name
    assert lib.sig_abi_version() == 7


def test_argument_errors_do_not_reach_the_device(lib):
    inv = 1     # hipErrorInvalidValue
    lib.sig_osc_bank.restype = ctypes.c_int
    lib.sig_osc_bank.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32,
                                 ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                 ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
    assert lib.sig_osc_bank(0, 0, 48000, 16, 4, None, 1, None, 1, None, 0, 4, None) == inv      # null pointers
    assert lib.sig_osc_bank(0, -1, 48000, 16, 4, 8, 1, None, 1, 8, 0, 4, None) == inv           # negative position
    assert lib.sig_osc_bank(0, 0, 48000, 16, 4, 8, 1, None, 1, 8, 0, 2, None) == inv            # ld < voices
    assert lib.sig_osc_bank(0, 0, 48000, 0, 4, 8, 1, None, 1, 8, 0, 4, None) == 0               # empty: no launch
    lib.sig_mix_matrix.restype = ctypes.c_int
    lib.sig_mix_matrix.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    assert lib.sig_mix_matrix(32, 100, 16, 100, 16, 16, 100, None) == inv                        # voices % 64
    # a control program longer than its register file (one register per instruction, SIG_CTL_MAX_INS == SIG_CTL_MAX_REGS == 48)
    lib.sig_control_program.restype = ctypes.c_int
    lib.sig_control_program.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    assert lib.sig_control_program(48000, 0, 256, 4, 8, -1, 0, 16, 49, 16, 1, None) == inv
    assert lib.sig_control_program(48000, 0, 256, 4, 8, -1, 0, 16, 64, 16, 1, None) == inv
    assert lib.sig_control_program(48000, 0, 256, 0, 8, -1, 0, 16, 48, 16, 1, None) == 0             # accepted; no blocks: no launch


def test_fused_geometry_needs_no_device():
    """sig_fused_geometry is pure host logic: the bench's f64 operation model calls it"""
    from signals_amd import _native
    assert _native.fused_geometry(1024, 256, 4096, 100) == (4, 8)        # bench default: 2048 waves of 4 voices x 8 blocks
    assert _native.fused_geometry(1024, 256, 1024, 100) == (4, 2)
    assert _native.fused_geometry(1024, 256, 256, 100) == (4, 1)
    assert _native.fused_geometry(1024, 64, 4096, 100)[1] == 1          # N < ctx: no spans
    vpt, span = _native.fused_geometry(1024, 256, 1, 100)                # latency mode: spread the voices
    assert vpt == 1 and span == 1


def test_voice_bus_plan_names_the_bench_kernel():
    """sig_fused_voice_bus_plan is pure host logic: the bench geometry runs the closed form at 8 voices x 8 blocks per lane"""
    from signals_amd import _native
    plan = _native.fused_voice_bus_plan('Sine', 0, 1024, 256, 4096, 100)
    assert plan == {'voices_per_lane': 8, 'blocks_per_lane': 8, 'closed_form': True, 'kernel': 'fused_steady_bus_kernel<8, C>'}
    assert _native.fused_voice_bus_plan('Sawtooth', 0, 1024, 256, 4096, 100)['closed_form'] is False
    assert _native.fused_voice_bus_plan('Sine', 0, 1024, 64, 8, 100)['closed_form'] is False     # N < ctx at position 0
    _native.set_fused_tuning(closed_form=0)
    try:
        assert _native.fused_voice_bus_plan('Sine', 0, 1024, 256, 4096, 100)['closed_form'] is False
    finally:
        _native.set_fused_tuning()
    assert _native.fused_voice_bus_plan('Sine', 0, 1024, 256, 4096, 100)['closed_form'] is True
