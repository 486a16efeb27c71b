"""ctypes binding of `signals_amd/csrc/libsignals_amd.so` (C ABI: include/signals_amd.h).

torch tensors are the buffer substrate; only their `data_ptr()`, strides and the current HIP stream
cross the boundary.  There is NO CPU fallback: a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes
import pathlib

import torch

import os

# SIG_LIB_PATH: load another build of the same ABI (kernel tuning A/B runs only)
LIB_PATH = pathlib.Path(os.environ.get('SIG_LIB_PATH') or pathlib.Path(__file__).resolve().parent / 'csrc' / 'libsignals_amd.so')

F32, F64 = 0, 1
OSC_KINDS = {'Sine': 0, 'Square': 1, 'Sawtooth': 2, 'Triangle': 3}
FILT_TYPES = {'lp': 0, 'hp': 1, 'bp': 2, 'bs': 3}
EW_OPS = {'Gain': 0, 'Mix': 1, 'RingMod': 2, 'Amp': 3}
STATUS_BAD_CUTOFF = 1
ABI_VERSION = 7
SINE_FAST_MAX_CYCLES = 2.0 ** 26     # sig_osc.h kSineFastMaxT: |t| up to which the fused Sine kernels advance the phase incrementally

EXPORTS = ('sig_abi_version', 'sig_osc_bank', 'sig_osc_bank_mod', 'sig_biquad_coldstart', 'sig_elementwise', 'sig_sum_bus',
           'sig_white_noise', 'sig_adsr', 'sig_mix_matrix', 'sig_fused_osc_biquad',
           'sig_fused_voice_bus', 'sig_fused_voice_bus_workspace', 'sig_band_coldstart',
           'sig_fused_osc_biquad_devpos', 'sig_advance_position', 'sig_adsr_apply', 'sig_biquad_coldstart_env',
           'sig_fused_geometry', 'sig_biquad_coldstart_bus', 'sig_fused_osc_biquad_mix', 'sig_latency_voice_bus',
           'sig_latency_voice_bus_workspace', 'sig_fused_voice_bus_prepared', 'sig_fused_voice_consts_size',
           'sig_fused_voice_bus_plan', 'sig_fused_set_tuning', 'sig_fused_voice_bus_walk',
           'sig_fused_cascade_bus', 'sig_fused_cascade_geometry', 'sig_fused_cascade_set_tuning',
           'sig_fused_osc_biquad_rows', 'sig_fused_voice_bus_rows', 'sig_fused_osc_pair_biquad', 'sig_fused_voice_pair_bus',
           'sig_fused_osc_biquad_fm', 'sig_fused_voice_bus_fm', 'sig_control_program',
           'sig_voice_program', 'sig_voice_program_set_tuning', 'sig_voice_program_geometry', 'sig_voice_program_args_size',
           'sig_voice_program_attach', 'sig_voice_program_detach_all', 'sig_voice_program_use_attached',
           'sig_control_program_attach', 'sig_control_program_attached', 'sig_fused_voice_bus_bound')


class NativeError(RuntimeError):
    """The HIP library is missing, was given a CPU tensor, or returned a hipError_t."""


class CtlIns(ctypes.Structure):
    """sig_ctl_ins: one instruction of a block-rate control program"""
    _fields_ = [('op', ctypes.c_int32), ('kind', ctypes.c_int32), ('a', ctypes.c_int32), ('b', ctypes.c_int32), ('c', ctypes.c_int32),
                ('dst', ctypes.c_int32), ('stride', ctypes.c_int32), ('rows', ctypes.c_int32), ('cols', ctypes.c_int32),
                ('reserved', ctypes.c_int32), ('row', ctypes.c_void_p)]


class CtlOut(ctypes.Structure):
    """sig_ctl_out: a register written to a (nblocks, cols) float64 output"""
    _fields_ = [('reg', ctypes.c_int32), ('cols', ctypes.c_int32), ('out', ctypes.c_void_p), ('front', ctypes.c_void_p)]


CTL_OPS = {'Row': 0, 'Osc': 1, 'Gain': 2, 'Mix': 3, 'RingMod': 4, 'Amp': 5}
CTL_MAX_REGS, CTL_MAX_INS = 48, 48


# ---- sig_voice_program: the per-voice graph as code for the accumulator machine of voice_program.hip
VP_OPS = {'Osc': 0, 'Filter': 1, 'Gain': 2, 'Mul': 3, 'Mix': 4, 'Save': 5, 'Load': 6, 'Const': 7, 'Amp': 8, 'Adsr': 9, 'Noise': 10}
VP_MAX_INS, VP_MAX_OSCS, VP_MAX_PARAMS, VP_MAX_FILTERS, VP_MAX_TEMPS, VP_MAX_HIST = 32, 4, 8, 4, 4, 3


class VpIns(ctypes.Structure):
    _fields_ = [('op', ctypes.c_int32), ('kind', ctypes.c_int32), ('a', ctypes.c_int32), ('b', ctypes.c_int32), ('c', ctypes.c_int32)]


class VpRows(ctypes.Structure):
    _fields_ = [('ptr', ctypes.c_void_p), ('col_stride', ctypes.c_int32), ('rows', ctypes.c_int32)]


class VoiceProgramT(ctypes.Structure):
    """sig_voice_program_t (host memory)"""
    _fields_ = [('n_ins', ctypes.c_int32), ('ins', VpIns * VP_MAX_INS),
                ('n_oscs', ctypes.c_int32), ('hertz', VpRows * VP_MAX_OSCS), ('phase', VpRows * VP_MAX_OSCS),
                ('n_params', ctypes.c_int32), ('params', VpRows * VP_MAX_PARAMS),
                ('n_filters', ctypes.c_int32), ('cutoff', VpRows * VP_MAX_FILTERS), ('filter_type', ctypes.c_int32 * VP_MAX_FILTERS),
                ('filter_level', ctypes.c_int32 * VP_MAX_FILTERS),
                ('n_temps', ctypes.c_int32), ('depth', ctypes.c_int32),
                ('adsr', ctypes.c_void_p * 6), ('adsr_stride', ctypes.c_int32 * 6), ('noise_seed', ctypes.c_uint64 * 2)]


class Operand(ctypes.Structure):
    _fields_ = [('ptr', ctypes.c_void_p), ('row_stride', ctypes.c_int64),
                ('col_stride', ctypes.c_int32), ('dtype', ctypes.c_int32),
                ('row_div', ctypes.c_int32), ('reserved', ctypes.c_int32)]


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise NativeError(f'{LIB_PATH} not built: run `python -c "import __graft_entry__ as g; g.build()"` '
                              f'(signals_amd/csrc/build.sh).  There is no CPU fallback.')
        L = ctypes.CDLL(str(LIB_PATH))
        i32, i64, vp, dp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p
        L.sig_abi_version.restype = ctypes.c_int
        L.sig_abi_version.argtypes = []
        L.sig_osc_bank.restype = ctypes.c_int
        L.sig_osc_bank.argtypes = [ctypes.c_int, i64, i32, i64, i32, dp, i32, dp, i32, vp, i32, i64, vp]
        L.sig_biquad_coldstart.restype = ctypes.c_int
        L.sig_biquad_coldstart.argtypes = [ctypes.c_int, i32, i64, i32, i32, i32, i32, dp, i32, i32,
                                           vp, i64, i64, vp, i64, i32, vp, vp]
        L.sig_elementwise.restype = ctypes.c_int
        L.sig_elementwise.argtypes = [ctypes.c_int, i64, i32, ctypes.POINTER(Operand), ctypes.POINTER(Operand),
                                      ctypes.POINTER(Operand), vp, i64, i32, vp]
        L.sig_sum_bus.restype = ctypes.c_int
        L.sig_sum_bus.argtypes = [i64, i32, vp, i64, i32, dp, i64, i32, vp, i64, i32, vp]
        L.sig_white_noise.restype = ctypes.c_int
        L.sig_white_noise.argtypes = [ctypes.c_uint64, i64, i64, i32, vp, i32, i64, vp]
        L.sig_adsr.restype = ctypes.c_int
        L.sig_adsr.argtypes = [i64, i32, i64, i32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32),
                               vp, i32, i64, vp]
        L.sig_mix_matrix.restype = ctypes.c_int
        L.sig_mix_matrix.argtypes = [i64, i32, vp, i64, vp, vp, i64, vp]
        L.sig_fused_osc_biquad.restype = ctypes.c_int
        L.sig_fused_osc_biquad.argtypes = [ctypes.c_int, ctypes.c_int, i32, i64, i32, i32, i32, i32,
                                           dp, i32, dp, i32, dp, i32, dp, i32, vp, i64, vp, vp]
        L.sig_fused_voice_bus_workspace.restype = ctypes.c_int64
        L.sig_fused_voice_bus_workspace.argtypes = [i32, i64, i32]
        L.sig_fused_voice_bus.restype = ctypes.c_int
        L.sig_fused_voice_bus.argtypes = [ctypes.c_int, ctypes.c_int, i32, i64, i32, i32, i32, i32,
                                          dp, i32, dp, i32, dp, i32, dp, i32, dp, i64, i32, vp, vp, i64, vp, vp]
        L.sig_fused_voice_bus_walk.restype = ctypes.c_int
        L.sig_fused_voice_bus_walk.argtypes = L.sig_fused_voice_bus.argtypes
        L.sig_fused_voice_consts_size.restype = ctypes.c_int64
        L.sig_fused_voice_consts_size.argtypes = [i32]
        L.sig_fused_voice_bus_prepared.restype = ctypes.c_int
        L.sig_fused_voice_bus_prepared.argtypes = [ctypes.c_int, ctypes.c_int, i32, i64, i32, i32, i32, i32,
                                                   dp, i32, dp, i32, dp, i32, dp, i32, dp, i64, i32, vp, vp, i64, vp, vp, vp, i32]
        L.sig_band_coldstart.restype = ctypes.c_int
        L.sig_band_coldstart.argtypes = [ctypes.c_int, i32, i64, i32, i32, i32, i32, dp, i32, dp, i32,
                                         vp, i64, i64, vp, i64, i32, vp, vp]
        L.sig_osc_bank_mod.restype = ctypes.c_int
        L.sig_osc_bank_mod.argtypes = [ctypes.c_int, i64, i64, i32, i64, i32, i32, dp, i32, i64, dp, i32, i64, vp, i32, i64, vp]
        L.sig_fused_osc_biquad_devpos.restype = ctypes.c_int
        L.sig_fused_osc_biquad_devpos.argtypes = [ctypes.c_int, ctypes.c_int, i32, vp, i32, i32, i32, i32,
                                                  dp, i32, dp, i32, dp, i32, dp, i32, vp, i64, vp, vp]
        L.sig_adsr_apply.restype = ctypes.c_int
        L.sig_adsr_apply.argtypes = [i64, i32, i64, i32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32),
                                     vp, i64, vp, i64, vp]
        L.sig_biquad_coldstart_env.restype = ctypes.c_int
        L.sig_biquad_coldstart_env.argtypes = [ctypes.c_int, i32, i64, i32, i32, i32, i32, dp, i32, i32,
                                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32),
                                               vp, i64, i64, vp, i64, vp, vp]
        L.sig_advance_position.restype = ctypes.c_int
        L.sig_advance_position.argtypes = [vp, i64, vp]
        L.sig_biquad_coldstart_bus.restype = ctypes.c_int
        L.sig_biquad_coldstart_bus.argtypes = [ctypes.c_int, i32, i64, i32, i32, i32, i32, dp, i32, i32,
                                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32),
                                               vp, i64, i64, vp, i64, i32, vp, vp, i64, vp, vp]
        L.sig_fused_osc_biquad_mix.restype = ctypes.c_int
        L.sig_fused_osc_biquad_mix.argtypes = [ctypes.c_int, ctypes.c_int, i32, i64, i32, i32, i32, i32,
                                               dp, i32, dp, i32, dp, i32, dp, i32, vp, vp, i64, vp, vp]
        L.sig_latency_voice_bus_workspace.restype = ctypes.c_int64
        L.sig_latency_voice_bus_workspace.argtypes = [i32, i32, i32]
        L.sig_latency_voice_bus.restype = ctypes.c_int
        L.sig_latency_voice_bus.argtypes = [ctypes.c_int, i32, i64, vp, i32, i32, i32, dp, i32, dp, i32, dp, i32, dp, i32,
                                            dp, i64, i32, vp, vp, i64, vp, vp]
        L.sig_fused_geometry.restype = ctypes.c_int
        L.sig_fused_geometry.argtypes = [i32, i32, i32, i32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
        L.sig_fused_cascade_bus.restype = ctypes.c_int
        L.sig_fused_cascade_bus.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, i32, i64, i64, i32, i32, i32, i32,
                                            dp, i32, dp, i32, dp, i32, dp, i32, dp, i32,
                                            ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32),
                                            dp, i64, i32, vp, vp, i64, vp, vp]
        L.sig_fused_cascade_geometry.restype = ctypes.c_int
        L.sig_fused_cascade_geometry.argtypes = [i32, i32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
        L.sig_fused_osc_biquad_rows.restype = ctypes.c_int
        L.sig_fused_osc_biquad_rows.argtypes = [ctypes.c_int, ctypes.c_int, i32, i64, i32, i32, i32, i32,
                                                dp, i32, dp, i32, dp, i32, i32, dp, i32, i32, vp, i64, vp, vp]
        L.sig_fused_voice_bus_rows.restype = ctypes.c_int
        L.sig_fused_voice_bus_rows.argtypes = [ctypes.c_int, ctypes.c_int, i32, i64, i32, i32, i32, i32,
                                               dp, i32, dp, i32, dp, i32, i32, dp, i32, i32, dp, i64, i32, vp, vp, i64, vp, vp]
        L.sig_fused_osc_biquad_fm.restype = ctypes.c_int
        L.sig_fused_osc_biquad_fm.argtypes = [ctypes.c_int, ctypes.c_int, i32, i64, i32, i32, i32, i32,
                                              dp, i32, i32, dp, dp, i32, i32, dp, dp, i32, i32, dp, i32, i32, vp, i64, vp, vp]
        L.sig_fused_voice_bus_fm.restype = ctypes.c_int
        L.sig_fused_voice_bus_fm.argtypes = [ctypes.c_int, ctypes.c_int, i32, i64, i32, i32, i32, i32,
                                             dp, i32, i32, dp, dp, i32, i32, dp, dp, i32, i32, dp, i32, i32, dp, i64, i32, vp, vp, i64, vp, vp]
        L.sig_control_program.restype = ctypes.c_int
        L.sig_control_program.argtypes = [i32, i64, i32, i32, i32, i64, i64, vp, i32, vp, i32, vp]
        L.sig_control_program_attach.restype = ctypes.c_int
        L.sig_control_program_attach.argtypes = [ctypes.POINTER(i32), i32, ctypes.c_char_p, ctypes.POINTER(i32)]
        L.sig_control_program_attached.restype = ctypes.c_int
        L.sig_control_program_attached.argtypes = [i32, i32, i64, i32, i32, i32, i64, i64, vp, i32, vp, i32, vp]
        L.sig_fused_voice_bus_bound.restype = ctypes.c_int
        L.sig_fused_voice_bus_bound.argtypes = [vp, i64, vp, i32, i32, vp]
        L.sig_voice_program.restype = ctypes.c_int
        L.sig_voice_program.argtypes = [ctypes.POINTER(VoiceProgramT), i32, i64, i32, i32, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_int64), i32,
                                        dp, i64, i32, vp, vp, i64, vp, vp]
        L.sig_voice_program_set_tuning.restype = ctypes.c_int
        L.sig_voice_program_set_tuning.argtypes = [i32, i32]
        L.sig_voice_program_geometry.restype = ctypes.c_int
        L.sig_voice_program_geometry.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32)]
        L.sig_voice_program_args_size.restype = ctypes.c_int64
        L.sig_voice_program_args_size.argtypes = []
        L.sig_voice_program_attach.restype = ctypes.c_int
        L.sig_voice_program_attach.argtypes = [ctypes.POINTER(VoiceProgramT), i32, i32, ctypes.c_char_p]
        L.sig_voice_program_detach_all.restype = ctypes.c_int
        L.sig_voice_program_detach_all.argtypes = []
        L.sig_voice_program_use_attached.restype = ctypes.c_int
        L.sig_voice_program_use_attached.argtypes = [i32]
        L.sig_fused_osc_pair_biquad.restype = ctypes.c_int
        L.sig_fused_osc_pair_biquad.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32, i64, i32, i32, i32, i32,
                                                dp, i32, dp, i32, dp, i32, dp, i32, dp, i32,
                                                dp, i32, i32, dp, i32, i32, vp, i64, vp, vp]
        L.sig_fused_voice_pair_bus.restype = ctypes.c_int
        L.sig_fused_voice_pair_bus.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, i32, i64, i32, i32, i32, i32,
                                               dp, i32, dp, i32, dp, i32, dp, i32, dp, i32,
                                               dp, i32, i32, dp, i32, i32, dp, i64, i32, vp, vp, i64, vp, vp]
        L.sig_fused_cascade_set_tuning.restype = ctypes.c_int
        L.sig_fused_cascade_set_tuning.argtypes = [i32, i32]
        L.sig_fused_voice_bus_plan.restype = ctypes.c_int
        L.sig_fused_voice_bus_plan.argtypes = [ctypes.c_int, i64, i32, i32, i32, i32] + [ctypes.POINTER(ctypes.c_int32)] * 3
        L.sig_fused_set_tuning.restype = ctypes.c_int
        L.sig_fused_set_tuning.argtypes = [i32, i32, i32, i32]
        if L.sig_abi_version() != ABI_VERSION:
            raise NativeError('libsignals_amd.so ABI version mismatch')
        _lib = L
    return _lib


def _check(err: int, what: str) -> None:
    if err != 0:
        raise NativeError(f'{what} failed: hipError_t {err}')


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise NativeError(f'unsupported buffer dtype {t.dtype}')


def _gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NativeError('HIP kernels need tensors resident on an MI355X (got a CPU tensor); no CPU fallback')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(t: torch.Tensor) -> int:
    """the hipStream_t torch currently launches on for `t`'s device"""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)              # the same handle as below, without building a Stream object
    return torch.cuda.current_stream(t.device).cuda_stream


def current_stream_handle(device_index: int) -> int:
    """the raw hipStream_t torch currently launches on"""
    if _raw_stream is not None:
        return _raw_stream(device_index)
    return torch.cuda.current_stream(device_index).cuda_stream


def _ctrl_row(t: torch.Tensor | None, what: str):
    """(ptr, stride) of an f64 control row shaped (1,V) or (1,1)."""
    if t is None:
        return None, 0
    if t.dtype != torch.float64 or t.dim() != 2 or t.shape[0] != 1 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise NativeError(f'{what}: control rows are contiguous float64 (1,V) or (1,1), got {tuple(t.shape)} {t.dtype}')
    return t.data_ptr(), (0 if t.shape[1] == 1 else 1)


def _audio(t: torch.Tensor, what: str) -> None:
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise NativeError(f'{what}: audio buffers are 2-D with contiguous channels, got strides {t.stride()}')


def _ctrl_rows(t: torch.Tensor | None, what: str):
    """(ptr, col_stride, row_stride, rows) of f64 control rows shaped (R, V) or (R, 1)"""
    if t is None:
        return None, 0, 0, 1
    if t.dtype != torch.float64 or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise NativeError(f'{what}: control rows are float64 (R,V) or (R,1) with contiguous channels, got '
                          f'{tuple(t.shape)} {t.dtype}')
    return t.data_ptr(), (0 if t.shape[1] == 1 else 1), (0 if t.shape[0] == 1 else t.stride(0)), t.shape[0]


def osc_bank(kind: str, position: int, rate: int, hertz: torch.Tensor, phase: torch.Tensor | None,
             out: torch.Tensor, step: int = 1, rows_per_param: int = 0) -> torch.Tensor:
    """out[(rows, voices)] <- oscillator `kind`; row r is absolute frame `position + r*step`.
    hertz/phase: (1|P, V|1) f64; with P > 1 parameter rows, output row r uses row r // rows_per_param."""
    _gpu(hertz, phase, out)
    _audio(out, 'osc out')
    rows, voices = out.shape
    hp, hs, hrs, hrows = _ctrl_rows(hertz, 'hertz')
    pp, ps, prs, prows = _ctrl_rows(phase, 'phase')
    for row, name in ((hertz, 'hertz'), (phase, 'phase')):
        if row is not None and row.shape[1] not in (1, voices):
            raise NativeError(f'{name} has {row.shape[1]} channels for {voices} voices')
    if step == 1 and hrows == 1 and prows == 1:
        _check(lib().sig_osc_bank(OSC_KINDS[kind], position, rate, rows, voices, hp, hs, pp, ps,
                                  out.data_ptr(), _dt(out), out.stride(0), _stream(out)), 'sig_osc_bank')
        return out
    if max(hrows, prows) > 1:
        if rows_per_param < 1:
            raise NativeError('per-block oscillator parameters need rows_per_param')
        need = (rows + rows_per_param - 1) // rows_per_param
        for n, name in ((hrows, 'hertz'), (prows, 'phase')):
            if n not in (1, need):
                raise NativeError(f'{name} has {n} parameter rows, launch needs 1 or {need}')
    else:
        rows_per_param = 0
    _check(lib().sig_osc_bank_mod(OSC_KINDS[kind], position, step, rate, rows, voices, rows_per_param,
                                  hp, hs, hrs, pp, ps, prs, out.data_ptr(), _dt(out), out.stride(0), _stream(out)),
           'sig_osc_bank_mod')
    return out


def biquad_coldstart(btype: str, rate: int, position: int, block_frames: int, nblocks: int, context: int,
                     cutoff: torch.Tensor, buf: torch.Tensor, history: int, out: torch.Tensor,
                     status: torch.Tensor | None = None, envelope: dict | None = None) -> torch.Tensor:
    """`buf` holds `history` context rows followed by nblocks*block_frames input rows;
    `out` (nblocks*block_frames, voices) receives the filtered blocks.
    cutoff: f64 (1|nblocks, V|1).  `envelope`: ADSR control rows (name -> (1,V)|(1,1) f64); the stored rows
    are then multiplied by the envelope (float32 buffers only)."""
    _gpu(cutoff, buf, out, status, *(envelope or {}).values())
    _audio(buf, 'biquad in')
    _audio(out, 'biquad out')
    rows, voices = out.shape
    if rows != block_frames * nblocks or buf.shape[0] != history + rows or buf.shape[1] != voices:
        raise NativeError(f'biquad shapes: in {tuple(buf.shape)} history {history} out {tuple(out.shape)} '
                          f'blocks {nblocks}x{block_frames}')
    if buf.dtype != out.dtype:
        raise NativeError('biquad in/out dtype differ')
    if cutoff.dtype != torch.float64 or cutoff.dim() != 2 or not cutoff.is_contiguous():
        raise NativeError('cutoff must be a contiguous float64 2-D tensor')
    if cutoff.shape[0] not in (1, nblocks) or cutoff.shape[1] not in (1, voices):
        raise NativeError(f'cutoff shape {tuple(cutoff.shape)} vs blocks {nblocks} voices {voices}')
    if cutoff.shape[1] != voices and voices != 1:
        # the reference indexes crit[0, i] for every channel i (fx.py:99)
        raise IndexError(f'index {cutoff.shape[1]} is out of bounds for axis 1 with size {cutoff.shape[1]}')
    in_ptr = buf.data_ptr() + history * buf.stride(0) * buf.element_size()
    if envelope is not None:
        if out.dtype != torch.float32:
            raise NativeError('the envelope epilogue is float32 only')
        ptrs = (ctypes.c_void_p * 6)()
        strides = (ctypes.c_int32 * 6)()
        for i, name in enumerate(ADSR_PARAMS):
            ptrs[i], strides[i] = _ctrl_row(envelope[name], name)
            if envelope[name].shape[1] not in (1, voices):
                raise NativeError(f'{name} has {envelope[name].shape[1]} channels for {voices} voices')
        _check(lib().sig_biquad_coldstart_env(FILT_TYPES[btype], rate, position, block_frames, nblocks, context, voices,
                                              cutoff.data_ptr(), 0 if cutoff.shape[1] == 1 else 1, cutoff.shape[0],
                                              ptrs, strides, in_ptr, buf.stride(0), history, out.data_ptr(), out.stride(0),
                                              status.data_ptr() if status is not None else None, _stream(out)),
               'sig_biquad_coldstart_env')
        return out
    _check(lib().sig_biquad_coldstart(FILT_TYPES[btype], rate, position, block_frames, nblocks, context, voices,
                                      cutoff.data_ptr(), 0 if cutoff.shape[1] == 1 else 1, cutoff.shape[0],
                                      in_ptr, buf.stride(0), history, out.data_ptr(), out.stride(0), _dt(out),
                                      status.data_ptr() if status is not None else None, _stream(out)),
           'sig_biquad_coldstart')
    return out


def _operand(t: torch.Tensor, rows: int, cols: int, what: str) -> Operand:
    """numpy-broadcast operand; an operand with R rows where rows % R == 0 is a per-block control operand
    (R blocks of rows // R frames each)."""
    if t.dim() != 2 or t.shape[1] not in (1, cols) or t.shape[0] < 1 or rows % t.shape[0]:
        raise NativeError(f'{what}: shape {tuple(t.shape)} does not broadcast to {(rows, cols)}')
    rs = 0 if t.shape[0] == 1 else t.stride(0)
    cs = 0 if t.shape[1] == 1 else t.stride(1)
    row_div = 0 if t.shape[0] in (1, rows) else rows // t.shape[0]
    return Operand(t.data_ptr(), rs, cs, _dt(t), row_div, 0)


def elementwise(op: str, a: torch.Tensor, b: torch.Tensor, c: torch.Tensor | None, out: torch.Tensor) -> torch.Tensor:
    _gpu(a, b, c, out)
    _audio(out, 'elementwise out')
    rows, cols = out.shape
    A = _operand(a, rows, cols, 'a')
    B = _operand(b, rows, cols, 'b')
    C = _operand(c, rows, cols, 'c') if c is not None else None
    _check(lib().sig_elementwise(EW_OPS[op], rows, cols, ctypes.byref(A), ctypes.byref(B),
                                 ctypes.byref(C) if C is not None else None,
                                 out.data_ptr(), out.stride(0), _dt(out), _stream(out)), 'sig_elementwise')
    return out


def sum_bus(x: torch.Tensor, gains: torch.Tensor | None, out: torch.Tensor) -> torch.Tensor:
    _gpu(x, gains, out)
    _audio(x, 'bus in')
    _audio(out, 'bus out')
    rows, voices = x.shape
    bus = out.shape[1]
    if out.shape[0] != rows:
        raise NativeError('bus rows mismatch')
    gp, gld = None, 0
    if gains is not None:
        if gains.dtype != torch.float64 or gains.dim() != 2 or gains.shape != (bus, voices) or gains.stride(1) != 1:
            raise NativeError(f'gains must be float64 ({bus},{voices}), got {tuple(gains.shape)} {gains.dtype}')
        gp, gld = gains.data_ptr(), gains.stride(0)
    # the kernel is built for 1, 2 and 4 bus channels; any other width is a few calls over column groups
    c0, esz = 0, out.element_size()
    while c0 < bus:
        w = 4 if bus - c0 >= 4 else (2 if bus - c0 >= 2 else 1)
        _check(lib().sig_sum_bus(rows, voices, x.data_ptr(), x.stride(0), _dt(x),
                                 gp + c0 * gld * 8 if gp is not None else None, gld, w,
                                 out.data_ptr() + c0 * esz, out.stride(0), _dt(out), _stream(out)), 'sig_sum_bus')
        c0 += w
    return out


def white_noise(seed: int, position: int, out: torch.Tensor) -> torch.Tensor:
    _gpu(out)
    _audio(out, 'noise out')
    _check(lib().sig_white_noise(seed, position, out.shape[0], out.shape[1], out.data_ptr(), _dt(out),
                                 out.stride(0), _stream(out)), 'sig_white_noise')
    return out


ADSR_PARAMS = ('attack', 'decay', 'sustain', 'release', 'gate_on', 'gate_off')


def adsr(position: int, rate: int, rows: dict, out: torch.Tensor) -> torch.Tensor:
    """rows: name -> f64 control row (1,V)|(1,1) for each of ADSR_PARAMS"""
    _gpu(out, *rows.values())
    _audio(out, 'adsr out')
    ptrs = (ctypes.c_void_p * 6)()
    strides = (ctypes.c_int32 * 6)()
    for i, name in enumerate(ADSR_PARAMS):
        ptrs[i], strides[i] = _ctrl_row(rows[name], name)
        if rows[name].shape[1] not in (1, out.shape[1]):
            raise NativeError(f'{name} has {rows[name].shape[1]} channels for {out.shape[1]} voices')
    _check(lib().sig_adsr(position, rate, out.shape[0], out.shape[1], ptrs, strides, out.data_ptr(), _dt(out),
                          out.stride(0), _stream(out)), 'sig_adsr')
    return out


def mix_matrix(x: torch.Tensor, matrix: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    _gpu(x, matrix, out)
    _audio(x, 'mix_matrix in')
    _audio(out, 'mix_matrix out')
    if x.dtype != torch.float32 or out.dtype != torch.float32 or matrix.dtype != torch.float32:
        raise NativeError('mix_matrix is float32 in / float32 out')
    if matrix.shape != (64, 64) or not matrix.is_contiguous() or x.shape != out.shape or x.shape[1] % 64:
        raise NativeError(f'mix_matrix shapes: x {tuple(x.shape)} matrix {tuple(matrix.shape)} out {tuple(out.shape)}')
    _check(lib().sig_mix_matrix(x.shape[0], x.shape[1], x.data_ptr(), x.stride(0), matrix.data_ptr(),
                                out.data_ptr(), out.stride(0), _stream(out)), 'sig_mix_matrix')
    return out


def fused_osc_biquad(kind: str, btype: str, rate: int, position, block_frames: int, nblocks: int, context: int,
                     hertz: torch.Tensor, phase: torch.Tensor | None, cutoff: torch.Tensor,
                     gain: torch.Tensor | None, out: torch.Tensor, status: torch.Tensor | None = None) -> torch.Tensor:
    """out (nblocks*block_frames, voices) f32 <- [gain *] Filter(Osc), every block cold-started.
    `position`: an int, or a one-element int64 device tensor read by the kernel (hipGraph replay)."""
    _gpu(hertz, phase, cutoff, gain, out, status)
    _audio(out, 'fused out')
    rows, voices = out.shape
    if out.dtype != torch.float32 or rows != block_frames * nblocks:
        raise NativeError(f'fused out must be float32 ({block_frames * nblocks}, V), got {tuple(out.shape)} {out.dtype}')
    ptrs = []
    for row, name in ((hertz, 'hertz'), (phase, 'phase'), (cutoff, 'cutoff'), (gain, 'gain')):
        if row is not None and row.shape[1] not in (1, voices):
            raise NativeError(f'{name} has {row.shape[1]} channels for {voices} voices')
        ptrs.extend(_ctrl_row(row, name))
    if isinstance(position, torch.Tensor):
        if position.dtype != torch.int64 or position.numel() != 1 or not position.is_cuda:
            raise NativeError('device position must be a one-element int64 GPU tensor')
        _check(lib().sig_fused_osc_biquad_devpos(OSC_KINDS[kind], FILT_TYPES[btype], rate, position.data_ptr(), block_frames,
                                                 nblocks, context, voices, *ptrs, out.data_ptr(), out.stride(0),
                                                 status.data_ptr() if status is not None else None, _stream(out)),
               'sig_fused_osc_biquad_devpos')
        return out
    _check(lib().sig_fused_osc_biquad(OSC_KINDS[kind], FILT_TYPES[btype], rate, position, block_frames, nblocks, context,
                                      voices, *ptrs, out.data_ptr(), out.stride(0),
                                      status.data_ptr() if status is not None else None, _stream(out)),
           'sig_fused_osc_biquad')
    return out


def biquad_coldstart_bus(btype: str, rate: int, position: int, block_frames: int, nblocks: int, context: int,
                         cutoff: torch.Tensor, buf: torch.Tensor, history: int, bus_gains: torch.Tensor | None,
                         out: torch.Tensor, envelope: dict | None = None, workspace: torch.Tensor | None = None,
                         status: torch.Tensor | None = None) -> torch.Tensor:
    """out (nblocks*block_frames, C) f32 <- sum over voices of bus_gains * [envelope *] Filter(buf); `buf` holds
    `history` context rows followed by the input rows, like `biquad_coldstart`; `envelope`: ADSR control rows"""
    _gpu(cutoff, buf, out, status, bus_gains, *(envelope or {}).values())
    _audio(buf, 'biquad bus in')
    _audio(out, 'biquad bus out')
    rows, bus = out.shape
    voices = buf.shape[1]
    if out.dtype != torch.float32 or buf.dtype != torch.float32 or rows != block_frames * nblocks or buf.shape[0] != history + rows:
        raise NativeError(f'biquad bus shapes: in {tuple(buf.shape)} {buf.dtype} history {history} out {tuple(out.shape)} {out.dtype}')
    if cutoff.dtype != torch.float64 or cutoff.shape[1] not in (1, voices) or cutoff.shape[0] not in (1, nblocks) \
            or not cutoff.is_contiguous():
        raise NativeError(f'cutoff must be contiguous float64 (1|{nblocks}, 1|{voices}), got {tuple(cutoff.shape)} {cutoff.dtype}')
    gp, gld = None, 0
    if bus_gains is not None:
        if bus_gains.dtype != torch.float64 or bus_gains.shape != (bus, voices) or bus_gains.stride(1) != 1:
            raise NativeError(f'bus gains must be float64 ({bus},{voices}), got {tuple(bus_gains.shape)} {bus_gains.dtype}')
        gp, gld = bus_gains.data_ptr(), bus_gains.stride(0)
    elif bus != 1:
        raise NativeError('a bus without gains is mono')
    ptrs = strides = None
    if envelope is not None:
        ptrs = (ctypes.c_void_p * 6)()
        strides = (ctypes.c_int32 * 6)()
        for i, name in enumerate(ADSR_PARAMS):
            ptrs[i], strides[i] = _ctrl_row(envelope[name], name)
            if envelope[name].shape[1] not in (1, voices):
                raise NativeError(f'{name} has {envelope[name].shape[1]} channels for {voices} voices')
    need = lib().sig_fused_voice_bus_workspace(voices, rows, bus)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need // 8, dtype=torch.float64, device=out.device)
    in_ptr = buf.data_ptr() + history * buf.stride(0) * buf.element_size()
    _check(lib().sig_biquad_coldstart_bus(FILT_TYPES[btype], rate, position, block_frames, nblocks, context, voices,
                                          cutoff.data_ptr(), 0 if cutoff.shape[1] == 1 else 1, cutoff.shape[0],
                                          ptrs, strides, in_ptr, buf.stride(0), history, gp, gld, bus,
                                          workspace.data_ptr(), out.data_ptr(), out.stride(0),
                                          status.data_ptr() if status is not None else None, _stream(out)),
           'sig_biquad_coldstart_bus')
    return out


def latency_voice_bus_workspace(voices: int, block_frames: int, bus_channels: int, device) -> torch.Tensor:
    """zeroed scratch of `latency_voice_bus` (per-tile partials + the arrival counter, which must start at zero)"""
    return torch.zeros(lib().sig_latency_voice_bus_workspace(voices, block_frames, bus_channels) // 8, dtype=torch.float64,
                       device=device)


def latency_voice_bus(btype: str, rate: int, position, block_frames: int, context: int, voices: int,
                      hertz: torch.Tensor, phase: torch.Tensor | None, cutoff: torch.Tensor, gain: torch.Tensor | None,
                      bus_gains: torch.Tensor | None, out: torch.Tensor, workspace: torch.Tensor,
                      status: torch.Tensor | None = None) -> torch.Tensor:
    """out (block_frames, C) f32 <- one block of sum over voices of pan * [gain *] Filter(Sine), in one launch.
    `position`: an int, or a one-element int64 device tensor that the launch reads AND advances by block_frames."""
    _gpu(hertz, phase, cutoff, gain, bus_gains, out, workspace, status)
    _audio(out, 'latency bus out')
    rows, bus = out.shape
    if out.dtype != torch.float32 or rows != block_frames:
        raise NativeError(f'latency bus out must be float32 ({block_frames}, C), got {tuple(out.shape)} {out.dtype}')
    need = lib().sig_latency_voice_bus_workspace(voices, block_frames, bus)
    if workspace.dtype != torch.float64 or workspace.numel() * 8 < need:
        raise NativeError(f'latency workspace needs {need} bytes of float64 (latency_voice_bus_workspace)')
    ptrs = []
    for row, name in ((hertz, 'hertz'), (phase, 'phase'), (cutoff, 'cutoff'), (gain, 'gain')):
        if row is not None and row.shape[1] not in (1, voices):
            raise NativeError(f'{name} has {row.shape[1]} channels for {voices} voices')
        ptrs.extend(_ctrl_row(row, name))
    gp, gld = None, 0
    if bus_gains is not None:
        if bus_gains.dtype != torch.float64 or bus_gains.shape != (bus, voices) or bus_gains.stride(1) != 1:
            raise NativeError(f'bus gains must be float64 ({bus},{voices}), got {tuple(bus_gains.shape)} {bus_gains.dtype}')
        gp, gld = bus_gains.data_ptr(), bus_gains.stride(0)
    elif bus != 1:
        raise NativeError('a bus without gains is mono')
    pos_int, pos_dev = position, None
    if isinstance(position, torch.Tensor):
        if position.dtype != torch.int64 or position.numel() != 1 or not position.is_cuda:
            raise NativeError('device position must be a one-element int64 GPU tensor')
        pos_int, pos_dev = 0, position.data_ptr()
    _check(lib().sig_latency_voice_bus(FILT_TYPES[btype], rate, pos_int, pos_dev, block_frames, context, voices, *ptrs,
                                       gp, gld, bus, workspace.data_ptr(), out.data_ptr(), out.stride(0),
                                       status.data_ptr() if status is not None else None, _stream(out)),
           'sig_latency_voice_bus')
    return out


class LatencyVoiceBusCall:
    """`latency_voice_bus` with everything but the position and the output buffer validated and converted ONCE: the
    per-block host path of latency mode is then one ctypes call (the tensors are kept alive by this object)."""

    def __init__(self, btype: str, rate: int, block_frames: int, context: int, voices: int,
                 hertz: torch.Tensor, phase: torch.Tensor | None, cutoff: torch.Tensor, gain: torch.Tensor | None,
                 bus_gains: torch.Tensor | None, bus_channels: int, workspace: torch.Tensor,
                 status: torch.Tensor | None = None):
        _gpu(hertz, phase, cutoff, gain, bus_gains, workspace, status)
        need = lib().sig_latency_voice_bus_workspace(voices, block_frames, bus_channels)
        if workspace.dtype != torch.float64 or workspace.numel() * 8 < need:
            raise NativeError(f'latency workspace needs {need} bytes of float64 (latency_voice_bus_workspace)')
        ptrs = []
        for row, name in ((hertz, 'hertz'), (phase, 'phase'), (cutoff, 'cutoff'), (gain, 'gain')):
            if row is not None and row.shape[1] not in (1, voices):
                raise NativeError(f'{name} has {row.shape[1]} channels for {voices} voices')
            ptrs.extend(_ctrl_row(row, name))
        gp, gld = None, 0
        if bus_gains is not None:
            if bus_gains.dtype != torch.float64 or bus_gains.shape != (bus_channels, voices) or bus_gains.stride(1) != 1:
                raise NativeError(f'bus gains must be float64 ({bus_channels},{voices}), got {tuple(bus_gains.shape)} {bus_gains.dtype}')
            gp, gld = bus_gains.data_ptr(), bus_gains.stride(0)
        elif bus_channels != 1:
            raise NativeError('a bus without gains is mono')
        self._keep = (hertz, phase, cutoff, gain, bus_gains, workspace, status)
        self._fn = lib().sig_latency_voice_bus
        self._head = (FILT_TYPES[btype], rate)
        self._mid = (block_frames, context, voices, *ptrs, gp, gld, bus_channels, workspace.data_ptr())
        self._status = status.data_ptr() if status is not None else None
        self.shape = (block_frames, bus_channels)
        self.device = workspace.device

    def __call__(self, position: int, out: torch.Tensor) -> torch.Tensor:
        """`out`: a contiguous float32 (block_frames, bus_channels) tensor on the launch device"""
        err = self._fn(*self._head, position, None, *self._mid, out.data_ptr(), self.shape[1], self._status, _stream(out))
        if err:
            raise NativeError(f'sig_latency_voice_bus failed: hipError_t {err}')
        return out


class FusedVoiceBusCallT(ctypes.Structure):
    """sig_fused_voice_bus_call (host memory)"""
    _fields_ = [(n, ctypes.c_int32) for n in ('osc_kind', 'filt_type', 'rate', 'block_frames', 'nblocks', 'context', 'voices',
                                              'hertz_stride', 'phase_stride', 'cutoff_stride', 'gain_stride', 'bus_channels')] + \
               [(n, ctypes.c_void_p) for n in ('hertz', 'phase', 'cutoff', 'gain', 'bus_gains')] + \
               [('bus_gains_ld', ctypes.c_int64), ('out_ld', ctypes.c_int64), ('workspace', ctypes.c_void_p), ('status', ctypes.c_void_p),
                ('consts', ctypes.c_void_p)]


class FusedVoiceBusCall:
    """`fused_voice_bus` with caller-held closed-form constants (sig_fused_voice_bus_prepared / _walk) and everything but the
    position and the output buffer validated and converted ONCE: the per-batch host path of the batched engine is then one
    ctypes call (at 256 blocks per batch the launch takes ~18 us; the generic binding's per-call validation took longer than
    that).  The tensors are kept alive by this object."""

    def __init__(self, kind: str, btype: str, rate: int, block_frames: int, nblocks: int, context: int, voices: int,
                 hertz: torch.Tensor, phase: torch.Tensor | None, cutoff: torch.Tensor, gain: torch.Tensor | None,
                 bus_gains: torch.Tensor | None, bus_channels: int, workspace: torch.Tensor, status: torch.Tensor | None,
                 consts: torch.Tensor):
        _gpu(hertz, phase, cutoff, gain, bus_gains, workspace, status, consts)
        rows = block_frames * nblocks
        ptrs = []
        for row, name in ((hertz, 'hertz'), (phase, 'phase'), (cutoff, 'cutoff'), (gain, 'gain')):
            if row is not None and row.shape[1] not in (1, voices):
                raise NativeError(f'{name} has {row.shape[1]} channels for {voices} voices')
            ptrs.extend(_ctrl_row(row, name))
        gp, gld = None, 0
        if bus_gains is not None:
            if bus_gains.dtype != torch.float64 or bus_gains.shape != (bus_channels, voices) or bus_gains.stride(1) != 1:
                raise NativeError(f'bus gains must be float64 ({bus_channels},{voices}), got {tuple(bus_gains.shape)} {bus_gains.dtype}')
            gp, gld = bus_gains.data_ptr(), bus_gains.stride(0)
        elif bus_channels != 1:
            raise NativeError('a bus without gains is mono')
        if workspace.dtype != torch.float64 or workspace.numel() * 8 < lib().sig_fused_voice_bus_workspace(voices, rows, bus_channels):
            raise NativeError('fused bus workspace too small')
        if consts.dtype != torch.float64 or consts.numel() * 8 < lib().sig_fused_voice_consts_size(voices):
            raise NativeError('consts must be float64 of sig_fused_voice_consts_size(voices) bytes')
        self._keep = (hertz, phase, cutoff, gain, bus_gains, workspace, status, consts)
        self._head = (OSC_KINDS[kind], FILT_TYPES[btype], rate)
        self._mid = (block_frames, nblocks, context, voices, *ptrs, gp, gld, bus_channels, workspace.data_ptr())
        self._status = status.data_ptr() if status is not None else None
        self._consts = consts.data_ptr()
        self._prepared, self._walk = lib().sig_fused_voice_bus_prepared, lib().sig_fused_voice_bus_walk
        self.shape = (rows, bus_channels)
        # ... and as one block for sig_fused_voice_bus_bound: six arguments per call instead of 26
        (hp, hs), (pp, ps), (cp, cs), (gp_, gs) = [ptrs[2 * k: 2 * k + 2] for k in range(4)]
        self._block = FusedVoiceBusCallT(OSC_KINDS[kind], FILT_TYPES[btype], rate, block_frames, nblocks, context, voices, hs, ps, cs, gs,
                                         bus_channels, hp, pp, cp, gp_, gp, gld, bus_channels, workspace.data_ptr(), self._status,
                                         self._consts)
        self._block_ref = ctypes.byref(self._block)
        self._bound = lib().sig_fused_voice_bus_bound

    def __call__(self, position: int, out: torch.Tensor, consts_ready: bool, walk: bool = False, stream: int | None = None) -> torch.Tensor:
        """`out`: a contiguous float32 (nblocks * block_frames, bus_channels) tensor on the launch device; `stream`: a raw
        hipStream_t to launch on instead of torch's current one (the caller orders it against whoever reads `out`)"""
        s = _stream(out) if stream is None else stream
        err = self._bound(self._block_ref, position, out.data_ptr(), 1 if consts_ready else 0, 1 if walk else 0, s)
        if err:
            raise NativeError(f'sig_fused_voice_bus failed: hipError_t {err}')
        return out


def fused_geometry(voices: int, block_frames: int, nblocks: int, context: int) -> tuple[int, int]:
    """(voices per lane, blocks per lane) the fused kernels use for this problem size"""
    vpt, span = ctypes.c_int32(), ctypes.c_int32()
    _check(lib().sig_fused_geometry(voices, block_frames, nblocks, context, ctypes.byref(vpt), ctypes.byref(span)),
           'sig_fused_geometry')
    return vpt.value, span.value


def _param_rows(t: torch.Tensor | None, what: str, voices: int, nblocks: int):
    """(ptr, stride, rows) of a per-block parameter: float64 (1|nblocks, V|1), rows contiguous"""
    if t is None:
        return None, 0, 1
    if t.dtype != torch.float64 or t.dim() != 2 or not t.is_contiguous() or t.shape[1] not in (1, voices) or t.shape[0] not in (1, nblocks):
        raise NativeError(f'{what}: per-block parameters are contiguous float64 (1|{nblocks}, 1|{voices}), got {tuple(t.shape)} {t.dtype}')
    return t.data_ptr(), (0 if t.shape[1] == 1 else 1), t.shape[0]


def fused_rows(kind: str, btype: str, rate: int, position: int, block_frames: int, nblocks: int, context: int, voices: int,
               hertz: torch.Tensor, phase: torch.Tensor | None, cutoff: torch.Tensor, gain: torch.Tensor | None,
               out: torch.Tensor, bus_gains: torch.Tensor | None = None, bus: bool = False,
               workspace: torch.Tensor | None = None, status: torch.Tensor | None = None,
               pair: tuple | None = None, hertz_hist: torch.Tensor | None = None,
               phase_hist: torch.Tensor | None = None) -> torch.Tensor:
    """[gain *] Filter(Osc) with cutoff / gain rows read per block: out (nblocks*block_frames, voices) f32
    (sig_fused_osc_biquad_rows), or with `bus` the sum over voices weighted by bus_gains, out (.., C) (sig_fused_voice_bus_rows).
    `pair` = (op, kind2, hertz2, phase2, mix): the filter reads Mix (op 'Mix') or RingMod (op 'RingMod') of the oscillator
    above and a second one (sig_fused_osc_pair_biquad / sig_fused_voice_pair_bus).
    hertz / phase with nblocks rows: block-rate FM (sig_fused_osc_biquad_fm / sig_fused_voice_bus_fm); `hertz_hist` /
    `phase_hist` = the (1, .) row in front of the launch (the previous block's, whose samples are block 0's context)."""
    _gpu(hertz, phase, cutoff, gain, out, bus_gains, workspace, status, hertz_hist, phase_hist,
         *((pair[2], pair[3], pair[4]) if pair else ()))
    fm = hertz_hist is not None or phase_hist is not None or hertz.shape[0] > 1 or (phase is not None and phase.shape[0] > 1)
    if fm:
        if pair is not None:
            raise NativeError('block-rate FM and a second oscillator: no fused entry point')
        hp, hs, hrows = _param_rows(hertz, 'hertz', voices, nblocks)
        pp, ps, prows = _param_rows(phase, 'phase', voices, nblocks)

        def hist(t, like, rows, what):
            if t is None and rows == 1:
                return None
            if t is None or t.dtype != torch.float64 or tuple(t.shape) != (1, like.shape[1]) or not t.is_contiguous():
                raise NativeError(f'{what}: the row in front of the launch is float64 (1, {like.shape[1]})')
            return t.data_ptr()
        hh, ph_ = hist(hertz_hist, hertz, hrows, 'hertz_hist'), hist(phase_hist, phase, prows, 'phase_hist') if phase is not None else None
        cp, cs, crows = _param_rows(cutoff, 'cutoff', voices, nblocks)
        gp, gs, grows = _param_rows(gain, 'gain', voices, nblocks)
        st = status.data_ptr() if status is not None else None
        _audio(out, 'fused rows out')
        if out.dtype != torch.float32 or out.shape[0] != block_frames * nblocks:
            raise NativeError(f'fused rows out must be float32 ({block_frames * nblocks}, .), got {tuple(out.shape)} {out.dtype}')
        head = (OSC_KINDS[kind], FILT_TYPES[btype], rate, position, block_frames, nblocks, context, voices,
                hp, hs, hrows, hh, pp, ps, prows, ph_, cp, cs, crows, gp, gs, grows)
        if not bus:
            if out.shape[1] != voices:
                raise NativeError(f'fused rows out has {out.shape[1]} channels for {voices} voices')
            _check(lib().sig_fused_osc_biquad_fm(*head, out.data_ptr(), out.stride(0), st, _stream(out)), 'sig_fused_osc_biquad_fm')
            return out
        C = out.shape[1]
        bp, bld = None, 0
        if bus_gains is not None:
            if bus_gains.dtype != torch.float64 or bus_gains.shape != (C, voices) or bus_gains.stride(1) != 1:
                raise NativeError(f'bus gains must be float64 ({C},{voices}), got {tuple(bus_gains.shape)} {bus_gains.dtype}')
            bp, bld = bus_gains.data_ptr(), bus_gains.stride(0)
        elif C != 1:
            raise NativeError('a bus without gains is mono')
        need = lib().sig_fused_voice_bus_workspace(voices, out.shape[0], C)
        if workspace is None or workspace.numel() * workspace.element_size() < need:
            workspace = torch.empty(need // 8, dtype=torch.float64, device=out.device)
        _check(lib().sig_fused_voice_bus_fm(*head, bp, bld, C, workspace.data_ptr(), out.data_ptr(), out.stride(0), st, _stream(out)),
               'sig_fused_voice_bus_fm')
        return out
    _audio(out, 'fused rows out')
    rows = out.shape[0]
    if out.dtype != torch.float32 or rows != block_frames * nblocks:
        raise NativeError(f'fused rows out must be float32 ({block_frames * nblocks}, .), got {tuple(out.shape)} {out.dtype}')
    ptrs = []
    for row, name in ((hertz, 'hertz'), (phase, 'phase')):
        if row is not None and row.shape[1] not in (1, voices):
            raise NativeError(f'{name} has {row.shape[1]} channels for {voices} voices')
        ptrs.extend(_ctrl_row(row, name))
    cp, cs, crows = _param_rows(cutoff, 'cutoff', voices, nblocks)
    gp, gs, grows = _param_rows(gain, 'gain', voices, nblocks)
    st = status.data_ptr() if status is not None else None
    pargs = None
    if pair is not None:
        op, kind2, hertz2, phase2, mixrow = pair
        pargs = []
        for row, name in ((hertz2, 'hertz2'), (phase2, 'phase2'), (mixrow, 'mix')):
            if row is not None and row.shape[1] not in (1, voices):
                raise NativeError(f'{name} has {row.shape[1]} channels for {voices} voices')
            pargs.extend(_ctrl_row(row, name))
        head = (OSC_KINDS[kind], OSC_KINDS[kind2], {'Mix': 1, 'RingMod': 2}[op], FILT_TYPES[btype])
    if not bus:
        if out.shape[1] != voices:
            raise NativeError(f'fused rows out has {out.shape[1]} channels for {voices} voices')
        if pargs is not None:
            _check(lib().sig_fused_osc_pair_biquad(*head, rate, position, block_frames, nblocks, context, voices, *ptrs, *pargs,
                                                   cp, cs, crows, gp, gs, grows, out.data_ptr(), out.stride(0), st, _stream(out)),
                   'sig_fused_osc_pair_biquad')
            return out
        _check(lib().sig_fused_osc_biquad_rows(OSC_KINDS[kind], FILT_TYPES[btype], rate, position, block_frames, nblocks, context,
                                               voices, *ptrs, cp, cs, crows, gp, gs, grows, out.data_ptr(), out.stride(0), st,
                                               _stream(out)), 'sig_fused_osc_biquad_rows')
        return out
    C = out.shape[1]
    bp, bld = None, 0
    if bus_gains is not None:
        if bus_gains.dtype != torch.float64 or bus_gains.shape != (C, voices) or bus_gains.stride(1) != 1:
            raise NativeError(f'bus gains must be float64 ({C},{voices}), got {tuple(bus_gains.shape)} {bus_gains.dtype}')
        bp, bld = bus_gains.data_ptr(), bus_gains.stride(0)
    elif C != 1:
        raise NativeError('a bus without gains is mono')
    need = lib().sig_fused_voice_bus_workspace(voices, rows, C)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need // 8, dtype=torch.float64, device=out.device)
    if pargs is not None:
        _check(lib().sig_fused_voice_pair_bus(*head, rate, position, block_frames, nblocks, context, voices, *ptrs, *pargs,
                                              cp, cs, crows, gp, gs, grows, bp, bld, C, workspace.data_ptr(), out.data_ptr(),
                                              out.stride(0), st, _stream(out)), 'sig_fused_voice_pair_bus')
        return out
    _check(lib().sig_fused_voice_bus_rows(OSC_KINDS[kind], FILT_TYPES[btype], rate, position, block_frames, nblocks, context, voices,
                                          *ptrs, cp, cs, crows, gp, gs, grows, bp, bld, C, workspace.data_ptr(), out.data_ptr(),
                                          out.stride(0), st, _stream(out)), 'sig_fused_voice_bus_rows')
    return out


def fused_cascade_bus(kind: str, btype1: str, btype2: str, rate: int, position: int, first_history_start: int,
                      block_frames: int, nblocks: int, context: int, voices: int,
                      hertz: torch.Tensor, phase: torch.Tensor | None, cutoff1: torch.Tensor, cutoff2: torch.Tensor,
                      gain: torch.Tensor | None, envelope: dict | None, bus_gains: torch.Tensor | None, out: torch.Tensor,
                      workspace: torch.Tensor | None = None, status: torch.Tensor | None = None) -> torch.Tensor:
    """out (nblocks*block_frames, C) f32 <- sum over voices of pan * [gain *] [ADSR *] Filter2(Filter1(Osc)), the two
    filters in series with the reference's block-cache history between them (sig_fused_cascade_bus)"""
    _gpu(hertz, phase, cutoff1, cutoff2, gain, bus_gains, out, workspace, status, *(envelope or {}).values())
    _audio(out, 'fused cascade out')
    rows, bus = out.shape
    if out.dtype != torch.float32 or rows != block_frames * nblocks:
        raise NativeError(f'fused cascade out must be float32 ({block_frames * nblocks}, C), got {tuple(out.shape)} {out.dtype}')
    ptrs = []
    for row, name in ((hertz, 'hertz'), (phase, 'phase'), (cutoff1, 'cutoff1'), (cutoff2, 'cutoff2'), (gain, 'gain')):
        if row is not None and row.shape[1] not in (1, voices):
            raise NativeError(f'{name} has {row.shape[1]} channels for {voices} voices')
        ptrs.extend(_ctrl_row(row, name))
    eptrs = estrides = None
    if envelope is not None:
        eptrs = (ctypes.c_void_p * 6)()
        estrides = (ctypes.c_int32 * 6)()
        for i, name in enumerate(ADSR_PARAMS):
            eptrs[i], estrides[i] = _ctrl_row(envelope[name], name)
            if envelope[name].shape[1] not in (1, voices):
                raise NativeError(f'{name} has {envelope[name].shape[1]} channels for {voices} voices')
    gp, gld = None, 0
    if bus_gains is not None:
        if bus_gains.dtype != torch.float64 or bus_gains.shape != (bus, voices) or bus_gains.stride(1) != 1:
            raise NativeError(f'bus gains must be float64 ({bus},{voices}), got {tuple(bus_gains.shape)} {bus_gains.dtype}')
        gp, gld = bus_gains.data_ptr(), bus_gains.stride(0)
    elif bus != 1:
        raise NativeError('a bus without gains is mono')
    need = lib().sig_fused_voice_bus_workspace(voices, rows, bus)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need // 8, dtype=torch.float64, device=out.device)
    _check(lib().sig_fused_cascade_bus(OSC_KINDS[kind], FILT_TYPES[btype1], FILT_TYPES[btype2], rate, position,
                                       first_history_start, block_frames, nblocks, context, voices, *ptrs, eptrs, estrides,
                                       gp, gld, bus, workspace.data_ptr(), out.data_ptr(), out.stride(0),
                                       status.data_ptr() if status is not None else None, _stream(out)),
           'sig_fused_cascade_bus')
    return out


def fused_cascade_geometry(voices: int, nblocks: int) -> tuple[int, int]:
    """(voices per lane, blocks per lane) of `fused_cascade_bus` for this problem size"""
    vpt, span = ctypes.c_int32(), ctypes.c_int32()
    _check(lib().sig_fused_cascade_geometry(voices, nblocks, ctypes.byref(vpt), ctypes.byref(span)), 'sig_fused_cascade_geometry')
    return vpt.value, span.value


def set_fused_cascade_tuning(voices_per_lane: int = 0, blocks_per_lane: int = 0) -> None:
    """tuning / test hook (process-wide): force `fused_cascade_bus`'s launch geometry; the defaults restore the heuristic"""
    _check(lib().sig_fused_cascade_set_tuning(voices_per_lane, blocks_per_lane), 'sig_fused_cascade_set_tuning')


def fused_cascade_model(voices: int, block_frames: int, nblocks: int, context: int = 100, bus_channels: int = 1,
                        osc_ops: float = 5.0, envelope: bool = True) -> dict:
    """f64-rate VALU instructions per stored voice-sample of `fused_cascade_bus` (bench.py's roofline): the exact-phase
    oscillator (5 for a Sawtooth: t = q * hertz + phase, t - 0.5, v_fract_f64, 2 m - 1) and the inner filter (4) on every
    row a lane walks -- span * N output rows and one history block of N + context rows per span -- the outer filter (4)
    on the output rows and the history's last `context` rows, envelope x weight (C) and bus (C) FMAs, the folded flush,
    and the two restarts per (voice, block boundary) (2x2 power by squaring: 10 products of 8 + 6 for `context` = 100)"""
    vpt, span = fused_cascade_geometry(voices, nblocks)
    n, c = block_frames, context
    walked = (span * n + n + c) / (span * n)                    # oscillator + inner filter rows per output row
    products = max(c, 1).bit_length() - 1 + bin(max(c, 1)).count('1')
    restart = 8.0 * products + 6.0
    ops = (osc_ops + 4.0) * walked + 4.0 * (1.0 + c / (span * n)) + (bus_channels if envelope else 0.0) + bus_channels \
        + 17.0 * bus_channels / (16 * vpt) + restart * (2 * (span - 1) + 1) / (span * n)
    return {'f64_ops_per_voice_sample': ops, 'voices_per_lane': vpt, 'blocks_per_lane': span,
            'rows_walked_per_output_row': walked}


def upload_structs(items: list) -> torch.Tensor:
    """a list of ctypes structures of one type as a device byte tensor (a control program, its output table)"""
    arr = (type(items[0]) * len(items))(*items)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(runtime_device())


def runtime_device():
    from . import runtime
    return runtime.device()


def control_program(rate: int, position: int, step: int, nblocks: int, cols: int, program: torch.Tensor, n_ins: int,
                    outs: torch.Tensor, n_outs: int, front_position: int = -1, min_position: int = 0) -> None:
    """run a block-rate control program (sig_control_program): `program` / `outs` are device byte tensors of CtlIns / CtlOut;
    `front_position` >= 0 also evaluates it at that position into the outputs' `front` rows; blocks whose position lies below
    `min_position` are evaluated there"""
    _gpu(program, outs)
    _check(lib().sig_control_program(rate, position, step, nblocks, cols, front_position, min_position, program.data_ptr(), n_ins,
                                     outs.data_ptr(), n_outs, _stream(program)), 'sig_control_program')


def control_program_description(ins: list, outs: list) -> list:
    """what a specialised build of control_program.hip is keyed by: [n_ins, n_outs, (op, kind, a, b, c, dst, wide) per
    instruction, (reg, wide) per output] -- the program's structure without its pointers (CtlIns / CtlOut lists)"""
    words = [len(ins), len(outs)]
    for x in ins:
        words += [x.op, x.kind, x.a, x.b, x.c, x.dst, 1 if x.cols > 1 else 0]
    for o in outs:
        words += [o.reg, 1 if o.cols > 1 else 0]
    return words


def control_program_attach(description: list, image: bytes) -> int:
    """hand the library a build of control_program.hip specialised for this structure; returns the handle to launch it with"""
    arr = (ctypes.c_int32 * len(description))(*description)
    handle = ctypes.c_int32(0)
    _check(lib().sig_control_program_attach(arr, len(description), image, ctypes.byref(handle)), 'sig_control_program_attach')
    return handle.value


def control_program_attached(handle: int, rate: int, position: int, step: int, nblocks: int, cols: int, program: torch.Tensor,
                             n_ins: int, outs: torch.Tensor, n_outs: int, front_position: int = -1, min_position: int = 0) -> None:
    """`control_program` through the specialised kernel behind `handle` (same arguments, same values)"""
    _gpu(program, outs)
    _check(lib().sig_control_program_attached(handle, rate, position, step, nblocks, cols, front_position, min_position,
                                              program.data_ptr(), n_ins, outs.data_ptr(), n_outs, _stream(program)),
           'sig_control_program_attached')


def fused_voice_bus_plan(kind: str, position: int, voices: int, block_frames: int, nblocks: int, context: int) -> dict:
    """what `fused_voice_bus` launches for this problem: {'voices_per_lane', 'blocks_per_lane', 'closed_form', 'kernel'}"""
    vpt, span, closed = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    _check(lib().sig_fused_voice_bus_plan(OSC_KINDS[kind], position, voices, block_frames, nblocks, context,
                                          ctypes.byref(vpt), ctypes.byref(span), ctypes.byref(closed)),
           'sig_fused_voice_bus_plan')
    kernel = f'fused_steady_bus_kernel<{vpt.value}, C>' if closed.value else f'fused_walk_kernel<{kind}, {vpt.value}, gain, C>'
    return {'voices_per_lane': vpt.value, 'blocks_per_lane': span.value, 'closed_form': bool(closed.value), 'kernel': kernel}


def set_fused_tuning(voices_per_lane: int = 0, blocks_per_lane: int = 0, closed_form: int = -1, scan: int = -1) -> None:
    """tuning / test hook (process-wide): force the fused kernels' launch geometry; the defaults restore the heuristics"""
    _check(lib().sig_fused_set_tuning(voices_per_lane, blocks_per_lane, closed_form, scan), 'sig_fused_set_tuning')


def fused_osc_biquad_mix(kind: str, btype: str, rate: int, position: int, block_frames: int, nblocks: int, context: int,
                         hertz: torch.Tensor, phase: torch.Tensor | None, cutoff: torch.Tensor,
                         gain: torch.Tensor | None, matrix: torch.Tensor, out: torch.Tensor,
                         status: torch.Tensor | None = None) -> torch.Tensor:
    """out (nblocks*block_frames, voices) f32 <- ([gain *] Filter(Osc)) @ blockdiag(matrix), 64-voice groups"""
    _gpu(hertz, phase, cutoff, gain, matrix, out, status)
    _audio(out, 'fused mix out')
    rows, voices = out.shape
    if out.dtype != torch.float32 or rows != block_frames * nblocks or voices % 64:
        raise NativeError(f'fused mix out must be float32 ({block_frames * nblocks}, 64*g), got {tuple(out.shape)} {out.dtype}')
    if matrix.dtype != torch.float32 or tuple(matrix.shape) != (64, 64) or not matrix.is_contiguous():
        raise NativeError(f'mix matrix must be contiguous float32 (64, 64), got {tuple(matrix.shape)} {matrix.dtype}')
    ptrs = []
    for row, name in ((hertz, 'hertz'), (phase, 'phase'), (cutoff, 'cutoff'), (gain, 'gain')):
        if row is not None and row.shape[1] not in (1, voices):
            raise NativeError(f'{name} has {row.shape[1]} channels for {voices} voices')
        ptrs.extend(_ctrl_row(row, name))
    _check(lib().sig_fused_osc_biquad_mix(OSC_KINDS[kind], FILT_TYPES[btype], rate, position, block_frames, nblocks, context,
                                          voices, *ptrs, matrix.data_ptr(), out.data_ptr(), out.stride(0),
                                          status.data_ptr() if status is not None else None, _stream(out)),
           'sig_fused_osc_biquad_mix')
    return out


def advance_position(position: torch.Tensor, delta: int) -> None:
    """position[0] += delta on the device, in stream order"""
    _gpu(position)
    _check(lib().sig_advance_position(position.data_ptr(), delta, _stream(position)), 'sig_advance_position')


def fused_voice_bus(kind: str, btype: str, rate: int, position: int, block_frames: int, nblocks: int, context: int,
                    voices: int, hertz: torch.Tensor, phase: torch.Tensor | None, cutoff: torch.Tensor,
                    gain: torch.Tensor | None, bus_gains: torch.Tensor | None, out: torch.Tensor,
                    workspace: torch.Tensor | None = None, status: torch.Tensor | None = None,
                    consts: torch.Tensor | None = None, consts_ready: bool = False, walk: bool = False) -> torch.Tensor:
    """out (nblocks*block_frames, bus_channels) f32 <- sum over voices of pan * [gain *] Filter(Osc).
    `consts`: a float64 device buffer of sig_fused_voice_consts_size(voices) bytes the caller keeps across calls for
    the Sine closed form's per-voice constants; `consts_ready`: it already holds them for these parameters.
    `walk`: the row-by-row span walker only (sig_fused_voice_bus_walk): for launches the caller knows to lie beyond the
    closed form's phase range (SINE_FAST_MAX_CYCLES)."""
    _gpu(hertz, phase, cutoff, gain, bus_gains, out, status, consts)
    _audio(out, 'fused bus out')
    rows, bus = out.shape
    if out.dtype != torch.float32 or rows != block_frames * nblocks:
        raise NativeError(f'fused bus out must be float32 ({block_frames * nblocks}, C), got {tuple(out.shape)} {out.dtype}')
    ptrs = []
    for row, name in ((hertz, 'hertz'), (phase, 'phase'), (cutoff, 'cutoff'), (gain, 'gain')):
        if row is not None and row.shape[1] not in (1, voices):
            raise NativeError(f'{name} has {row.shape[1]} channels for {voices} voices')
        ptrs.extend(_ctrl_row(row, name))
    gp, gld = None, 0
    if bus_gains is not None:
        if bus_gains.dtype != torch.float64 or bus_gains.shape != (bus, voices) or bus_gains.stride(1) != 1:
            raise NativeError(f'bus gains must be float64 ({bus},{voices}), got {tuple(bus_gains.shape)} {bus_gains.dtype}')
        gp, gld = bus_gains.data_ptr(), bus_gains.stride(0)
    need = lib().sig_fused_voice_bus_workspace(voices, rows, bus)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need // 8, dtype=torch.float64, device=out.device)
    if walk:
        _check(lib().sig_fused_voice_bus_walk(OSC_KINDS[kind], FILT_TYPES[btype], rate, position, block_frames, nblocks, context,
                                              voices, *ptrs, gp, gld, bus, workspace.data_ptr(), out.data_ptr(), out.stride(0),
                                              status.data_ptr() if status is not None else None, _stream(out)),
               'sig_fused_voice_bus_walk')
        return out
    if consts is not None:
        if consts.dtype != torch.float64 or consts.numel() * 8 < lib().sig_fused_voice_consts_size(voices):
            raise NativeError('consts must be float64 of sig_fused_voice_consts_size(voices) bytes')
        _check(lib().sig_fused_voice_bus_prepared(OSC_KINDS[kind], FILT_TYPES[btype], rate, position, block_frames, nblocks,
                                                  context, voices, *ptrs, gp, gld, bus, workspace.data_ptr(), out.data_ptr(),
                                                  out.stride(0), status.data_ptr() if status is not None else None,
                                                  _stream(out), consts.data_ptr(), 1 if consts_ready else 0),
               'sig_fused_voice_bus_prepared')
        return out
    _check(lib().sig_fused_voice_bus(OSC_KINDS[kind], FILT_TYPES[btype], rate, position, block_frames, nblocks, context,
                                     voices, *ptrs, gp, gld, bus, workspace.data_ptr(), out.data_ptr(), out.stride(0),
                                     status.data_ptr() if status is not None else None, _stream(out)),
           'sig_fused_voice_bus')
    return out


def band_coldstart(btype: str, rate: int, position: int, block_frames: int, nblocks: int, context: int,
                   low: torch.Tensor, high: torch.Tensor, buf: torch.Tensor, history: int, out: torch.Tensor,
                   status: torch.Tensor | None = None) -> torch.Tensor:
    """BandPass ('bp') / BandStop ('bs'): two biquad sections; buffers as in `biquad_coldstart`."""
    _gpu(low, high, buf, out, status)
    _audio(buf, 'band in')
    _audio(out, 'band out')
    rows, voices = out.shape
    if rows != block_frames * nblocks or buf.shape[0] != history + rows or buf.shape[1] != voices or buf.dtype != out.dtype:
        raise NativeError(f'band shapes: in {tuple(buf.shape)} history {history} out {tuple(out.shape)}')
    ptrs = []
    for row, name in ((low, 'low'), (high, 'high')):
        if row.shape[1] != voices and voices != 1:
            raise IndexError(f'index {row.shape[1]} is out of bounds for axis 1 with size {row.shape[1]}')
        ptrs.extend(_ctrl_row(row, name))
    in_ptr = buf.data_ptr() + history * buf.stride(0) * buf.element_size()
    _check(lib().sig_band_coldstart(FILT_TYPES[btype], rate, position, block_frames, nblocks, context, voices, *ptrs,
                                    in_ptr, buf.stride(0), history, out.data_ptr(), out.stride(0), _dt(out),
                                    status.data_ptr() if status is not None else None, _stream(out)),
           'sig_band_coldstart')
    return out


def adsr_apply(position: int, rate: int, rows: dict, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out = ADSR(rows) * x, float32, one pass"""
    _gpu(x, out, *rows.values())
    _audio(x, 'adsr_apply in')
    _audio(out, 'adsr_apply out')
    if x.dtype != torch.float32 or out.dtype != torch.float32 or x.shape != out.shape:
        raise NativeError(f'adsr_apply is float32 (rows, V) in and out, got {tuple(x.shape)} {x.dtype} -> {tuple(out.shape)} {out.dtype}')
    ptrs = (ctypes.c_void_p * 6)()
    strides = (ctypes.c_int32 * 6)()
    for i, name in enumerate(ADSR_PARAMS):
        ptrs[i], strides[i] = _ctrl_row(rows[name], name)
        if rows[name].shape[1] not in (1, out.shape[1]):
            raise NativeError(f'{name} has {rows[name].shape[1]} channels for {out.shape[1]} voices')
    _check(lib().sig_adsr_apply(position, rate, out.shape[0], out.shape[1], ptrs, strides, x.data_ptr(), x.stride(0),
                                out.data_ptr(), out.stride(0), _stream(out)), 'sig_adsr_apply')
    return out


def _vp_rows(t: torch.Tensor | None, what: str, voices: int, control_rows: int) -> VpRows:
    if t is None:
        return VpRows(None, 0, 1)
    if (t.dtype != torch.float64 or t.dim() != 2 or not t.is_contiguous() or t.shape[1] not in (1, voices)
            or t.shape[0] not in (1, control_rows)):
        raise NativeError(f'{what}: rows of a voice program are contiguous float64 (1|{control_rows}, 1|{voices}), got {tuple(t.shape)} {t.dtype}')
    return VpRows(t.data_ptr(), 0 if t.shape[1] == 1 else 1, t.shape[0])


def voice_program(code: list, oscs: list, params: list, filters: list, n_temps: int, depth: int, rate: int, position: int,
                  block_frames: int, nblocks: int, context: int, voices: int, control_rows: int, hist_positions: list,
                  out: torch.Tensor, bus_gains: torch.Tensor | None = None, bus: bool = False,
                  adsr: dict | None = None, noise_seeds: tuple = (0, 0), workspace: torch.Tensor | None = None,
                  status: torch.Tensor | None = None, blocks_before: int = 0) -> torch.Tensor:
    """One launch for a whole per-voice graph (sig_voice_program).  `code`: (op name, kind, a, b, c) tuples; `oscs`: (hertz,
    phase | None) row tensors per oscillator slot; `params`: row tensors per parameter register; `filters`: (cutoff rows,
    'lp' | 'hp', level = 1 + the filters in series in front of it) per filter slot.  Rows are float64 (1 | control_rows, 1 | voices).  out (nblocks * block_frames, voices) float32,
    or with `bus` (.., C) = the sum over voices weighted by bus_gains."""
    tensors = [t for pair in oscs for t in pair] + list(params) + [f[0] for f in filters] + list((adsr or {}).values())
    _gpu(out, bus_gains, workspace, status, *tensors)
    _audio(out, 'voice program out')
    rows = block_frames * nblocks
    if out.dtype != torch.float32 or out.shape[0] != rows:
        raise NativeError(f'voice program out must be float32 ({rows}, .), got {tuple(out.shape)} {out.dtype}')
    if len(code) > VP_MAX_INS or len(oscs) > VP_MAX_OSCS or len(params) > VP_MAX_PARAMS or len(filters) > VP_MAX_FILTERS \
            or n_temps > VP_MAX_TEMPS or len(hist_positions) > VP_MAX_HIST:
        raise NativeError('voice program larger than the machine')
    P = VoiceProgramT()
    P.n_ins = len(code)
    for k, (op, kind, a, b, c) in enumerate(code):
        P.ins[k] = VpIns(VP_OPS[op], kind, a, b, c)
    P.n_oscs = len(oscs)
    for k, (hz, ph) in enumerate(oscs):
        P.hertz[k] = _vp_rows(hz, 'hertz', voices, control_rows)
        P.phase[k] = _vp_rows(ph, 'phase', voices, control_rows)
    P.n_params = len(params)
    for k, t in enumerate(params):
        P.params[k] = _vp_rows(t, 'parameter', voices, control_rows)
    P.n_filters = len(filters)
    for k, (cut, btype, level) in enumerate(filters):
        if cut.shape[1] != voices and voices != 1:
            raise IndexError(f'index {cut.shape[1]} is out of bounds for axis 1 with size {cut.shape[1]}')      # fx.py:99
        P.cutoff[k] = _vp_rows(cut, 'cutoff', voices, control_rows)
        P.filter_type[k] = FILT_TYPES[btype]
        P.filter_level[k] = level
    P.n_temps, P.depth = n_temps, depth
    if adsr is not None:
        for i, name in enumerate(ADSR_PARAMS):
            P.adsr[i], P.adsr_stride[i] = _ctrl_row(adsr[name], name)
            if adsr[name].shape[1] not in (1, voices):
                raise NativeError(f'{name} has {adsr[name].shape[1]} channels for {voices} voices')
    P.noise_seed[0], P.noise_seed[1] = noise_seeds
    hist = (ctypes.c_int64 * max(1, len(hist_positions)))(*hist_positions)
    C = 0
    gp, gld = None, 0
    if bus:
        C = out.shape[1]
        if bus_gains is not None:
            if bus_gains.dtype != torch.float64 or bus_gains.shape != (C, voices) or bus_gains.stride(1) != 1:
                raise NativeError(f'bus gains must be float64 ({C},{voices}), got {tuple(bus_gains.shape)} {bus_gains.dtype}')
            gp, gld = bus_gains.data_ptr(), bus_gains.stride(0)
        elif C != 1:
            raise NativeError('a bus without gains is mono')
        need = lib().sig_fused_voice_bus_workspace(voices, rows, C)
        if workspace is None or workspace.numel() * workspace.element_size() < need:
            workspace = torch.empty(need // 8, dtype=torch.float64, device=out.device)
    elif out.shape[1] != voices:
        raise NativeError(f'voice program out has {out.shape[1]} channels for {voices} voices')
    _check(lib().sig_voice_program(ctypes.byref(P), rate, position, block_frames, nblocks, context, voices, control_rows,
                                   len(hist_positions), hist, blocks_before, gp, gld, C,
                                   workspace.data_ptr() if workspace is not None else None, out.data_ptr(), out.stride(0),
                                   status.data_ptr() if status is not None else None, _stream(out)), 'sig_voice_program')
    return out


def voice_program_words(code: list) -> list:
    """the machine words of a program given as (op name, kind, a, b, c) tuples: op | kind << 5 | a << 8 | b << 12 | c << 16"""
    return [VP_OPS[op] | (kind << 5) | (a << 8) | (b << 12) | (c << 16) for op, kind, a, b, c in code]


def voice_program_geometry(voices: int, block_frames: int, nblocks: int, context: int, depth: int, bus_channels: int,
                           store_aligned: int, specialised: bool = False) -> tuple:
    """(voices per lane, blocks per lane) sig_voice_program picks for this problem (introspection, no device work);
    store_aligned: 4 | 2 | 1 (see the header); specialised: with a kernel built for four voices per lane at hand"""
    vpt, span = ctypes.c_int32(0), ctypes.c_int32(0)
    _check(lib().sig_voice_program_geometry(voices, block_frames, nblocks, context, depth, bus_channels, int(store_aligned),
                                            1 if specialised else 0, ctypes.byref(vpt), ctypes.byref(span)), 'sig_voice_program_geometry')
    return vpt.value, span.value


def voice_program_attach(code: list, n_oscs: int, n_params: int, n_filters: int, n_temps: int, voices_per_lane: int,
                         bus_channels: int, image: bytes) -> None:
    """hand the library a specialised build of voice_program.hip for exactly this program (signals_amd/specialise.py): later
    sig_voice_program calls with the same program, slot counts, voices per lane and sink launch it instead of the interpreter.
    Loads the image, runs its self-description kernel and synchronises: a set-up call, not a render call."""
    P = VoiceProgramT()
    P.n_ins = len(code)
    for k, (op, kind, a, b, c) in enumerate(code):
        P.ins[k] = VpIns(VP_OPS[op], kind, a, b, c)
    P.n_oscs, P.n_params, P.n_filters, P.n_temps = n_oscs, n_params, n_filters, n_temps
    _check(lib().sig_voice_program_attach(ctypes.byref(P), voices_per_lane, bus_channels, image), 'sig_voice_program_attach')


def voice_program_detach_all() -> None:
    _check(lib().sig_voice_program_detach_all(), 'sig_voice_program_detach_all')


def voice_program_use_attached(on: bool) -> None:
    """test hook (process-wide): launch attached specialised kernels (default) or always the interpreter"""
    _check(lib().sig_voice_program_use_attached(1 if on else 0), 'sig_voice_program_use_attached')


def set_voice_program_tuning(voices_per_lane: int = 0, blocks_per_lane: int = 0) -> None:
    """tuning / test hook (process-wide): force `voice_program`'s launch geometry; the defaults restore the heuristic"""
    _check(lib().sig_voice_program_set_tuning(voices_per_lane, blocks_per_lane), 'sig_voice_program_set_tuning')
