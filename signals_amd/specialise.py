"""Run-time specialisation of the two interpreted kernels: voice programs (signals_amd/csrc/voice_program.hip) and block-rate
control programs (control_program.hip).

The interpreter pays for its generality: the dispatch loop carries the whole register file through a `switch`, and the
register allocator copies ~34 doubles per instruction word back into place.  The SAME source built with the
program as a compile-time constant (`-DSIG_VP_STATIC_CODE={...}`) unrolls that loop and folds every switch: straight-line
HIP for exactly one voice graph, 1.6-2x the interpreter's rate.  Control programs likewise, keyed by their STRUCTURE (ops, register
indices, which instructions are wide; `-DSIG_CTL_STATIC_INS / _OUTS`): their registers become VGPRs instead of an LDS file
behind an interpretive loop, 2.3-5x.  This module builds such an image with the ROCm compiler
(`hipcc --genco`, ~3 s; cross-compiles without a GPU), caches it next to the package keyed by the kernel sources and the
build's parameters, and attaches it to the library (`sig_voice_program_attach`), which from then on launches it whenever
`sig_voice_program` is called with that program.  No hipcc, or a failed build: the interpreter keeps running (it is the
same arithmetic; both are checked against the oracle by tests/test_gpu_specialise.py).

Reference semantics are not touched: the image is voice_program.hip itself (same handlers, same block-sequence machine)."""
import hashlib
import os
import pathlib
import shutil
import subprocess
import threading

from . import _native

CSRC = pathlib.Path(__file__).resolve().parent / 'csrc'
CACHE = pathlib.Path(os.environ.get('SIG_SPECIALISE_CACHE') or pathlib.Path(__file__).resolve().parent / '_specialised')
SOURCES = ('voice_program.hip', 'control_program.hip', 'sig_adsr.h', 'sig_biquad.h', 'sig_bus_tile.h', 'sig_osc.h', 'sig_common.h', '../../include/signals_amd.h')

_attached: set = set()
_failed: set = set()
_lock = threading.Lock()


class SpecialiseError(RuntimeError):
    pass


def hipcc() -> str | None:
    found = os.environ.get('HIPCC') or shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    return found if pathlib.Path(found).exists() else None


def _source_digest() -> str:
    h = hashlib.sha1()
    for name in SOURCES:
        h.update((CSRC / name).read_bytes())
    return h.hexdigest()


def flags(code: list, n_oscs: int, n_params: int, n_filters: int, n_temps: int, voices_per_lane: int, bus_channels: int) -> list:
    """the macros that make voice_program.hip a kernel for exactly this program: the words, a register file of exactly the
    slots it uses, voices per lane, sink, and the waves per SIMD the interpreter's small build asks for"""
    words = ','.join(f'0x{w:x}' for w in _native.voice_program_words(code))
    ext = int(any(_native.VP_OPS[op] >= _native.VP_OPS['Amp'] for op, *_ in code))        # Amp, Adsr, Noise: the extended handlers
    return [f'-DSIG_VP_STATIC_CODE={{{words}}}', f'-DSIG_VP_S_NF={max(n_filters, 1)}', f'-DSIG_VP_S_NO={max(n_oscs, 1)}',
            f'-DSIG_VP_S_NP={max(n_params, 1)}', f'-DSIG_VP_S_NT={n_temps}', f'-DSIG_VP_S_EXT={ext}',
            f'-DSIG_VP_STATIC_VPT={voices_per_lane}', f'-DSIG_VP_STATIC_C={bus_channels}',
            f'-DSIG_VP_STATIC_WAVES={ {1: 3, 2: 2, 4: 1}[voices_per_lane] }']


def _compile(source: str, defs: list, prefix: str) -> bytes:
    """`source` (a file of signals_amd/csrc) built as a gfx950 code object with these macros; cached on disk"""
    key = hashlib.sha1((_source_digest() + source + ' '.join(defs)).encode()).hexdigest()[:24]
    path = CACHE / f'{prefix}_{key}.hsaco'
    if path.exists():
        return path.read_bytes()
    cc = hipcc()
    if cc is None:
        raise SpecialiseError('hipcc not found (set HIPCC): the interpreters keep running')
    try:
        CACHE.mkdir(parents=True, exist_ok=True)
        tmp = path.with_suffix(f'.{os.getpid()}.{threading.get_ident()}.tmp')
        cmd = [cc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-Wno-unused-function', '--genco', *defs,
               '-I', str(CSRC.parent.parent / 'include'), '-o', str(tmp), str(CSRC / source)]
        done = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        if done.returncode != 0 or not tmp.exists():
            tmp.unlink(missing_ok=True)
            raise SpecialiseError(f'hipcc failed on the specialised {source}:\n{done.stderr[-2000:]}')
        os.replace(tmp, path)                             # (atomic: another process may be building the same image)
        return path.read_bytes()
    except (OSError, subprocess.TimeoutExpired) as e:     # a read-only package directory, a compiler that cannot be started ...
        raise SpecialiseError(f'specialised {source} not built: {e}') from e


def build(code: list, n_oscs: int, n_params: int, n_filters: int, n_temps: int, voices_per_lane: int, bus_channels: int) -> bytes:
    """the code object (gfx950) of voice_program.hip specialised for this program; cached on disk"""
    return _compile('voice_program.hip', flags(code, n_oscs, n_params, n_filters, n_temps, voices_per_lane, bus_channels), 'vp')


# ---- block-rate control programs (control_program.hip): the same idea, keyed by the program's STRUCTURE (ops, register indices,
# which instructions and outputs are wide); row and output pointers stay run-time arguments
_ctl_handles: dict = {}


def control_flags(description: list) -> list:
    n_ins, n_outs = description[0], description[1]
    ins = [description[2 + 7 * k: 2 + 7 * k + 7] for k in range(n_ins)]
    outs = [description[2 + 7 * n_ins + 2 * k: 2 + 7 * n_ins + 2 * k + 2] for k in range(n_outs)]
    braces = lambda rows: '{' + ','.join('{' + ','.join(str(int(w)) for w in row) + '}' for row in rows) + '}'
    return [f'-DSIG_CTL_STATIC_INS={braces(ins)}', f'-DSIG_CTL_STATIC_OUTS={braces(outs)}']


def ensure_control(description: list, background: bool = False):
    """the handle of the kernel specialised for this control-program structure (built and attached once per process), or None
    while it is not available (no hipcc, a failed build, or -- `background` -- still being built)"""
    key = tuple(description)
    with _lock:
        if key in _ctl_handles:
            return _ctl_handles[key]
        if key in _failed or key in _pending:
            return None
    if description[0] == 0 or description[1] == 0:
        return None

    def make():
        try:
            image = _compile('control_program.hip', control_flags(description), 'ctl')
            handle = _native.control_program_attach(list(description), image)
        except (SpecialiseError, _native.NativeError) as e:
            with _lock:
                _failed.add(key)
            import warnings
            warnings.warn(f'control program not specialised, the interpreter runs it: {e}')
            return None
        with _lock:
            _ctl_handles[key] = handle
        return handle
    if not background:
        return make()
    global _pool
    with _lock:
        if _pool is None:
            import concurrent.futures
            _pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix='sig-specialise')

        def job():
            try:
                return make()
            finally:
                with _lock:
                    _pending.pop(key, None)
        _pending[key] = _pool.submit(job)
    return None


def ensure(code: list, n_oscs: int, n_params: int, n_filters: int, n_temps: int, voices_per_lane: int, bus_channels: int) -> bool:
    """build (or load from the cache) and attach the specialised kernel for this program once per process; False when that is
    not possible here -- the caller's launches then run the interpreter"""
    key = (tuple(code), n_oscs, n_params, n_filters, n_temps, voices_per_lane, bus_channels)
    with _lock:
        if key in _attached:
            return True
        if key in _failed:
            return False
    try:                                                  # (outside the lock: the compiler takes seconds)
        image = build(code, n_oscs, n_params, n_filters, n_temps, voices_per_lane, bus_channels)
        with _lock:
            if key not in _attached:
                _native.voice_program_attach(code, n_oscs, n_params, n_filters, n_temps, voices_per_lane, bus_channels, image)
                _attached.add(key)
    except (SpecialiseError, _native.NativeError) as e:
        with _lock:
            _failed.add(key)
        import warnings
        warnings.warn(f'voice program not specialised, the interpreter runs it: {e}')
        return False
    return True


_pending: dict = {}
_pool = None


def ensure_in_background(code: list, n_oscs: int, n_params: int, n_filters: int, n_temps: int, voices_per_lane: int,
                         bus_channels: int) -> bool:
    """`ensure` on a worker thread: True once the kernel is attached, False while it is being built (or cannot be) -- the
    caller's launches run the interpreter until then, so a real-time sink never waits for the compiler"""
    global _pool
    key = (tuple(code), n_oscs, n_params, n_filters, n_temps, voices_per_lane, bus_channels)
    with _lock:
        if key in _attached:
            return True
        if key in _failed or key in _pending:
            return False
        if _pool is None:
            import concurrent.futures
            _pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix='sig-specialise')
        _pending[key] = _pool.submit(_finish, key)
        return False


def _finish(key) -> bool:
    try:
        return ensure(list(key[0]), *key[1:])
    finally:
        with _lock:
            _pending.pop(key, None)


def wait() -> None:
    """block until every background build has finished (tests, benchmarks)"""
    while True:
        with _lock:
            futures = list(_pending.values())
        if not futures:
            return
        for f in futures:
            f.result()


def forget() -> None:
    """detach every specialised kernel (tests)"""
    wait()
    with _lock:
        _native.voice_program_detach_all()
        _attached.clear()
        _failed.clear()
