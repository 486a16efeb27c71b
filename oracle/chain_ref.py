"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy/scipy restatement of the reference's per-block node-graph evaluation
(noah-aviel-dove/signals, `src/signals/chain/{__init__,osc,fx,fixed,shape}.py`).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; nothing under `signals_amd/` does.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function
here against `tests/golden/*.npz`, which `tests/golden/gen_golden.py` produced by
importing and running the reference itself in the build container (numpy 2.2.6,
scipy 1.15.3 -- the reference pins numpy 1.23.0 / scipy 1.10.1 in
`requirements.txt:5-6`; the filter arithmetic lives in scipy.signal.butter /
sosfilt, which this file both calls, like the reference does, and restates
in closed form / as a DF2T loop so the HIP kernels have a formula to follow).

All arrays are float64, C-order `(frames, channels)`, exactly like the reference.
File:line citations are relative to /root/reference/src/signals/.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.signal

CONTEXT_FRAMES = 100          # fx.py:82-83
MAX_CACHED_BLOCKS = 16        # chain/__init__.py:429


# --------------------------------------------------------------------------- A1
def frame_range(position: int, frames: int) -> np.ndarray:
    """chain/__init__.py:121-125 -- int64 column arange(position, position+frames)."""
    return np.arange(position, position + frames, dtype=np.int64).reshape(-1, 1)


def before(position: int, frames: int, ctx: int) -> tuple[int, int]:
    """chain/__init__.py:149-153 -> (position, frames) of the 'before' window."""
    return max(position - ctx, 0), min(ctx, position)


def after(position: int, frames: int, ctx: int) -> tuple[int, int]:
    """chain/__init__.py:155-159."""
    return position + frames, ctx


def shape_le(a: tuple[int, int], b: tuple[int, int]) -> bool:
    """chain/__init__.py:59-60 -- broadcast-compatible reply check."""
    return (a[0] in (1, b[0])) and (a[1] in (1, b[1]))


# --------------------------------------------------------------------------- A6
def osc_cycles(position: int, frames: int, rate: int, hertz: np.ndarray, phase: np.ndarray) -> np.ndarray:
    """osc.py:32 -- `frame_range / rate * hertz + phase`, evaluated left to right in float64."""
    return frame_range(position, frames) / rate * hertz + phase


def osc_wave(kind: str, t: np.ndarray) -> np.ndarray:
    if kind == 'Sine':            # osc.py:42-43
        return np.sin(t * 2 * np.pi)
    if kind == 'Square':          # osc.py:48-49
        return np.sign(0.5 - np.mod(t, 1))
    if kind == 'Sawtooth':        # osc.py:54-55
        return 2 * np.mod(t - 0.5, 1) - 1
    if kind == 'Triangle':        # osc.py:60-62
        t = t - 0.25
        return (4 * np.mod(t, 0.5) - 1) * np.sign(np.mod(t, 1) - 0.5)
    raise ValueError(kind)


def osc(kind: str, position: int, frames: int, rate: int, hertz, phase=None) -> np.ndarray:
    """osc.py:26-33.  `phase=None` = unplugged port -> zeros((1,1)) (chain/__init__.py:297-298)."""
    hertz = np.asarray(hertz)
    phase = np.zeros((1, 1)) if phase is None else np.asarray(phase)
    return osc_wave(kind, osc_cycles(position, frames, rate, hertz, phase))


# --------------------------------------------------------------------------- A7
def butter2_sos(wn: float, btype: str) -> np.ndarray:
    """Closed form of scipy.signal.butter(2, wn, btype, output='sos') for btype in {'lp','hp'}.

    scipy's chain (_filter_design.py: buttap -> lp2lp_zpk/lp2hp_zpk with warped
    frequency 2*fs*tan(pi*wn/fs), fs=2 -> bilinear_zpk -> zpk2sos) collapses, for one
    conjugate pole pair, to k = tan(pi*wn/2):
        nrm = 1/(1 + sqrt2 k + k^2)
        lp: b = (k^2, 2k^2, k^2) nrm      hp: b = (1, -2, 1) nrm
        a = (1, 2(k^2-1) nrm, (1 - sqrt2 k + k^2) nrm)
    Agreement with scipy is ~1e-15 (tests), not bit-exact: scipy goes through complex
    pole arithmetic.  Raises like scipy for wn outside (0,1) (fx.py:99-102).
    """
    if not (0.0 < wn < 1.0):
        raise ValueError('Digital filter critical frequencies must be 0 < Wn < 1')
    k = math.tan(math.pi * wn / 2.0)
    k2 = k * k
    nrm = 1.0 / (1.0 + math.sqrt(2.0) * k + k2)
    if btype == 'lp':
        b = (k2 * nrm, 2.0 * k2 * nrm, k2 * nrm)
    elif btype == 'hp':
        b = (nrm, -2.0 * nrm, nrm)
    else:
        raise ValueError(btype)
    return np.array([[b[0], b[1], b[2], 1.0, 2.0 * (k2 - 1.0) * nrm, (1.0 - math.sqrt(2.0) * k + k2) * nrm]])


def sosfilt_df2t(sos: np.ndarray, x: np.ndarray) -> np.ndarray:
    """Restatement of scipy.signal.sosfilt (compiled `_sosfilt`): transposed direct form II,
    zero initial state, one rounding per multiply/add in this order.  Bit-identical to scipy
    (tests).  Pure-Python loop: small inputs only."""
    y = np.array(x, dtype=np.float64)
    for s in range(sos.shape[0]):
        b0, b1, b2, _, a1, a2 = (float(v) for v in sos[s])
        z0 = z1 = 0.0
        for n in range(y.shape[0]):
            xn = float(y[n])
            yn = b0 * xn + z0
            z0 = b1 * xn - a1 * yn + z1
            z1 = b2 * xn - a2 * yn
            y[n] = yn
        # next section filters this section's output
    return y


def crit_filter(btype: str, window: np.ndarray, cutoff: np.ndarray, rate: int, frames: int,
                ctx: int = CONTEXT_FRAMES, *, closed_form: bool = False, loop: bool = False) -> np.ndarray:
    """fx.py:85-106.  `window` is the concatenated [before | block | after] input; per channel
    design one biquad, filter the whole window from zero state, keep `[-(frames+ctx):-ctx]`.
    `closed_form`/`loop` swap scipy's butter/sosfilt for the restatements above."""
    channels = window.shape[1]
    result = np.empty((frames, channels))
    for i in range(channels):
        wn = np.array([cutoff[0, i]], dtype=float)      # IndexError if cutoff is narrower (fx.py:99)
        wn /= rate / 2
        wn.clip(0, 1, out=wn)
        if closed_form:
            sos = butter2_sos(float(wn[0]), btype)
        else:
            sos = scipy.signal.butter(N=2, Wn=wn, btype=btype, output='sos')
        col = window[:, i]
        y = sosfilt_df2t(sos, col) if loop else scipy.signal.sosfilt(sos, col, axis=0)
        result[:, i] = y[-(frames + ctx):-ctx]
    return result


def band2_sos(wn_lo: float, wn_hi: float, btype: str) -> np.ndarray:
    """Closed form of scipy.signal.butter(2, [wn_lo, wn_hi], 'bp'|'bs', output='sos') -- two sections.
    Restates buttap -> lp2bp_zpk / lp2bs_zpk -> bilinear_zpk (fs = 2) -> zpk2sos ('nearest' pairing: the
    pole pair closest to the unit circle goes last and takes the nearest zeros; gain on the first section).
    This is the arithmetic signals_amd/csrc/sig_biquad.h:design_band2 runs on the GPU."""
    if not (0.0 < wn_lo < 1.0 and 0.0 < wn_hi < 1.0):
        raise ValueError('Digital filter critical frequencies must be 0 < Wn < 1')
    if wn_lo >= wn_hi:
        raise ValueError('Wn[0] must be less than Wn[1]')

    def csqrt(z: complex) -> complex:
        m = math.hypot(z.real, z.imag)
        if m == 0:
            return 0j
        if z.real >= 0:
            t = math.sqrt((m + z.real) / 2)
            return complex(t, z.imag / (2 * t))
        t = math.sqrt((m - z.real) / 2)
        return complex(abs(z.imag) / (2 * t), math.copysign(t, z.imag))

    w1, w2 = 4 * math.tan(math.pi * wn_lo / 2), 4 * math.tan(math.pi * wn_hi / 2)
    bw, wo2 = w2 - w1, w1 * w2
    p = complex(-math.sqrt(0.5), math.sqrt(0.5))
    c = p * bw / 2 if btype == 'bp' else (bw / 2) / p
    s = csqrt(c * c - wo2)
    qa, qb = c + s, c - s
    poles = []
    for q in (qa, qb):
        P = (4 + q) / (4 - q)
        poles.append(P if P.imag >= 0 else P.conjugate())
    den = abs(4 - qa) ** 2 * abs(4 - qb) ** 2
    a_worst = abs(1 - abs(poles[0])) <= abs(1 - abs(poles[1]))
    worst, other = (poles[0], poles[1]) if a_worst else (poles[1], poles[0])
    if btype == 'bp':
        kz = bw * bw * 16.0 / den
        z1 = 1.0 if abs(worst - 1) <= abs(worst + 1) else -1.0
        b_last, b_first = [1.0, -2 * z1, 1.0], [1.0, 2 * z1, 1.0]
    elif btype == 'bs':
        z0 = complex(4, math.sqrt(wo2)) / complex(4, -math.sqrt(wo2))
        kz = (16.0 + wo2) ** 2 / den
        b_last = b_first = [1.0, -2 * z0.real, 1.0]
    else:
        raise ValueError(btype)

    def a_of(P):
        return [1.0, -2 * P.real, P.real ** 2 + P.imag ** 2]
    return np.array([[kz * v for v in b_first] + a_of(other), b_last + a_of(worst)])


def band_filter(btype: str, window: np.ndarray, low: np.ndarray, high: np.ndarray, rate: int, frames: int,
                ctx: int = CONTEXT_FRAMES, *, closed_form: bool = False) -> np.ndarray:
    """BandPass/BandStop as the reference INTENDS them (fx.py:85-106 with crit_2; the shipped code raises
    TypeError at :99): per channel butter(N=2, Wn=[low, high]/(rate/2), btype) -> 2 sections, sosfilt over the
    window from zero state, keep `[-(frames+ctx):-ctx]`.  Parity unpinned by the reference; pinned to scipy."""
    result = np.empty((frames, window.shape[1]))
    for i in range(window.shape[1]):
        wn = np.array([low[0, i], high[0, i]], dtype=float)
        wn /= rate / 2
        wn.clip(0, 1, out=wn)
        sos = band2_sos(float(wn[0]), float(wn[1]), btype) if closed_form else \
            scipy.signal.butter(N=2, Wn=wn, btype=btype, output='sos')
        result[:, i] = scipy.signal.sosfilt(sos, window[:, i], axis=0)[-(frames + ctx):-ctx]
    return result


def filter_block(btype: str, source, position: int, frames: int, rate: int, cutoff: np.ndarray,
                 ctx: int = CONTEXT_FRAMES, **kw) -> np.ndarray:
    """Single filter over a position-pure `source(position, frames) -> array`
    (chain/__init__.py:308-315 forward_with_context + fx.py:93-105)."""
    blocks = []
    if position > 0:
        bp, bf = before(position, frames, ctx)
        blocks.append(source(bp, bf))
    blocks.append(source(position, frames))
    blocks.append(source(*after(position, frames, ctx)))
    return crit_filter(btype, np.concatenate(blocks), cutoff, rate, frames, ctx, **kw)


# --------------------------------------------------------------------------- A8
def gain(left, right_block_rate):           # fx.py:51-52
    return left * right_block_rate


def mix(left, right, mix_block_rate):       # fx.py:38-40
    return mix_block_rate * left + (1 - mix_block_rate) * right


def ringmod(left, right):                   # fx.py:45-46
    return left * right


def amp(left, exp_block_rate):              # fx.py:57-60
    with np.errstate(invalid='ignore'):
        return np.copysign(left ** exp_block_rate, left)


def merge(left, right):                     # shape.py:73-74
    return np.hstack((left, right))


# --------------------------------------------------------------------------- A11 (build-defined)
def sum_bus(x: np.ndarray, gains: np.ndarray | None = None) -> np.ndarray:
    """Build-defined (the reference's Flatten sums over frames and crashes, shape.py:35).
    gains None -> (N,1) mono sum over voices; gains (C,V) -> (N,C) = x @ gains.T.
    Parity unpinned by the reference; this restatement is the definition."""
    if gains is None:
        return np.sum(x, axis=1, keepdims=True)
    return x @ np.asarray(gains).T


def adsr(position: int, frames: int, rate: int, attack, decay, sustain, release, gate_on, gate_off) -> np.ndarray:
    """Build-defined position-pure piecewise-linear envelope, per voice, frame rate.
    All parameters are (1,V) rows; times in seconds from position 0, sustain a level.
        t = n / rate;  u = t - gate_on;  v = u - attack;  w = t - gate_off
        held(t)  = 0 (u < 0) | clip(u * (1/attack), 0, 1) (v < 0) | 1 + (sustain - 1) * clip(v * (1/decay), 0, 1)
                   (a zero-length stage counts as complete)
        level(t) = held(t) (w < 0) | held(gate_off) * clip(1 - w * (1/release), 0, 1)   (release == 0: 0)
    Parity unpinned by the reference (only a dead sketch exists, sig.py:89-100); this restatement is the
    definition and signals_amd/csrc/adsr.hip follows it operation for operation."""
    attack, decay, sustain, release, gate_on, gate_off = (
        np.asarray(a, dtype=np.float64) for a in (attack, decay, sustain, release, gate_on, gate_off))
    t = frame_range(position, frames) / rate
    with np.errstate(divide='ignore'):
        ia = np.where(attack > 0, 1.0 / np.where(attack > 0, attack, 1.0), 0.0)
        id_ = np.where(decay > 0, 1.0 / np.where(decay > 0, decay, 1.0), 0.0)
        ir = np.where(release > 0, 1.0 / np.where(release > 0, release, 1.0), 0.0)

    def held(tt):
        u = tt - gate_on
        v = u - attack
        a = np.where(ia > 0, np.clip(u * ia, 0.0, 1.0), 1.0)
        d = np.where(id_ > 0, np.clip(v * id_, 0.0, 1.0), 1.0)
        return np.where(u < 0, 0.0, np.where(v < 0, a, 1.0 + (sustain - 1.0) * d))

    hold_off = held(gate_off + np.zeros((1, 1)))
    w = t - gate_off
    rel = np.where(ir > 0, np.clip(1.0 - w * ir, 0.0, 1.0), 0.0)
    return np.where(w < 0, held(t), hold_off * rel)


def mix_matrix(x: np.ndarray, m: np.ndarray) -> np.ndarray:
    """Build-defined: out[n, 64g:64g+64] = x[n, 64g:64g+64] @ M, M (64,64).  Parity unpinned."""
    n, v = x.shape
    g = m.shape[0]
    return (x.reshape(n, v // g, g) @ m).reshape(n, v)


# --------------------------------------------------------------------------- A3/A4/A9: the pull protocol
class Node:
    """Minimal restatement of Emitter/Receiver/BlockCachingEmitter semantics
    (chain/__init__.py:212-263, :266-364, :424-457) for graph-level oracles: a disabled or
    unplugged input answers zeros((1,1)); cached emitters answer exact hits or the first cached
    block that CONTAINS the request, sliced; FIFO of 16."""
    cached = True

    def __init__(self, **inputs):
        self.inputs = inputs
        self.enabled = True
        self._cache: dict[tuple[int, int, int, int], np.ndarray] = {}

    # -- port helpers (BoundPort.request / forward_at_block_rate / forward_with_context)
    def _req(self, port, position, frames, channels, rate):
        src = self.inputs.get(port)
        if src is None:
            return np.zeros((1, 1))
        block = src.respond(position, frames, channels, rate)
        if not shape_le(block.shape, (frames, channels)):
            raise ValueError(f'BadShape {block.shape} vs {(frames, channels)}')
        return block

    def _ctrl(self, port, position, channels, rate):
        return self._req(port, position, 1, channels, rate)

    def _with_context(self, port, position, frames, channels, rate, ctx):
        blocks = []
        if position > 0:
            bp, bf = before(position, frames, ctx)
            blocks.append(self._req(port, bp, bf, channels, rate))
        blocks.append(self._req(port, position, frames, channels, rate))
        ap, af = after(position, frames, ctx)
        blocks.append(self._req(port, ap, af, channels, rate))
        return np.concatenate(blocks)

    # -- emitter side
    def eval(self, position, frames, channels, rate) -> np.ndarray:
        raise NotImplementedError

    def respond(self, position, frames, channels, rate) -> np.ndarray:
        if not self.cached:
            return self.eval(position, frames, channels, rate) if self.enabled else np.zeros((1, 1))
        key = (position, frames, channels, rate)
        if key in self._cache:
            return self._cache[key]
        for (p, f, c, r), block in self._cache.items():          # chain/__init__.py:435-442
            if r == rate and position >= p and position + frames <= p + f and channels <= c:
                start = position - p
                return block[start:start + frames, :channels]
        result = self.eval(position, frames, channels, rate) if self.enabled else np.zeros((1, 1))
        self._cache[(position, result.shape[0], result.shape[1], rate)] = result   # :445-447
        if len(self._cache) > MAX_CACHED_BLOCKS:
            self._cache.pop(next(iter(self._cache)))
        return result


class Fixed(Node):
    cached = False                                               # fixed.py:21 is a plain Emitter

    def __init__(self, value):
        super().__init__()
        self.value = np.array(value, ndmin=2)

    def eval(self, position, frames, channels, rate):
        return self.value


class Osc(Node):
    def __init__(self, kind, hertz=None, phase=None):
        super().__init__(hertz=hertz, phase=phase)
        self.kind = kind

    def eval(self, position, frames, channels, rate):
        phase = self._ctrl('phase', position, channels, rate)
        hertz = self._ctrl('hertz', position, channels, rate)
        return osc_wave(self.kind, osc_cycles(position, frames, rate, hertz, phase))


class Filter(Node):
    def __init__(self, btype, input=None, cutoff=None, **kw):
        super().__init__(input=input, cutoff=cutoff)
        self.btype = btype
        self.kw = kw

    def eval(self, position, frames, channels, rate):
        cutoff = self._ctrl('cutoff', position, channels, rate)
        window = self._with_context('input', position, frames, channels, rate, CONTEXT_FRAMES)
        if window.shape[1] != channels:
            raise IndexError('filter input narrower than the request (fx.py:98-105)')
        return crit_filter(self.btype, window, cutoff, rate, frames, **self.kw)


class BandFilter(Node):
    def __init__(self, btype, input=None, low=None, high=None):
        super().__init__(input=input, low=low, high=high)
        self.btype = btype

    def eval(self, position, frames, channels, rate):
        low = self._ctrl('low', position, channels, rate)
        high = self._ctrl('high', position, channels, rate)
        window = self._with_context('input', position, frames, channels, rate, CONTEXT_FRAMES)
        return band_filter(self.btype, window, low, high, rate, frames)


class Binary(Node):
    def __init__(self, op, left=None, right=None, mix=None):
        super().__init__(left=left, right=right, mix=mix)
        self.op = op

    def eval(self, position, frames, channels, rate):
        if self.op == 'Gain':
            return gain(self._req('left', position, frames, channels, rate),
                        self._ctrl('right', position, channels, rate))
        if self.op == 'Mix':
            m = self._ctrl('mix', position, channels, rate)
            return mix(self._req('left', position, frames, channels, rate),
                       self._req('right', position, frames, channels, rate), m)
        if self.op == 'RingMod':
            return ringmod(self._req('left', position, frames, channels, rate),
                           self._req('right', position, frames, channels, rate))
        if self.op == 'Amp':
            return amp(self._req('left', position, frames, channels, rate),
                       self._ctrl('right', position, channels, rate))
        raise ValueError(self.op)


class Merge(Node):
    def __init__(self, left, right, left_channels, right_channels):
        super().__init__(left=left, right=right)
        self.lc, self.rc = left_channels, right_channels

    def eval(self, position, frames, channels, rate):
        return merge(self._req('left', position, frames, self.lc, rate),
                     self._req('right', position, frames, self.rc, rate))


class SumBus(Node):
    def __init__(self, input, gains=None):
        super().__init__(input=input)
        self.gains = gains
        self.in_channels = None

    def eval(self, position, frames, channels, rate):
        v = self.in_channels or (np.asarray(self.gains).shape[1] if self.gains is not None else channels)
        return sum_bus(self._req('input', position, frames, v, rate), self.gains)


class Adsr(Node):
    def __init__(self, **rows):
        super().__init__()
        self.rows = {k: np.array(v, ndmin=2, dtype=float) for k, v in rows.items()}

    def eval(self, position, frames, channels, rate):
        return adsr(position, frames, rate, **self.rows)


class MixMatrix(Node):
    def __init__(self, input, matrix):
        super().__init__(input=input)
        self.matrix = np.asarray(matrix, dtype=float)

    def eval(self, position, frames, channels, rate):
        return mix_matrix(self._req('input', position, frames, channels, rate), self.matrix)


def render(node: Node, position: int, frames: int, channels: int, rate: int = 48000) -> np.ndarray:
    return np.array(node.respond(position, frames, channels, rate), dtype=np.float64)


def render_stream(node: Node, position: int, frames: int, blocks: int, channels: int, rate: int = 48000) -> np.ndarray:
    """Sequential render like dev.py:167-179 (position += frames each callback)."""
    return np.concatenate([render(node, position + b * frames, frames, channels, rate) for b in range(blocks)])
