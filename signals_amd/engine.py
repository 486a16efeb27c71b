"""Batched block renderer: K consecutive blocks of a node graph per kernel launch.

The eager pull path (`signals_amd.chain`) mirrors the reference one request at a time; at 1 MiB per
1024x256 buffer that is launch-bound.  Because oscillators are closed-form in position
and every filter block cold-starts from zero state <=100 frames early, the K
blocks of a stream are independent given each filter input's history rows, so one launch per node can
cover all of them.  `BatchRenderer.render` returns exactly what K sequential `request()` calls
through the eager path return (tests/test_gpu_engine.py), including the cache-history semantics of
cascaded filters:

  * a filter's context rows for block b>0 are its input's rows of block b-1 (what the reference's
    block cache serves by slicing the previous block, chain/__init__.py:431-442);
  * for block 0 of a continuing stream they are the input's saved tail of the previous batch;
  * on a fresh start at position p>0 they are the input rendered as its own block [p-c, p), which
    cold-starts an upstream filter at p-c-100 -- the reference's behaviour on a fresh graph.

The reference's `after(100)` requests never reach kept samples (sosfilt is causal) and, for constant
block size N > 100, never serve a later request from the cache; they are skipped here.

Control inputs (forward_at_block_rate) that are not constant -- an LFO on a cutoff, block-rate FM -- are
evaluated for all K blocks in block-rate launches and handed to their consumers as K parameter rows; a
node with such an input answers differently depending on which request produced a frame range (the
control value is read at the REQUEST's position), so, like a filter, its history rows come from the
previous batch's tail or from a fresh block, never from re-rendering.

With `fuse=True` (default) `[SumBus(] [Gain(] LowPass|HighPass(Osc) [)] [)]` runs as one fused launch
when nothing else consumes the intermediates; a graph that is exactly one such launch is replayed per
call without re-walking it (latency mode).  `fuse=False` is one kernel per node and bit-identical to the
eager path.

Node classes without a kernel schedule (plugins, including ones written against the reference that
answer numpy arrays) are pulled block by block through their own `respond()` and laid out as a batch
buffer; everything downstream of them still runs one launch per node.

Graphs that do not fit (a filter inside a control path, per-block ADSR / band-filter parameters,
cascaded filters with N <= 100) raise `NotBatchable`; callers fall back to the eager path
(`BlockDriver` does so by itself).
"""
from __future__ import annotations

import os
import typing

import torch

from signals_amd import _native, runtime
from signals_amd.chain import (
    AUDIO_DTYPE,
    CTRL_DTYPE,
    Emitter,
    Receiver,
    as_control,
    broadcast_shape,
    graph_clock,
    port,
)
from signals_amd.chain import ext, files, fixed, fx, noise, osc, shape

CONTEXT = 100
SCAN_MAX_CHAINS = 16384      # signals_amd/csrc/fused_voice.hip: kScanMaxChains
SCAN_MAX_ROWS = 512          # kScanMaxL * 64


class NotBatchable(Exception):
    """This graph needs the eager pull path."""


class KernelTimer:
    """Optional per-kernel HIP-event timing on the launch stream (bench.py's roofline leg)."""

    def __init__(self, sample_every: int = 1, region: bool = False):
        """`sample_every` = n: only every n-th launch of a kernel is bracketed by events (an event pair between two
        launches drains the queue: bracketing every 200-us launch cost the stream 12 %); the summary's calls, time and
        units then count the bracketed launches only, so its averages stay per launch.
        `region`: ONE event in front of the first launch and one behind the last (`close()`): nothing between the launches, so
        the stream runs exactly as it does untimed; the elapsed time is shared out evenly over the launches (for schedules
        that are one launch per step -- the average then includes the gap between two launches, as the wall clock does)"""
        self.records: list[tuple[str, torch.cuda.Event, torch.cuda.Event, dict]] = []
        self.sample_every = max(1, int(sample_every))
        self.region = region
        self._seen: dict[str, int] = {}
        self._units: dict[str, int] = {}
        self._start = self._stop = None

    def launch(self, name: str, fn: typing.Callable, **meta):
        n = self._seen.get(name, 0)
        self._seen[name] = n + 1
        if self.region:
            self._units[name] = self._units.get(name, 0) + meta.get('units', 0)
            if self._start is None:
                self._start = torch.cuda.Event(enable_timing=True)
                self._start.record()
            return fn()
        if n % self.sample_every:
            return fn()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        out = fn()
        stop.record()
        self.records.append((name, start, stop, meta))
        return out

    def close(self) -> None:
        """region mode: the event behind the last launch"""
        if self.region and self._start is not None and self._stop is None:
            self._stop = torch.cuda.Event(enable_timing=True)
            self._stop.record()

    def summary(self) -> dict[str, dict]:
        """call after a device sync"""
        acc: dict[str, dict] = {}
        if self.region:
            self.close()
            if self._start is None:
                return acc
            total, launches = self._start.elapsed_time(self._stop), sum(self._seen.values())
            for name, calls in self._seen.items():
                acc[name] = {'calls': calls, 'ms': total * calls / launches, 'units': self._units.get(name, 0)}
            return acc
        for name, start, stop, meta in self.records:
            e = acc.setdefault(name, {'calls': 0, 'ms': 0.0, 'units': 0})
            e['calls'] += 1
            e['ms'] += start.elapsed_time(stop)
            e['units'] += meta.get('units', 0)
        return acc

    def reset(self):
        self.records.clear()
        self._seen.clear()
        self._units.clear()
        self._start = self._stop = None


class _CapturedLaunches:
    """A short launch sequence captured once into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed per
    call: the frame position lives in a device int64 that the kernels read and the graph's last node
    advances, so consecutive blocks replay with no host work beyond one graph launch.  The output buffer is
    owned by the graph and overwritten by every replay."""

    def __init__(self, record: typing.Callable[[torch.Tensor], torch.Tensor], advance: int, position: int, keys: tuple,
                 self_advancing: bool = False):
        """`self_advancing`: the recorded launches move the device position themselves (sig_latency_voice_bus)"""
        dev = runtime.device()
        self.keys = keys                                    # identities of the tensors baked into the graph
        self.advance = advance
        self.pos = torch.tensor([position], dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up outside capture, as graph capture requires
            record(self.pos)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.pos.fill_(position)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = record(self.pos)
            if not self_advancing:
                _native.advance_position(self.pos, advance)
        self.pos.fill_(position)
        self.next_position = position

    def run(self, position: int) -> torch.Tensor:
        if position != self.next_position:
            self.pos.fill_(position)                        # seek
        self.graph.replay()
        self.next_position = position + self.advance
        return self.out


class _EngineSink(Receiver):
    """stands in for the consumer in requests the engine issues itself (plugin nodes pulled block by block)"""
    input = port('input')
    HOST_ARRAYS = False

    @classmethod
    def flags(cls):
        from signals_amd import SignalFlags
        return SignalFlags(0)


class BatchRenderer:
    """Renders `node` (as seen through a request of `channels` channels at `rate`) in batches of consecutive
    blocks.  Keeps what a stream needs between batches: the last <=100 rows of every request-dependent
    node (tails), the device status words of its filters, and the replay closure of a one-launch plan."""

    def __init__(self, node: Emitter, channels: int, rate: int = 48000, timer: KernelTimer | None = None,
                 fuse: bool = True, fuse_bus: bool = True, graph_replay: bool = False, fuse_program: bool | None = None,
                 specialise: bool | str | None = None, pipeline: int = 1):
        """`fuse`: let Filter(Osc) [and a Gain on top] run as one kernel when the intermediate outputs have
        no other consumer (sig_fused_osc_biquad); `fuse_bus`: also fold a SumBus on top into that launch
        (sig_fused_voice_bus).  fuse=False = one kernel per node, bit-identical to the eager path.
        `fuse_program` (default: as `fuse`): graphs none of the fused kernels covers run as one interpreted launch per sink
        (sig_voice_program) instead of one kernel per node -- where the interpreter beats that schedule (programs that fit its
        small register file, and every block size below the filter context, which the per-node schedule cannot batch at all);
        'always': wherever the graph compiles.
        `specialise`: build the voice-program kernel once more for exactly this graph's program (signals_amd/specialise.py:
        hipcc, a few seconds at the first render, cached on disk) and launch that instead of the interpreter -- the same
        arithmetic as straight-line code, 1.6-2x its rate; without hipcc the interpreter keeps running.  'background': the build
        runs on a worker thread and the interpreter renders until the kernel is attached (a real-time sink never waits for
        the compiler).  Default: off, or the environment's SIG_SPECIALISE=1.
        `pipeline` (2 or 3; default 1 = off): consecutive batches of a graph that is ONE fused launch without history (the C2
        voice graph) go to alternating HIP streams, each with its own workspace, so the tail of one launch overlaps the head of
        the next (a 256-block batch is a single round of waves: 24.4 -> ~18 us per batch).  The caller's stream waits for each
        batch, so the returned tensor is ordered like any other -- but it comes from a ring of 4 buffers per stream and is
        OVERWRITTEN by the (4 x pipeline)-th render after it: consume or copy it before (like `graph_replay`).
        `timer`: optional KernelTimer that brackets every launch with HIP events.
        `graph_replay`: in the latency regime, capture the launch sequence of a one-plan graph into a hipGraph
        and replay it per call; the returned tensor is then owned by the graph and OVERWRITTEN by the next
        render -- only for callers that consume each block before asking for the next (BlockDriver.pull)."""
        self.graph_replay = graph_replay and timer is None
        self._captured: _CapturedLaunches | None = None
        self.fuse = fuse
        self.fuse_bus = fuse and fuse_bus           # also fold a SumBus on top of the chain into the launch
        self.fuse_cascade = fuse and fuse_bus       # SumBus([RingMod(] Filter(Filter(Osc)) [, ADSR)]) in one launch (sig_fused_cascade_bus)
        self.node = node
        self.channels = channels
        self.rate = rate
        self.timer = timer
        self._tails: dict[Emitter, tuple[int, torch.Tensor]] = {}    # node -> (end position, last <=100 rows)
        self._stream_end: int | None = None
        self._ctl_programs: dict = {}                      # compiled block-rate control programs, by (port sources, K)
        self._tremolo_order = None                         # (key, held tensors, index, ordered constant rows, ordered pan) of a tremolo-only Sine voice
        self._prev_block_frames: int | None = None         # N of the previous render (where a fused cascade's history block starts)
        self._virtual_history: set = set()                 # bus nodes whose launch of THIS batch kept its filter history implicit (no tails)
        self._cascade_stream: set = set()                  # ... of the previous batch, when this one continues it
        self._tails_rebuilt = False                        # the previous block was re-rendered per node for this batch's tails
        self._recent_blocks: list[tuple[int, int]] = []    # (start, end) of the last few blocks of the contiguous stream rendered so far
        self._stream_blocks = 0                            # blocks of the CURRENT size rendered contiguously since the stream started
        # graphs no fused kernel covers: the per-voice graph as ONE interpreted launch (sig_voice_program) -- True: where that beats
        # one kernel per node (_VoiceProgram.worthwhile), 'always': wherever the graph compiles
        self.fuse_program = fuse if fuse_program is None else (fuse and fuse_program)
        self.specialise = bool(int(os.environ.get('SIG_SPECIALISE', '0'))) if specialise is None else specialise
        self.pipeline = max(1, int(pipeline))
        self._pipe = None                                  # streams, workspaces, pre-bound calls and output rings of the pipelined replay
        self._status: dict[Emitter, runtime.StatusWord] = {}
        self._workspace: torch.Tensor | None = None       # f64 scratch of the fused bus kernel, reused
        self._latency_ws = None                            # ((voices, N, C), zero-initialised scratch of sig_latency_voice_bus)
        self._steady_consts = None                         # (key, control tensors, buffer): closed-form constants kept across calls
        self._gain_products: dict = {}                     # (id, id) -> (gain row, gain row, their product): Gain on both sides of a filter
        self.latency_kernel = True                         # one-launch blocks for Sine chains in the latency regime
        self._replay = None                                # (graph version, N, K, launch(position)) of a one-launch plan
        self.scan_max_chains = SCAN_MAX_CHAINS             # latency regime threshold (tests set 0 to force the serial kernels)
        self.requestor = _EngineSink()                     # the `requestor` of requests the engine itself issues to plugin nodes

    # ------------------------------------------------------------------ public
    def render(self, position: int, block_frames: int, nblocks: int) -> torch.Tensor:
        """Rows = nblocks*block_frames frames from `position`; same values as nblocks sequential
        eager requests of `block_frames` frames."""
        if block_frames < 2:
            raise NotBatchable('block-rate (frames == 1) requests go through the eager path')
        if self._replay is not None:
            version, n, k, launch = self._replay
            if version == graph_clock.version and (n, k) == (block_frames, nblocks):
                # the whole graph is ONE fused launch with no history state: replay it at the new position
                # without re-walking the graph (latency mode: ~10 us of host work per block)
                self._stream_end = position + block_frames * nblocks
                return launch(position)
            self._replay = None
        continuing = self._stream_end == position and (bool(self._tails) or bool(self._virtual_history))
        if not continuing:
            self._tails.clear()
        if self._stream_end != position or self._prev_block_frames != block_frames:
            self._stream_blocks = 0                        # (what a block of another size left in the reference's caches is not modelled: a fresh stream)
        if self._stream_end != position:
            self._recent_blocks = []
        self._cascade_stream = self._virtual_history if continuing else set()    # buses the previous batch ran through the fused cascade
        self._virtual_history = set()                      # filled again by launches that keep their history implicit (fused cascade)
        self._tails_rebuilt = False
        batch = _Batch(self, position, block_frames, nblocks, continuing)
        out = batch.buffer(self.node, self.channels, 0)
        for node, buf in batch.impure_outputs():
            keep = min(CONTEXT, buf.shape[0])
            self._tails[node] = (position + block_frames * nblocks, buf[buf.shape[0] - keep:].clone())
        self._stream_end = position + block_frames * nblocks
        self._prev_block_frames = block_frames
        self._stream_blocks += nblocks
        if block_frames >= CONTEXT:
            first = max(0, nblocks - 4)
            self._recent_blocks = (self._recent_blocks + [(position + b * block_frames, position + (b + 1) * block_frames)
                                                          for b in range(first, nblocks)])[-4:]
        else:
            self._recent_blocks = []
        return out

    def history_starts(self, position: int, count: int) -> list[int]:
        """where the `count` blocks in front of `position` start, oldest first: the previous renders' blocks while the stream is
        contiguous, then -- a fresh graph -- the virtual blocks the reference's context requests create, [p - 100, p) answered
        as a block of its own (chain/__init__.py:149-153, :431-442), clipped at 0.  Fewer than `count` when position 0 is reached."""
        starts = []
        edge = position
        for start, end in reversed(self._recent_blocks):
            if end != edge or len(starts) == count:
                break
            starts.insert(0, start)
            edge = start
        while len(starts) < count and edge > 0:
            edge = max(edge - CONTEXT, 0)
            starts.insert(0, edge)
        return starts

    def _rebuild_tails(self, position: int) -> None:
        """The previous batch ran (part of) the graph through the fused cascade, which leaves no tails, and this one needs
        the per-node schedule for it (a block size the kernel does not take, a node that gained a reader, `fuse_cascade`
        switched off ...): render the previous block [position - prevN, position) per node, as its own block -- an inner
        filter then cold-starts min(100, .) rows in front of it, exactly where the reference cold-started the block it keeps
        cached (chain/__init__.py:431-442, fx.py:93-94) -- and keep the last 100 rows of every request-dependent node."""
        self._tails_rebuilt = True
        prev = self._prev_block_frames
        if not prev or position - prev < 0:
            return
        keep_fuse, keep_virtual, keep_replay = self.fuse_cascade, self._virtual_history, self._replay
        self.fuse_cascade = False
        try:
            sub = _Batch(self, position - prev, prev, 1, False)
            sub.buffer(self.node, self.channels, 0)
            for node, buf in sub.impure_outputs():
                if node not in self._tails:
                    keep = min(CONTEXT, buf.shape[0])
                    self._tails[node] = (position, buf[buf.shape[0] - keep:].clone())
        finally:
            self.fuse_cascade, self._virtual_history, self._replay = keep_fuse, keep_virtual, keep_replay

    def reset(self) -> None:
        self._tails.clear()
        self._virtual_history = set()
        self._recent_blocks = []
        self._stream_blocks = 0
        self._stream_end = None
        self._replay = None
        self._captured = None

    def _remember_replay(self, N: int, K: int, launch) -> None:
        self._replay = (graph_clock.version, N, K, launch)

    # ------------------------------------------------------------------ kernel launch helper
    def _launch(self, name: str, fn: typing.Callable, **meta):
        if self.timer is not None:
            return self.timer.launch(name, fn, **meta)
        return fn()

    def _status_word(self, node: Emitter) -> torch.Tensor:
        word = self._status.get(node)
        if word is None:
            word = self._status[node] = runtime.StatusWord(node.cls_name())
        return word.tensor


def _ctl_const(port: Receiver.BoundPort) -> bool:
    """a control port whose value cannot change from block to block: unplugged, disabled, or a Fixed"""
    src = port.sig
    return src is None or not src.get_state().enabled or isinstance(src, fixed.Fixed)


def _control_ports(node: Emitter) -> list:
    """ports a node reads with forward_at_block_rate"""
    if isinstance(node, osc.Osc):
        return [node.hertz, node.phase]
    if isinstance(node, (fx.Gain, fx.Amp)):
        return [node.right]
    if isinstance(node, fx.Mix):
        return [node.mix]
    if isinstance(node, fx.SingleCritFilter):
        return [node.cutoff]
    if isinstance(node, fx.DoubleCritFilter):
        return [node.low, node.high]
    if isinstance(node, ext.ADSR):
        return [getattr(node, name) for name in _native.ADSR_PARAMS]
    return []


def _audio_ports(node: Emitter) -> list:
    """ports a node reads at frame rate"""
    if isinstance(node, (fx.Mix, fx.RingMod)):
        return [node.left, node.right]
    if isinstance(node, (fx.Gain, fx.Amp)):
        return [node.left]
    if isinstance(node, (fx.CritFilter, ext.SumBus, ext.MixMatrix, ext.Tap, files.FileWriter)):
        return [node.input]
    if isinstance(node, shape.Merge):
        return [node.left, node.right]
    return []


def _modulated(node: Emitter | None) -> bool:
    """some control input of this node is re-evaluated every block (an LFO on a cutoff, FM at block rate ...):
    its reply to a block then depends on which request produced it, like a filter's (the block-rate value is
    read at the REQUEST's position, chain/__init__.py:305-306), so its history rows come from tails, not from
    re-rendering"""
    return node is not None and any(not _ctl_const(p) for p in _control_ports(node))


def _foreign(node: Emitter) -> bool:
    """a node class the engine has no kernel schedule for -- a plugin written against the node API; it is pulled block by
    block through its own respond(), so its reply may depend on the request like a filter's"""
    return not isinstance(node, _KNOWN_TYPES)


def _is_pure(node: Emitter | None, memo: dict) -> bool:
    """position-pure: no filter and no per-block control anywhere upstream on the audio path, so any row
    range can be rendered in one launch and history rows can simply be re-rendered"""
    if node is None or not node.get_state().enabled:
        return True
    if node not in memo:
        memo[node] = True       # cycles are rejected elsewhere
        if isinstance(node, fx.CritFilter) or _modulated(node) or _foreign(node):
            memo[node] = False
        else:
            memo[node] = all(_is_pure(p.sig, memo) for p in _audio_ports(node))
    return memo[node]


class _ControlProgram:
    """The block-rate subgraphs under a set of control ports as ONE launch (sig_control_program): oscillators evaluated at
    one frame per block, element-wise nodes, Fixed rows -- the same expressions as the node-by-node launches of
    `_Batch._control_node`, so the same bits.  Compiled once per (ports, K, graph version); Fixed rows are re-checked per run
    (an edited value re-uploads into a new tensor: recompile).  Outputs are owned by the program and overwritten by the
    next run -- their consumers are launches enqueued before that on the same stream."""

    def __init__(self, srcs: tuple, K: int, lead: int = 0, into: dict | None = None):
        """`lead`: every computed output is rows [lead:] of a (lead + K, cols) buffer `self.full[i]` whose row lead - 1 is the
        `front` row -- the layout sig_voice_program takes its per-block rows in ([rows in front | K blocks]); `into`: {source
        index: (K, cols) tensor} to write into instead of buffers of its own (rows of another program's `full`)"""
        self.srcs, self.K = srcs, K
        self.full: dict[int, torch.Tensor] = {}
        self.ins: list = []
        self.keep: list[torch.Tensor] = []                  # tensors the instructions point into
        self.fixed: list[tuple[fixed.Fixed, torch.Tensor]] = []
        self._reg: dict = {}
        dev = runtime.device()
        self._zero = torch.zeros((1, 1), dtype=CTRL_DTYPE, device=dev)
        self.results: list = []                             # per source: a constant row, None (a Fixed: its resident row, per run) or the output
        self.fronts: dict[int, torch.Tensor] = {}           # per computed source: its (1, cols) row at the front position
        outs = []
        for src in srcs:
            if src is None or not src.get_state().enabled:
                self.results.append(Emitter.empty_result())
            elif isinstance(src, fixed.Fixed):
                if src.resident().shape[0] != 1:
                    raise NotBatchable('multi-row Fixed on a control port')
                self.results.append(None)                   # its resident row, fetched per run
            else:
                reg, cols = self._emit(src)
                i = len(self.results)
                if into is not None:
                    out = into[i]
                    if tuple(out.shape) != (K, cols) or not out.is_contiguous():
                        raise NotBatchable('control rows of another width than the buffer they go into')
                    front = torch.empty((1, cols), dtype=CTRL_DTYPE, device=dev)
                elif lead:
                    self.full[i] = torch.empty((lead + K, cols), dtype=CTRL_DTYPE, device=dev)
                    out, front = self.full[i][lead:], self.full[i][lead - 1:lead]
                else:
                    out = torch.empty((K, cols), dtype=CTRL_DTYPE, device=dev)
                    front = torch.empty((1, cols), dtype=CTRL_DTYPE, device=dev)
                outs.append(_native.CtlOut(reg, cols, out.data_ptr(), front.data_ptr()))
                self.results.append(out)
                self.fronts[i] = front
        self.cols = max([c for _, c in self._reg.values()] + [1])
        # Row instructions first (they depend on nothing; the kernel keeps the one-column ones' loads four in flight), registers renumbered
        row_op = _native.CTL_OPS['Row']
        rank = lambda x: 2 if x.op != row_op else (0 if x.cols == 1 else 1)      # one-column rows, wide rows, the rest
        order = sorted(range(len(self.ins)), key=lambda i: (rank(self.ins[i]), i))
        new_of = {old: new for new, old in enumerate(order)}
        self.ins = [self.ins[i] for i in order]
        for x in self.ins:
            x.dst = new_of[x.dst]
            x.a, x.b, x.c = (new_of.get(r, -1) for r in (x.a, x.b, x.c))
        for o in outs:
            o.reg = new_of[o.reg]
        self.n_ins, self.n_outs = len(self.ins), len(outs)
        self.description = _native.control_program_description(self.ins, outs)     # (its structure: what a specialised kernel is built for)
        self.program_t = _native.upload_structs(self.ins) if self.ins else None
        self.outs_t = _native.upload_structs(outs) if outs else None

    def _row(self, t: torch.Tensor) -> tuple[int, int]:
        if t.shape[0] != 1:
            raise NotBatchable('multi-row Fixed on a control port')
        t = as_control(t)
        self.keep.append(t)
        cols = t.shape[1]
        return self._push(_native.CtlIns(_native.CTL_OPS['Row'], 0, -1, -1, -1, 0, 0 if cols == 1 else 1, 1, cols, 0, t.data_ptr()), cols)

    def _push(self, ins, cols: int) -> tuple[int, int]:
        reg = len(self.ins)
        if reg >= _native.CTL_MAX_REGS:
            raise NotBatchable('control subgraph larger than one program')
        ins.dst = reg
        self.ins.append(ins)
        return reg, cols

    def _emit(self, src) -> tuple[int, int]:
        if src is None or not src.get_state().enabled:
            key = None
            if key not in self._reg:
                self._reg[key] = self._row(self._zero)
            return self._reg[key]
        if src in self._reg:
            return self._reg[src]
        if isinstance(src, fixed.Fixed):
            row = src.resident()
            self.fixed.append((src, row))
            got = self._row(row)
        elif isinstance(src, osc.Osc):
            (a, ca), (b, cb) = self._emit(src.hertz.sig), self._emit(src.phase.sig)
            got = self._push(_native.CtlIns(_native.CTL_OPS['Osc'], _native.OSC_KINDS[src.kind()], a, b, -1, 0, 0, 0, max(ca, cb), 0, None), max(ca, cb))
        elif isinstance(src, (fx.Gain, fx.Amp, fx.Mix, fx.RingMod)):
            (a, ca), (b, cb) = self._emit(src.left.sig), self._emit(src.right.sig)
            c, cc = self._emit(src.mix.sig) if isinstance(src, fx.Mix) else (-1, 1)
            cols = broadcast_shape((1, ca), (1, cb), (1, cc))[1]
            got = self._push(_native.CtlIns(_native.CTL_OPS[type(src).__name__], 0, a, b, c, 0, 0, 0, cols, 0, None), cols)
        else:
            raise NotBatchable(f'no block-rate program for {src.cls_name()}')
        self._reg[src] = got
        return got

    def current(self) -> bool:
        return all(f.resident() is t for f, t in self.fixed)

    def run(self, owner, rate: int, position: int, step: int, front_position: int = -1, min_position: int = 0):
        """the K-row replies; with `front_position` also the (1, cols) replies at that position -> (rows, fronts)"""
        if self.n_outs:
            handle = None
            if owner.specialise:                              # the kernel built for this program's structure (signals_amd/specialise.py)
                from . import specialise
                handle = specialise.ensure_control(self.description, background=owner.specialise == 'background')
            if handle is not None:
                owner._launch('control_program[block-rate]*specialised',
                              lambda: _native.control_program_attached(handle, rate, position, step, self.K, self.cols, self.program_t,
                                                                       self.n_ins, self.outs_t, self.n_outs, front_position, min_position),
                              units=self.K * self.cols)
            else:
                owner._launch('control_program[block-rate]',
                              lambda: _native.control_program(rate, position, step, self.K, self.cols, self.program_t, self.n_ins,
                                                              self.outs_t, self.n_outs, front_position, min_position), units=self.K * self.cols)
        rows = [as_control(src.resident()) if r is None else r for src, r in zip(self.srcs, self.results)]
        if any(t.shape[0] not in (1, self.K) or (r is None and t.shape[0] != 1) for t, r in zip(rows, self.results)):
            raise NotBatchable('multi-row Fixed on a control port')             # (edited since the program was compiled)
        if front_position < 0:
            return rows
        return rows, [self.fronts.get(i, rows[i]) for i in range(len(rows))]


class _Batch:
    """One render call: position, N, K fixed; buffers memoised per (node, channels)."""

    def __init__(self, owner: BatchRenderer, position: int, N: int, K: int, continuing: bool):
        self.owner = owner
        self.pos, self.N, self.K = position, N, K
        self.continuing = continuing
        self.rate = owner.rate
        self._pure: dict = {}
        self._memo: dict[tuple[Emitter, int], tuple[torch.Tensor, int]] = {}
        self._need: dict[tuple[Emitter, int], int] = {}
        self._impure: dict[Emitter, torch.Tensor] = {}
        self._ctl_memo: dict[Emitter, torch.Tensor] = {}

    # -------------------------------------------------------------- control rows
    def _control_const(self, port: Receiver.BoundPort, what: str) -> torch.Tensor:
        """one (1, C) row that holds for every block (Fixed / unplugged); NotBatchable otherwise"""
        src = port.sig
        if src is None or not src.get_state().enabled:
            return Emitter.empty_result()
        if isinstance(src, fixed.Fixed):
            row = src.resident()
            if row.shape[0] != 1:
                raise NotBatchable(f'{what}: multi-row Fixed on a control port')
            return as_control(row)
        raise NotBatchable(f'{what} is driven by {src.cls_name()}; this stage needs block-invariant control rows')

    def _control(self, port: Receiver.BoundPort, what: str) -> torch.Tensor:
        """(1, C) if the value holds for every block, else (K, C): row b is the port's reply to the block-rate
        request at position pos + b*N (forward_at_block_rate of block b)."""
        return self._control_node(port.sig, what)

    def _control_many(self, ports: list, front_position: int = -1):
        """the block-rate replies of several control ports, their subgraphs evaluated in one launch where they consist of
        oscillators, element-wise nodes and Fixed rows (else node by node, like `_control`); with `front_position` also their
        one-row replies at that position: (rows, fronts)"""
        o = self.owner
        srcs = tuple(p.sig for p in ports)
        key = (tuple(id(x) for x in srcs), self.K)
        try:
            held = o._ctl_programs.get(key)
            if held is None or held[0] != graph_clock.version or not held[1].current():
                if len(o._ctl_programs) > 16:
                    o._ctl_programs.clear()
                held = o._ctl_programs[key] = (graph_clock.version, _ControlProgram(srcs, self.K))
            return held[1].run(o, self.rate, self.pos, self.N, front_position)
        except NotBatchable:
            rows = [self._control(p, p.name) for p in ports]
            if front_position < 0:
                return rows
            front = _Batch(o, front_position, 2, 1, False)
            return rows, [front._control(p, p.name) for p in ports]

    def _control_node(self, src: Emitter | None, what: str) -> torch.Tensor:
        if src is None or not src.get_state().enabled:
            return Emitter.empty_result()
        if src in self._ctl_memo:
            return self._ctl_memo[src]
        o, K, dev = self.owner, self.K, runtime.device()
        if isinstance(src, fixed.Fixed):
            row = src.resident()
            if row.shape[0] != 1:
                raise NotBatchable(f'{what}: multi-row Fixed on a control port')
            result = as_control(row)
        elif isinstance(src, osc.Osc):
            hertz, phase = self._control(src.hertz, 'hertz'), self._control(src.phase, 'phase')
            _, voices = broadcast_shape((1, 1), hertz.shape[-2:], phase.shape[-2:])
            result = torch.empty((K, voices), dtype=CTRL_DTYPE, device=dev)
            o._launch(f'osc_bank[{src.kind()},block-rate]',
                      lambda: _native.osc_bank(src.kind(), self.pos, self.rate, hertz, phase, result,
                                               step=self.N, rows_per_param=1),
                      units=K * voices)
        elif isinstance(src, (fx.Gain, fx.Amp, fx.Mix, fx.RingMod)):
            name = type(src).__name__
            a = self._control(src.left, 'left')
            b = self._control(src.right, 'right')
            c = self._control(src.mix, 'mix') if isinstance(src, fx.Mix) else None
            ops = [a, b] + ([c] if c is not None else [])
            _, cols = broadcast_shape(*((1, t.shape[1]) for t in ops))
            rows = max(t.shape[0] for t in ops)
            result = torch.empty((rows, cols), dtype=CTRL_DTYPE, device=dev)
            o._launch(f'elementwise[{name},block-rate]', lambda: _native.elementwise(name, a, b, c, result),
                      units=rows * cols)
        else:
            raise NotBatchable(f'{what}: no block-rate schedule for {src.cls_name()}')
        self._ctl_memo[src] = result
        return result

    # -------------------------------------------------------------- history requirements
    def _require(self, node: Emitter | None, channels: int, hist: int) -> None:
        """propagate how many history rows each (node, channels) buffer must carry"""
        if node is None:
            return
        key = (node, channels)
        if self._need.get(key, -1) >= hist:
            return
        self._need[key] = hist
        if not node.get_state().enabled and not isinstance(node, (ext.Tap, files.FileWriter)):
            return
        if isinstance(node, fx.CritFilter):
            self._require(node.input.sig, channels, min(CONTEXT, self.pos))
        elif _modulated(node):
            for port in _audio_ports(node):                       # own history comes from the tail / a fresh block
                self._require(port.sig, channels, 0)
        elif isinstance(node, (fx.Mix, fx.RingMod)):
            self._require(node.left.sig, channels, hist)
            self._require(node.right.sig, channels, hist)
        elif isinstance(node, (fx.Gain, fx.Amp)):
            self._require(node.left.sig, channels, hist)
        elif isinstance(node, ext.SumBus):
            self._require(node.input.sig, node.input.channels, hist)
        elif isinstance(node, (ext.MixMatrix, ext.Tap, files.FileWriter)):
            self._require(node.input.sig, channels, hist)
        elif isinstance(node, shape.Merge):
            self._require(node.left.sig, node.left.channels, hist)
            self._require(node.right.sig, node.right.channels, hist)

    # -------------------------------------------------------------- buffers
    def buffer(self, node: Emitter | None, channels: int, hist: int) -> torch.Tensor:
        """(hist + K*N, C') tensor: rows [hist:] are the K blocks, rows [:hist] the node's reply to a
        single block request [pos-hist, pos)."""
        self._require(node, channels, hist)
        full, have = self._materialise(node, channels)
        return full[have - hist:] if have != hist else full

    def _materialise(self, node: Emitter | None, channels: int) -> tuple[torch.Tensor, int]:
        """(buffer, history rows it carries) of one node for this batch, memoised per (node, channels)"""
        key = (node, channels)
        if key in self._memo:
            return self._memo[key]
        if node is not None and isinstance(node, (ext.Tap, files.FileWriter)) and not node.get_state().enabled:
            self._memo[key] = self._materialise(node.input.sig, channels)      # PASSTHRU: disabled = forward input
            return self._memo[key]
        if node is None or not node.get_state().enabled:
            self._memo[key] = (Emitter.empty_result(), 0)                      # (1,1) zeros broadcast everywhere
            return self._memo[key]
        hist = self._need[key]
        rows = hist + self.N * self.K

        result = _VoiceChain.match_and_launch(self, node, channels, hist) if self.owner.fuse else None
        if result is None and self.owner.fuse_program and hist == 0 and isinstance(node, _VoiceProgram.KERNEL_NODES):
            result = self._program_store(node, channels)
        if result is None:
            for types, build in self._SCHEDULES:
                if isinstance(node, types):
                    result = build(self, node, channels, hist, rows)
                    break
            else:
                result = self._sched_foreign(node, channels, hist, rows)
        if isinstance(result, tuple):                                          # a pass-through shares its input's buffer
            self._memo[key] = result
            return result
        if not _is_pure(node, self._pure) and result.shape[0] > 1:
            self._impure[node] = result
        self._memo[key] = (result, hist if result.shape[0] > 1 else 0)
        return self._memo[key]

    def _operand(self, port: Receiver.BoundPort, channels: int, hist: int) -> torch.Tensor:
        """an input as rows [pos-hist, pos+K*N) or a one-row broadcast"""
        full, have = self._materialise(port.sig, channels)
        if full.shape[0] == 1:
            return full
        return full[have - hist:] if have != hist else full

    # -------------------------------------------------------------- one schedule per node family
    def _sched_fixed(self, node, channels, hist, rows):
        value = node.resident()
        if value.shape[0] != 1:
            raise NotBatchable('multi-row Fixed as an audio source')
        return value

    def _sched_osc(self, node, channels, hist, rows):
        o = self.owner
        hertz = self._control(node.hertz, 'hertz')
        phase = self._control(node.phase, 'phase')
        _, voices = broadcast_shape((1, 1), (1, hertz.shape[1]), (1, phase.shape[1]))
        result = torch.empty((rows, voices), dtype=AUDIO_DTYPE, device=runtime.device())
        if _modulated(node):
            # hertz / phase are re-read every block: K parameter rows, N output rows each
            main = result[hist:]
            o._launch(f'osc_bank[{node.kind()},per-block]',
                      lambda: _native.osc_bank(node.kind(), self.pos, self.rate, hertz, phase, main,
                                               rows_per_param=self.N),
                      units=main.shape[0] * voices)
            self._own_history(node, voices, hist, result)
        else:
            start = self.pos - hist
            o._launch(f'osc_bank[{node.kind()}]',
                      lambda: _native.osc_bank(node.kind(), start, self.rate, hertz, phase, result),
                      units=rows * voices)
        return result

    def _sched_noise(self, node, channels, hist, rows):
        result = torch.empty((rows, channels), dtype=AUDIO_DTYPE, device=runtime.device())
        seed, start = node.get_state().seed, self.pos - hist
        return self.owner._launch('white_noise', lambda: _native.white_noise(seed, start, result), units=rows * channels)

    def _sched_filter(self, node, channels, hist, rows):
        return self._filter(node, channels, hist, rows)

    def _sched_elementwise(self, node, channels, hist, rows):
        o, dev = self.owner, runtime.device()
        if o.fuse and isinstance(node, fx.RingMod):
            enveloped = self._ringmod_with_envelope(node, channels, hist, rows)
            if enveloped is not None:
                return enveloped
        mod = _modulated(node)
        in_hist = 0 if mod else hist                              # a modulated node's history comes from its tail
        a = self._operand(node.left, channels, in_hist)
        if isinstance(node, (fx.Gain, fx.Amp)):
            b, c = self._control(node.right, 'right'), None
        else:
            b = self._operand(node.right, channels, in_hist)
            c = self._control(node.mix, 'mix') if isinstance(node, fx.Mix) else None
        ctl = c if isinstance(node, fx.Mix) else (b if isinstance(node, (fx.Gain, fx.Amp)) else None)
        audio = [a] + ([b] if isinstance(node, (fx.Mix, fx.RingMod)) else [])
        cols = broadcast_shape(*((1, t.shape[1]) for t in audio + ([ctl] if ctl is not None else [])))[1]
        name = type(node).__name__
        if all(t.shape[0] == 1 for t in audio) and not mod:
            result = torch.empty((1, cols), dtype=CTRL_DTYPE, device=dev)         # every operand is a one-row reply
            return o._launch(f'elementwise[{name}]', lambda: _native.elementwise(name, a, b, c, result), units=cols)
        result = torch.empty((rows, cols), dtype=AUDIO_DTYPE, device=dev)
        main = result[hist:] if mod else result
        o._launch(f'elementwise[{name}{",per-block" if mod else ""}]',
                  lambda: _native.elementwise(name, a, b, c, main), units=main.shape[0] * cols)
        if mod:
            self._own_history(node, cols, hist, result)
        return result

    def _enveloped_filter(self, node, channels):
        """RingMod(filter, ADSR) where nothing else reads the envelope or the filter and the filter's rows are not
        in the batch yet: (filter, ADSR control rows), else None"""
        for env_port, x_port in ((node.right, node.left), (node.left, node.right)):
            env, flt = env_port.sig, x_port.sig
            if (isinstance(env, ext.ADSR) and env.get_state().enabled and len(env.outputs_with_ports) == 1
                    and not _modulated(env) and isinstance(flt, fx.SingleCritFilter) and flt.get_state().enabled
                    and len(flt.outputs_with_ports) == 1 and (flt, channels) not in self._memo):
                ctl = env.control_rows(lambda bound: self._control_const(bound, bound.name))
                voices = broadcast_shape((1, 1), *(r.shape for r in ctl.values()))[1]
                if voices in (1, channels):
                    return flt, ctl
        return None

    def _ringmod_with_envelope(self, node, channels, hist, rows):
        """RingMod(x, ADSR) with an envelope nobody else reads: the envelope multiplies x on the fly -- in the
        epilogue of the filter that produces x when nothing else reads that filter (sig_biquad_coldstart_env),
        else in one envelope * x pass (sig_adsr_apply)"""
        for env_port, x_port in ((node.right, node.left), (node.left, node.right)):
            env = env_port.sig
            if (isinstance(env, ext.ADSR) and env.get_state().enabled and len(env.outputs_with_ports) == 1
                    and not _modulated(env) and x_port.sig is not None and not isinstance(x_port.sig, ext.ADSR)):
                ctl = env.control_rows(lambda bound: self._control_const(bound, bound.name))
                voices = broadcast_shape((1, 1), *(r.shape for r in ctl.values()))[1]
                flt = x_port.sig
                if (isinstance(flt, fx.SingleCritFilter) and flt.get_state().enabled and len(flt.outputs_with_ports) == 1
                        and voices in (1, channels) and (flt, channels) not in self._memo):
                    return self._filter(flt, channels, hist, rows, envelope=ctl, owner_node=node)
                x = self._operand(x_port, channels, hist)
                if x.shape[0] == 1 or x.shape[1] != voices or x.dtype != AUDIO_DTYPE:
                    return None
                result = torch.empty((rows, voices), dtype=AUDIO_DTYPE, device=runtime.device())
                start = self.pos - hist
                return self.owner._launch('adsr_apply', lambda: _native.adsr_apply(start, self.rate, ctl, x, result),
                                          units=rows * voices)
        return None

    def _sched_bus(self, node, channels, hist, rows):
        o = self.owner
        src_port, gains = node.input, node.resident_gains()
        top = node.input.sig
        if (o.fuse and isinstance(top, fx.Gain) and top.get_state().enabled and _ctl_const(top.right)
                and len(top.outputs_with_ports) == 1 and top.left.sig is not None):
            # a per-voice Gain feeding only this bus is a diagonal scaling of the mix weights:
            # sum_v pan[c,v] * (g[v] * x[n,v]) = sum_v (pan[c,v] * g[v]) * x[n,v]  -- fold it, skip the launch
            g = self._control_const(top.right, 'right')
            voices = top.left.channels
            if g.shape[1] in (1, voices) and (gains is None or gains.shape[1] == voices):
                gains = (gains * g) if gains is not None else g.expand(1, voices).contiguous()
                src_port = top.left
                self._require(src_port.sig, voices, hist)
        fused = self._bus_over_cascade(node, src_port, gains, rows) if o.fuse_cascade and hist == 0 else None
        if fused is None and o.fuse_program and hist == 0:
            fused = self._program_bus(node, src_port, gains, rows)
        if fused is None:
            fused = self._bus_over_filter(node, src_port, gains, hist, rows) if o.fuse and hist == 0 else None
        if fused is not None:
            return fused
        x = self._operand(src_port, src_port.channels, hist)
        if x.shape[0] == 1:
            raise NotBatchable('SumBus over a one-row input')
        result = torch.empty((rows, node.channels), dtype=AUDIO_DTYPE, device=runtime.device())
        return o._launch('sum_bus', lambda: _native.sum_bus(x, gains, result), units=rows * x.shape[1])

    def _program_store(self, node, channels):
        """the per-voice graph under `node` as one interpreted launch (sig_voice_program) storing its rows, or None"""
        memo_keys = {k[0] for k in self._memo}
        prog = _VoiceProgram.compile(self, node, channels)
        if prog is None or any(n in memo_keys for n in prog.uses if n is not node):
            return None                                                        # (an inner node already has rows in this batch: someone else reads it)
        if self.owner.fuse_program != 'always' and not prog.worthwhile():
            return None
        # the node's natural width: every control row is one column wide -> the reply is (rows, 1), broadcast by the consumer
        try:
            tensors = [c if c is not None else self._control_const(p, p.name) if _ctl_const(p) else None for p, c, _ in prog.controls]
        except NotBatchable:
            return None
        wide = any(t is None or t.shape[1] > 1 for t in tensors) or any(isinstance(n, (noise.White, ext.ADSR)) for n in prog.uses)
        voices = channels if wide else 1
        if voices != channels:
            prog = _VoiceProgram.compile(self, node, voices)
            if prog is None:
                return None
        out = torch.empty((self.N * self.K, voices), dtype=AUDIO_DTYPE, device=runtime.device())
        try:
            return prog.launch(out, None, False, f'voice_program[{prog.describe()}]')
        except (_NoProgram, NotBatchable):
            return None                                                        # (the per-node schedule decides: it may refuse the batch as a whole)

    def _program_bus(self, node, src_port, gains, rows):
        """SumBus over a per-voice graph no fused kernel covers: graph and bus in one interpreted launch, or None"""
        voices, C = src_port.channels, node.channels
        if C not in (1, 2) or voices is None or (gains is not None and gains.shape[1] != voices):
            return None
        memo_keys = {k[0] for k in self._memo}
        prog = _VoiceProgram.compile(self, src_port.sig, voices, min_nodes=1)
        if prog is None or any(n in memo_keys for n in prog.uses):
            return None
        if self.owner.fuse_program != 'always' and not prog.worthwhile():
            return None
        if len(src_port.sig.outputs_with_ports) != 1:
            return None                                                        # (the bus input has another reader: its rows must exist)
        out = torch.empty((rows, C), dtype=AUDIO_DTYPE, device=runtime.device())
        try:
            result = prog.launch(out, gains, True, f'voice_program_bus[{prog.describe()}]')
        except (_NoProgram, NotBatchable):
            return None
        if prog.depth:
            self.owner._virtual_history.add(node)
        return result

    def _bus_over_cascade(self, node, src_port, gains, rows):
        """SumBus([RingMod(] Filter2(Filter1(Osc)) [, ADSR)]) with block-invariant controls and no other reader of any of
        them: the whole voice in ONE launch (sig_fused_cascade_bus), nothing per-voice through HBM.  The reference's
        cache history between the two filters is reproduced inside the kernel from where the previous
        block started: position - N on a continuing stream, position - min(100, position) on a fresh graph."""
        o, top, voices, N = self.owner, src_port.sig, src_port.channels, self.N
        C = node.channels
        if C not in (1, 2, 4) or N <= CONTEXT or N % (16 // C):
            return None

        def sole(n, kind):
            return (isinstance(n, kind) and n.get_state().enabled and len(n.outputs_with_ports) == 1 and (n, voices) not in self._memo)
        ctl = None
        f2 = top
        if sole(top, fx.RingMod) and not _modulated(top):
            for env_port, x_port in ((top.right, top.left), (top.left, top.right)):
                env = env_port.sig
                if sole(env, ext.ADSR) and not _modulated(env) and isinstance(x_port.sig, fx.SingleCritFilter):
                    ctl = env.control_rows(lambda bound: self._control_const(bound, bound.name))
                    f2 = x_port.sig
                    break
            else:
                return None
        if not sole(f2, fx.SingleCritFilter):
            return None
        f1 = f2.input.sig
        if not sole(f1, fx.SingleCritFilter):
            return None
        src = f1.input.sig
        if not sole(src, osc.Osc) or any(not _ctl_const(p) for p in (src.hertz, src.phase, f1.cutoff, f2.cutoff)):
            return None
        hertz, phase = self._control_const(src.hertz, 'hertz'), self._control_const(src.phase, 'phase')
        cut1, cut2 = self._control_const(f1.cutoff, 'cutoff'), self._control_const(f2.cutoff, 'cutoff')
        if max(hertz.shape[1], phase.shape[1]) != voices or cut1.shape[1] != voices or cut2.shape[1] != voices:
            return None                                                        # (the reference indexes cutoff[0, i] per channel)
        if any(t.shape[1] not in (1, voices) for t in (hertz, phase, *(ctl or {}).values())):
            return None
        if gains is not None and gains.shape[1] != voices:
            return None
        if self.pos == 0:
            history = 0
        elif self.continuing:
            # the block in front of this batch: the kernel re-walks it from where the reference cold-started it.  Only if
            # the previous render kept its history implicit as well (a per-node batch left tails of rounded float32 rows)
            # and that block covers the outer filter's context
            if node not in o._cascade_stream or not o._prev_block_frames or o._prev_block_frames < min(CONTEXT, self.pos):
                return None
            history = self.pos - o._prev_block_frames
        else:
            history = self.pos - min(CONTEXT, self.pos)
        result = torch.empty((rows, C), dtype=AUDIO_DTYPE, device=runtime.device())
        need = _native.lib().sig_fused_voice_bus_workspace(voices, rows, C) // 8
        if o._workspace is None or o._workspace.numel() < need:
            o._workspace = torch.empty(need, dtype=CTRL_DTYPE, device=runtime.device())
        status = o._status_word(f2)                                            # one word for the launch: both designs report here
        kind, t1, t2 = src.kind(), str(f1.type()), str(f2.type())
        o._virtual_history.add(node)
        out = o._launch(f'fused_cascade_bus[{kind},{t1},{t2}{",env" if ctl else ""}]',
                        lambda: _native.fused_cascade_bus(kind, t1, t2, self.rate, self.pos, history, N, self.K, CONTEXT, voices,
                                                          hertz, phase, cut1, cut2, None, ctl, gains, result,
                                                          workspace=o._workspace, status=status),
                        units=rows * voices)
        return out

    def _bus_over_filter(self, node, src_port, gains, hist, rows):
        """SumBus(Filter(x)) / SumBus(RingMod(Filter(x), ADSR)) with no other reader of the filter (and of the
        RingMod and the envelope): one pass over x, nothing per-voice stored (sig_biquad_coldstart_bus)"""
        o, top, voices = self.owner, src_port.sig, src_port.channels
        if node.channels not in (1, 2, 4):                                     # the bus widths the tile reduction is built for
            return None
        if top is None or not top.get_state().enabled or len(top.outputs_with_ports) != 1 or (top, voices) in self._memo:
            return None
        flt, ctl = top, None
        if isinstance(top, fx.RingMod) and not _modulated(top):
            found = self._enveloped_filter(top, voices)
            if found is None:
                return None
            flt, ctl = found
        if not isinstance(flt, fx.SingleCritFilter) or not flt.get_state().enabled or (flt, voices) in self._memo:
            return None
        if gains is not None and gains.shape[1] != voices:
            return None
        cutoff, window, c0 = self._filter_window(flt, voices)
        if window.dtype != AUDIO_DTYPE or window.shape[1] != voices:
            return None
        result = torch.empty((rows, node.channels), dtype=AUDIO_DTYPE, device=runtime.device())
        need = _native.lib().sig_fused_voice_bus_workspace(voices, rows, node.channels) // 8
        if o._workspace is None or o._workspace.numel() < need:
            o._workspace = torch.empty(need, dtype=CTRL_DTYPE, device=runtime.device())
        btype, status = str(flt.type()), o._status_word(flt)
        return o._launch(f'biquad_bus[{btype}{",env" if ctl else ""}]',
                         lambda: _native.biquad_coldstart_bus(btype, self.rate, self.pos, self.N, self.K, CONTEXT, cutoff,
                                                              window, c0, gains, result, envelope=ctl,
                                                              workspace=o._workspace, status=status),
                         units=rows * voices)

    def _sched_tap(self, node, channels, hist, rows):
        return self._materialise(node.input.sig, channels)                    # pass-through: same buffer

    def _sched_file_writer(self, node, channels, hist, rows):
        full, have = self._materialise(node.input.sig, channels)              # pass-through + record the batch
        if full.shape[0] > 1:
            node.write_rows(self.pos, self.rate, channels, full[have:])
        return full, have

    def _sched_file_reader(self, node, channels, hist, rows):
        result = node.read_rows(self.pos - hist, rows, self.rate, channels)
        if result.shape[0] != rows:
            raise NotBatchable('FileReader ran past the end of the file inside a batch')
        return result

    def _sched_adsr(self, node, channels, hist, rows):
        ctl = node.control_rows(lambda bound: self._control_const(bound, bound.name))
        _, voices = broadcast_shape((1, 1), *(r.shape for r in ctl.values()))
        result = torch.empty((rows, voices), dtype=AUDIO_DTYPE, device=runtime.device())
        start = self.pos - hist
        return self.owner._launch('adsr', lambda: _native.adsr(start, self.rate, ctl, result), units=rows * voices)

    def _sched_mix_matrix(self, node, channels, hist, rows):
        x = self._operand(node.input, channels, hist)
        if x.shape[0] == 1 or x.shape[1] % 64:
            raise ValueError(f'MixMatrix needs (rows, 64*g) audio, got {tuple(x.shape)}')
        if x.dtype != AUDIO_DTYPE or not x.is_contiguous():
            x = x.to(AUDIO_DTYPE).contiguous()
        matrix = node.resident_matrix()
        result = torch.empty_like(x)
        return self.owner._launch('mix_matrix', lambda: _native.mix_matrix(x, matrix, result),
                                  units=x.shape[0] * x.shape[1])

    def _sched_merge(self, node, channels, hist, rows):
        left = self._operand(node.left, node.left.channels, hist)
        right = self._operand(node.right, node.right.channels, hist)
        if left.shape[0] != right.shape[0]:
            raise ValueError('all the input array dimensions except for the concatenation axis must match exactly')
        return torch.cat((left.to(AUDIO_DTYPE), right.to(AUDIO_DTYPE)), dim=1)     # buffer plumbing

    def _sched_foreign(self, node, channels, hist, rows):
        """A node class without a kernel schedule (a plugin, e.g. one written against the reference and answering numpy
        arrays, chain/__init__.py:245-247): its K blocks are pulled one request at a time through its own respond() --
        which pulls ITS inputs through the eager path -- and laid out as one batch buffer, so everything downstream
        still runs one launch per node.  History rows come from the previous batch's tail or a fresh block request."""
        from signals_amd.chain import BadShape, BlockLoc, Request, Shape, adopt_reply
        N, K = self.N, self.K
        blocks = []
        for b in range(K):
            loc = BlockLoc(position=self.pos + b * N, rate=self.rate, shape=Shape(frames=N, channels=channels))
            block = adopt_reply(node.respond(Request(requestor=self.owner.requestor, port='input', loc=loc)))
            if not (Shape.of_array(block) <= loc.shape):
                raise BadShape(node, block.shape, loc.shape)
            blocks.append(block)
        widths = {int(b.shape[1]) for b in blocks}
        if len(widths) != 1:
            raise NotBatchable(f'{node.cls_name()} answered blocks of different widths {sorted(widths)}')
        if all(b.shape[0] == 1 for b in blocks) and K == 1:
            return blocks[0].to(CTRL_DTYPE)                                   # a one-row reply broadcasts, like a Fixed
        result = torch.empty((rows, widths.pop()), dtype=AUDIO_DTYPE, device=runtime.device())
        for b, block in enumerate(blocks):
            result[hist + b * N: hist + (b + 1) * N] = block                  # (a one-row reply broadcasts over its block)
        self._own_history(node, result.shape[1], hist, result)
        return result

    _SCHEDULES = (
        (fixed.Fixed, _sched_fixed),
        (osc.Osc, _sched_osc),
        (noise.White, _sched_noise),
        (fx.CritFilter, _sched_filter),
        ((fx.Mix, fx.RingMod, fx.Gain, fx.Amp), _sched_elementwise),
        (ext.SumBus, _sched_bus),
        (ext.Tap, _sched_tap),
        (files.FileWriter, _sched_file_writer),
        (files.FileReader, _sched_file_reader),
        (ext.ADSR, _sched_adsr),
        (ext.MixMatrix, _sched_mix_matrix),
        (shape.Merge, _sched_merge),
    )

    # -------------------------------------------------------------- filters
    def _filter_window(self, node: fx.CritFilter, channels: int, cutoff: torch.Tensor | None = None):
        """(cutoff rows, input window with c0 = min(100, pos) context rows in front, c0) of a filter; `cutoff`
        defaults to the single-cutoff filters' control rows"""
        N, K, pos = self.N, self.K, self.pos
        if cutoff is None:
            cutoff = self._control(node.cutoff, 'cutoff')
        c0 = min(CONTEXT, pos)
        src = node.input.sig
        pure_in = _is_pure(src, self._pure)
        if not pure_in and N <= CONTEXT and (K > 1 or self.continuing):
            raise NotBatchable('cascaded filters with block size <= 100 depend on the after-window cache entries')
        window, have = self._materialise(src, channels)
        if window.shape[0] == 1:
            raise ValueError('filter input answered a single row (unplugged or disabled input)')
        window = window[have - c0:] if have != c0 else window
        if window.shape[1] < channels:
            raise IndexError(f'index {window.shape[1]} is out of bounds for axis 1 with size {window.shape[1]}')
        if cutoff.shape[1] < channels:
            raise IndexError(f'index {cutoff.shape[1]} is out of bounds for axis 1 with size {cutoff.shape[1]}')
        window = window[:, :channels]
        cutoff = cutoff[:, :channels]
        if not cutoff.is_contiguous():
            cutoff = cutoff.contiguous()
        return cutoff, window, c0

    def _filter(self, node: fx.CritFilter, channels: int, hist: int, rows: int, envelope: dict | None = None,
                owner_node: Emitter | None = None) -> torch.Tensor:
        """`envelope` / `owner_node`: the filter runs on behalf of RingMod(filter, ADSR) -- its stored rows are
        multiplied by the envelope and the buffer (history rows, tail) belongs to that RingMod node"""
        o = self.owner
        N, K, pos = self.N, self.K, self.pos
        band = isinstance(node, fx.DoubleCritFilter)
        cutoff, window, c0 = self._filter_window(node, channels, self._control_const(node.low, 'low') if band else None)
        result = torch.empty((rows, channels), dtype=AUDIO_DTYPE, device=window.device)
        main = result[hist:]
        btype = str(node.type())
        status = o._status_word(node)
        if band:
            high = self._control_const(node.high, 'high')
            if high.shape[1] < channels:
                raise IndexError(f'index {high.shape[1]} is out of bounds for axis 1 with size {high.shape[1]}')
            high = high[:, :channels]
            if not high.is_contiguous():
                high = high.contiguous()
            o._launch(f'band_coldstart[{btype}]',
                      lambda: _native.band_coldstart(btype, self.rate, pos, N, K, CONTEXT, cutoff, high, window, c0, main,
                                                     status=status),
                      units=N * K * channels)
        else:
            o._launch(f'biquad_coldstart[{btype}{",env" if envelope else ""}]',
                      lambda: _native.biquad_coldstart(btype, self.rate, pos, N, K, CONTEXT, cutoff, window, c0, main,
                                                       status=status, envelope=envelope),
                      units=N * K * channels)
        self._own_history(owner_node or node, channels, hist, result)
        return result

    def _own_history(self, node: Emitter, channels: int, hist: int, result: torch.Tensor) -> None:
        """rows [:hist] of a request-dependent node (filter, or a node with per-block control): the previous
        batch's tail when the stream continues, else the node rendered as its own block [pos-hist, pos) -- what
        the reference's block cache, respectively a fresh graph, would answer"""
        if not hist:
            return
        o, pos = self.owner, self.pos
        tail = o._tails.get(node) if self.continuing else None
        if tail is None and self.continuing and o._cascade_stream and not o._tails_rebuilt:
            o._rebuild_tails(pos)                                              # (the previous batch kept this history implicit)
            tail = o._tails.get(node)
        if tail is not None and tail[0] == pos and tail[1].shape[0] >= hist and tail[1].shape[1] >= channels:
            result[:hist].copy_(tail[1][tail[1].shape[0] - hist:, :channels])
        else:
            sub = _Batch(o, pos - hist, hist, 1, False)
            result[:hist].copy_(sub.buffer(node, channels, 0))

    def impure_outputs(self):
        return self._impure.items()


class _VoiceChain:
    """The fusable pattern  [SumBus(] [Gain(] LowPass|HighPass(Osc) [)] [)]  matched on a graph, when nothing else
    consumes the intermediate nodes and every control input is block-invariant.  Launches it as
    sig_fused_osc_biquad (chain) or sig_fused_voice_bus (chain + bus); in the latency regime the chain runs as a
    prefix scan and the bus as its own launch, optionally captured into a hipGraph."""

    def __init__(self, batch: _Batch, src, filt, gain_node, bus_node, channels: int, pre_gain=None, pair=None):
        self.batch, self.src, self.filt, self.gain_node, self.bus_node = batch, src, filt, gain_node, bus_node
        self.pre_gain = pre_gain                                   # a Gain between oscillator and filter, folded into the output gain
        self.pair = pair                                           # (Mix | RingMod node, second oscillator): the filter reads op(src, second)
        self.pair_rows = None                                      # (hertz2, phase2, mix) as resolve() found them
        self.channels = channels                                   # voices of the chain
        self.gain_ports = [g.right for g in (pre_gain, gain_node) if g is not None]
        self.ports = [src.hertz, src.phase, filt.cutoff] + self.gain_ports
        self.involved = [n for n in (src, filt, gain_node, bus_node, pre_gain, *(pair or ())) if n is not None]
        self.kind, self.btype = src.kind(), str(filt.type())
        self.fm = not (_ctl_const(src.hertz) and _ctl_const(src.phase))   # block-rate frequency / phase modulation (osc.py:28-30)
        self.hist_rows = (None, None)                              # hertz / phase of the block in front of the batch, as resolve() found them
        self.modulated = self.fm or any(not _ctl_const(p) for p in [filt.cutoff] + self.gain_ports)   # per-block parameter rows
        self.general = self.modulated or pair is not None          # the walker's general entry points (sig_fused_*_rows / *_pair / *_fm)
        source = self.kind if pair is None else f'{type(pair[0]).__name__}({self.kind},{pair[1].kind()})'
        self.tag = (f'{source},{self.btype}{",gain" if self.gain_ports else ""}{",per-block" if self.modulated else ""}'
                    f'{",fm" if self.fm else ""}')
        if pair is not None:
            self.ports += [pair[1].hertz, pair[1].phase] + ([pair[0].mix] if isinstance(pair[0], fx.Mix) else [])

    # ---- matching
    @classmethod
    def match_and_launch(cls, batch: _Batch, node: Emitter, channels: int, hist: int) -> torch.Tensor | None:
        bus_node, mix_node, top = None, None, node
        if isinstance(node, ext.SumBus):
            if not batch.owner.fuse_bus or hist != 0 or node.channels not in (1, 2, 4):
                return None
            bus_node, top = node, node.input.sig
            if top is None or not top.get_state().enabled or len(top.outputs_with_ports) != 1:
                return None
            channels = node.input.channels
        elif isinstance(node, ext.MixMatrix):
            if hist != 0 or channels % 64:
                return None
            mix_node, top = node, node.input.sig
            if top is None or not top.get_state().enabled or len(top.outputs_with_ports) != 1:
                return None
        gain_node, filt = None, top
        if isinstance(top, fx.Gain):
            gain_node, filt = top, top.left.sig
            if not isinstance(filt, fx.SingleCritFilter) or len(filt.outputs_with_ports) != 1:
                return None
        if not isinstance(filt, fx.SingleCritFilter) or not filt.get_state().enabled:
            return None
        src, pre_gain = filt.input.sig, None
        if (isinstance(src, fx.Gain) and src.get_state().enabled and _ctl_const(src.right) and isinstance(src.left.sig, osc.Osc)
                and (src, channels) not in batch._memo):
            # Filter(Gain(Osc)) = Gain(Filter(Osc)): the filter is linear and starts every block from zero state, so a
            # block-invariant gain in front of it is a factor of the output weight (lowpass_test.sigs: Triangle -> Gain ->
            # LowPass).  The Gain may have other readers: they get its rows from the per-node schedule as usual.
            pre_gain, src = src, src.left.sig
        pair = None
        if (pre_gain is None and isinstance(src, (fx.Mix, fx.RingMod)) and src.get_state().enabled and not _modulated(src)
                and len(src.outputs_with_ports) == 1 and (src, channels) not in batch._memo):
            # Filter(Mix | RingMod(Osc, Osc)): both oscillators are evaluated per row inside the walker
            left, right = src.left.sig, src.right.sig
            if (isinstance(left, osc.Osc) and isinstance(right, osc.Osc) and left is not right and right.get_state().enabled
                    and len(right.outputs_with_ports) == 1 and _ctl_const(right.hertz) and _ctl_const(right.phase)):
                pair, src = (src, right), left
        if not isinstance(src, osc.Osc) or not src.get_state().enabled or len(src.outputs_with_ports) != 1:
            return None
        fm = not (_ctl_const(src.hertz) and _ctl_const(src.phase))
        if fm and (pair is not None or batch.N < CONTEXT):
            # two oscillators AND block-rate FM: per node.  Blocks shorter than the context: the context request [p - 100, p) is
            # then contained in no single cached block of the oscillator, so the reference answers it as a block of its own
            # (controls read at p - 100) -- not the previous block's samples the fused walker warms up on: per node too
            return None
        chain = cls(batch, src, filt, gain_node, bus_node, channels, pre_gain, pair)
        controls = chain.resolve()
        if controls is None or not chain.widths_ok(controls):
            return None
        if mix_node is not None:
            return chain.launch_mix(mix_node, controls)
        return chain.launch_bus(node, controls) if bus_node is not None else chain.launch_chain(node, controls, hist)

    def resolve(self):
        """[hertz, phase, cutoff, gain|None] as they are NOW (a Fixed re-uploads when its array changed);
        None if the pattern no longer holds"""
        if not all(n.get_state().enabled for n in self.involved):
            return None
        try:
            if self.pair is not None:
                op, second = self.pair
                self.pair_rows = (self.batch._control_const(second.hertz, 'hertz'), self.batch._control_const(second.phase, 'phase'),
                                  self.batch._control_const(op.mix, 'mix') if isinstance(op, fx.Mix) else None)
            if self.modulated:
                # hertz / phase / cutoff / gain driven by computed block-rate signals (vibrato, an LFO sweep, a tremolo): K rows
                # each, one per block, all of them from ONE control-program launch
                b, o = self.batch, self.batch.owner
                ports = self.ports[:2] + [self.filt.cutoff] + self.gain_ports
                # ... and, under block-rate FM, the hertz / phase of the block in FRONT of the batch (evaluated by the same
                # launch): the reference's oscillators keep their previous block (BlockCachingEmitter), so block 0's context rows
                # are that block's samples -- the previous batch's last block on a contiguous stream, else the context request
                # [pos - c, pos) answered as a block of its own (its controls read at pos - c)
                q = -1
                if self.fm:
                    contiguous = o._stream_end == b.pos and bool(o._prev_block_frames) and o._prev_block_frames >= min(CONTEXT, b.pos)
                    q = b.pos - (o._prev_block_frames if contiguous else min(CONTEXT, b.pos))
                got = b._control_many(ports, q)
                vals, fronts = (got, None) if q < 0 else got
                vals = [as_control(t) for t in vals]
                rows, gains = vals[:3], vals[3:]
                if self.fm:
                    self.hist_rows = tuple(None if _ctl_const(p) else as_control(f) for p, f in zip(self.ports[:2], fronts[:2]))
                if len(gains) == 2:
                    gains = [(gains[0] * gains[1]).contiguous()]
                return rows + (gains or [None])
            rows = [self.batch._control_const(p, p.name) for p in self.ports[:2]]
            rows.append(self.batch._control_const(self.filt.cutoff, 'cutoff'))
            gains = [self.batch._control_const(p, p.name) for p in self.gain_ports]
        except NotBatchable:
            return None
        if len(gains) == 2:
            # one output gain: the product of the two rows, kept while both uploads are (so that its identity is as stable as
            # theirs: the closed form's constants are keyed on it)
            held = self.batch.owner._gain_products.get((id(gains[0]), id(gains[1])))
            if held is None or held[0] is not gains[0] or held[1] is not gains[1]:
                self.batch.owner._gain_products.clear()
                held = self.batch.owner._gain_products[(id(gains[0]), id(gains[1]))] = (gains[0], gains[1], (gains[0] * gains[1]).contiguous())
            gains = [held[2]]
        return rows + (gains or [None])

    def widths_ok(self, controls) -> bool:
        hertz, phase, cutoff, gain = controls
        v = self.channels
        source = [hertz, phase] + [t for t in (self.pair_rows or ()) if t is not None]
        return (max(t.shape[1] for t in source) == v and cutoff.shape[1] == v
                and all(t.shape[1] in (1, v) for t in source) and (gain is None or gain.shape[1] in (1, v)))

    def pair_arg(self):
        """the `pair` argument of _native.fused_rows: (op, second kind, hertz2, phase2, mix)"""
        if self.pair is None:
            return None
        return (type(self.pair[0]).__name__, self.pair[1].kind(), *self.pair_rows)

    def cycles_per_frame_bound(self) -> tuple[float, float]:
        """(max |hertz| / rate, max |phase|) over the chain's voices, from the host arrays behind the Fixed controls
        (an unplugged port is 0): bounds |t| = |frame / rate * hertz + phase| of any frame without touching the device"""
        import numpy as np
        out = []
        for p in (self.src.hertz, self.src.phase):
            src = p.sig
            if src is None or not src._state.enabled:
                out.append(0.0)
            else:
                value = np.abs(np.asarray(src._state.value, dtype=np.float64))
                out.append(float(value.max()) if value.size and np.isfinite(value).all() else float('inf'))
        return out[0] / self.batch.rate, out[1]

    def ordered_by_cutoff(self, ctl, pan, voices_per_lane: int):
        """(controls, pan) with the voices re-ordered for the bus launch: groups of 64 neighbours in cutoff, dealt round
        robin over the voice tiles of 64 * voices_per_lane, slot-major (slot i of every lane of a wave = one group).  The bus is a sum
        over voices, so any order renders the same bus up to the rounding of the sum; this one lets the Sine closed
        form drop the decayed homogeneous part of whole voice slots (fused_voice.hip: fused_steady_bus_kernel).
        Built once per parameter upload (it is kept with the closed form's constants)."""
        index = self.cutoff_order(ctl, voices_per_lane)
        if index is None:
            return ctl, pan
        v = self.channels
        pick = lambda t: t if t is None or t.shape[1] != v else t.index_select(1, index).contiguous()
        return [pick(t) for t in ctl], pick(pan)

    def cutoff_order(self, ctl, voices_per_lane: int):
        """the permutation of `ordered_by_cutoff` as a device index, or None when the cutoffs are not one Fixed row"""
        import numpy as np
        v, src = self.channels, self.filt.cutoff.sig
        if ctl[2].shape[1] != v or not isinstance(src, fixed.Fixed):
            return None
        cut = np.asarray(src._state.value, dtype=np.float64).reshape(-1)
        if cut.size != v or not np.isfinite(cut).all():
            return None
        order = np.argsort(cut, kind='stable')
        tile = 64 * voices_per_lane
        tiles = v // tile
        perm = order.copy()                                                  # a ragged last tile stays lane-major
        q = np.arange(tiles * tile)
        # the j-th group of 64 neighbours in cutoff goes to tile j % tiles, slot j // tiles: every wave gets the same mix
        # of slow- and fast-decaying slots (one wave per SIMD: the launch takes as long as its slowest wave)
        group, lane = q // 64, q % 64
        perm[((group % tiles) * 64 + lane) * voices_per_lane + group // tiles] = order[q]
        return torch.from_numpy(perm).to(runtime.device())

    def live_key(self, bus_node=None):
        """identities of the resident control tensors (and the bus gains) as they are NOW -- `resident()` re-uploads an
        edited array, which changes the identity; None if the pattern no longer holds.  The cheap per-block check of the
        latency path: a bound call is reused while every identity is unchanged."""
        for n in self.involved:
            if not n._state.enabled:
                return None
        key = []
        for p in self.ports:
            src = p.sig
            if src is None or not src._state.enabled:
                key.append(None)
            elif type(src) is fixed.Fixed:
                key.append(src.resident())
            else:
                return None
        if bus_node is not None:
            key.append(bus_node.resident_gains())
        return key

    def _same_shapes(self, now, then) -> bool:
        return now is not None and all(a.shape == b.shape for a, b in zip(now[:3], then[:3]))

    # ---- chain only: out (K*N, voices)
    def launch_chain(self, node: Emitter, controls, hist: int) -> torch.Tensor:
        b, o, dev = self.batch, self.batch.owner, runtime.device()
        N, K, rate, v = b.N, b.K, b.rate, self.channels
        rows = N * K
        status = o._status_word(self.filt)
        name = f'fused_osc_biquad[{self.tag}]'

        def run(position, ctl, out):
            if self.general:
                return o._launch(name, lambda: _native.fused_rows(self.kind, self.btype, rate, position, N, K, CONTEXT, v,
                                                                  ctl[0], ctl[1], ctl[2], ctl[3], out, status=status,
                                                                  pair=self.pair_arg(), hertz_hist=self.hist_rows[0],
                                                                  phase_hist=self.hist_rows[1]),
                                 units=rows * v)
            return o._launch(name, lambda: _native.fused_osc_biquad(self.kind, self.btype, rate, position, N, K, CONTEXT,
                                                                    ctl[0], ctl[1], ctl[2], ctl[3], out, status=status),
                             units=rows * v)
        if hist:
            # a consumer (a second filter) needs history rows: the launch writes the K blocks, the rows in front
            # come from the tail / a fresh block like any filter's
            result = torch.empty((hist + rows, v), dtype=AUDIO_DTYPE, device=dev)
            run(b.pos, controls, result[hist:])
            b._own_history(node, v, hist, result)
            return result

        def replay(position: int) -> torch.Tensor:
            ctl = self.resolve()
            if not self._same_shapes(ctl, controls):
                o._replay = None
                return o.render(position, N, K)                  # pattern no longer holds: re-plan
            return run(position, ctl, torch.empty((rows, v), dtype=AUDIO_DTYPE, device=dev))
        if node is o.node and not self.modulated:                      # (a replay would re-read the control rows at a stale position)
            o._remember_replay(N, K, replay)
        return run(b.pos, controls, torch.empty((rows, v), dtype=AUDIO_DTYPE, device=dev))

    # ---- chain + MixMatrix: out (K*N, voices), the per-voice rows only ever exist as 32-row LDS tiles
    def launch_mix(self, mix_node, controls) -> torch.Tensor:
        b, o = self.batch, self.batch.owner
        N, K, v = b.N, b.K, self.channels
        if self.general:
            return None                                             # per-block rows / two oscillators: the chain runs fused, the matrix as its own launch
        out = torch.empty((N * K, v), dtype=AUDIO_DTYPE, device=runtime.device())
        matrix, status = mix_node.resident_matrix(), o._status_word(self.filt)
        return o._launch(f'fused_osc_biquad_mix[{self.tag}]',
                         lambda: _native.fused_osc_biquad_mix(self.kind, self.btype, b.rate, b.pos, N, K, CONTEXT,
                                                              controls[0], controls[1], controls[2], controls[3], matrix, out,
                                                              status=status),
                         units=N * K * v)

    # ---- chain + bus: out (K*N, bus channels)
    def launch_bus(self, node: Emitter, controls) -> torch.Tensor:
        b, o, dev = self.batch, self.batch.owner, runtime.device()
        N, K, rate, v = b.N, b.K, b.rate, self.channels
        rows, bus_c = N * K, self.bus_node.channels
        pan = self.bus_node.resident_gains()
        if pan is not None and pan.shape[1] != v:
            return None
        status = o._status_word(self.filt)
        need = _native.lib().sig_fused_voice_bus_workspace(v, rows, bus_c) // 8
        if o._workspace is None or o._workspace.numel() < need:
            o._workspace = torch.empty(need, dtype=CTRL_DTYPE, device=dev)
        if self.general:
            if bus_c not in (1, 2):
                return None
            swept = controls[2].shape[0] > 1                                    # the cutoff is read per block
            if (self.kind == 'Sine' and self.pair is None and not self.fm and controls[2].shape[1] == v
                    and (swept or (controls[3] is not None and controls[3].shape[0] > 1 and controls[3].shape[1] == v))):
                # a swept cutoff and / or a tremolo: sig_fused_voice_bus_rows keeps the closed form (per-block filter constants,
                # bus weights rebuilt per block), and the closed form wants its voices ordered by cutoff (ordered_by_cutoff: it
                # drops the decayed homogeneous part per voice slot); the order is kept while the constant rows are.  Under a
                # sweep the order of the batch's first row stands for all of them (an LFO scales every voice's cutoff alike;
                # any order is correct) -- sorted on the device, no host round trip
                held = o._tremolo_order
                key = (id(controls[0]), id(controls[1]), id(controls[2]) if not swept else None, id(pan), K, swept)
                if held is None or held[0] != key:
                    plan = _native.fused_voice_bus_plan('Sine', b.pos, v, N, K, CONTEXT)
                    vpl = min(plan['voices_per_lane'], 8)
                    if swept and vpl < 8:
                        vpl = min(vpl, 2)                                        # (per-block constants: eight voices per lane, or two)
                    if not plan['closed_form']:
                        index = None
                    elif not swept:
                        index = self.cutoff_order(controls, vpl)
                    else:
                        tile, tiles = 64 * vpl, v // (64 * vpl)
                        order = torch.argsort(controls[2][0], stable=True)
                        index = order.clone()                                    # (a ragged last tile stays lane-major)
                        if tiles:
                            q = torch.arange(tiles * tile, device=dev)
                            group, lane = q // 64, q % 64
                            index[((group % tiles) * 64 + lane) * vpl + group // tiles] = order[q]
                    pick = lambda t: t if t is None or index is None or t.shape[1] != v else t.index_select(1, index).contiguous()
                    held = o._tremolo_order = (key, (controls[0], controls[1], controls[2], pan), index,
                                               [pick(controls[0]), pick(controls[1]), None if swept else pick(controls[2])], pick(pan))
                if held[2] is not None:
                    pick = lambda t: t if t is None or t.shape[1] != v else t.index_select(1, held[2])
                    cut = pick(controls[2]) if swept else held[3][2]
                    gain_rows = controls[3]
                    if gain_rows is not None and gain_rows.shape[1] == v:
                        gain_rows = pick(gain_rows)
                    controls = [held[3][0], held[3][1], cut, gain_rows]
                    pan = held[4]
            out = torch.empty((rows, bus_c), dtype=AUDIO_DTYPE, device=dev)
            return o._launch(f'fused_voice_bus[{self.tag}]',
                             lambda: _native.fused_rows(self.kind, self.btype, rate, b.pos, N, K, CONTEXT, v, controls[0], controls[1],
                                                        controls[2], controls[3], out, bus_gains=pan, bus=True,
                                                        workspace=o._workspace, status=status, pair=self.pair_arg(),
                                                        hertz_hist=self.hist_rows[0], phase_hist=self.hist_rows[1]),
                             units=rows * v)
        # latency regime: too few (voice, block) chains to fill the chip with serial walks -> the chain runs as a
        # time-parallel prefix scan (sig_fused_osc_biquad picks it) and the bus as its own launch
        small = v * K <= o.scan_max_chains and CONTEXT + N <= SCAN_MAX_ROWS
        chain_name, bus_name = f'fused_osc_biquad[{self.tag}]', f'fused_voice_bus[{self.tag}]'
        # one block of a Sine chain: a single launch does chain, bus and (under hipGraph replay) the position advance
        one_launch = small and K == 1 and self.kind == 'Sine' and o.latency_kernel
        if one_launch and (o._latency_ws is None or o._latency_ws[0] != (v, N, bus_c)):
            o._latency_ws = ((v, N, bus_c), _native.latency_voice_bus_workspace(v, N, bus_c, dev))

        def run(position, ctl, pan_now, out):
            if one_launch:
                return o._launch(f'latency_voice_bus[{self.tag}]',
                                 lambda: _native.latency_voice_bus(self.btype, rate, position, N, CONTEXT, v, ctl[0], ctl[1],
                                                                   ctl[2], ctl[3], pan_now, out, o._latency_ws[1],
                                                                   status=status), units=rows * v)
            if small:
                voices_buf = torch.empty((rows, v), dtype=AUDIO_DTYPE, device=dev)
                o._launch(chain_name, lambda: _native.fused_osc_biquad(self.kind, self.btype, rate, position, N, K, CONTEXT,
                                                                       ctl[0], ctl[1], ctl[2], ctl[3], voices_buf,
                                                                       status=status), units=rows * v)
                return o._launch('sum_bus', lambda: _native.sum_bus(voices_buf, pan_now, out), units=rows * v)
            # the Sine closed form's per-voice constants survive from call to call while the control tensors (held here,
            # so their addresses cannot be recycled), the filter type and min(context, position) are the same
            key = (tuple(id(t) for t in ctl), id(pan_now), self.btype, rate, v, min(CONTEXT, position))
            held = o._steady_consts
            ready = held is not None and held[0] == key
            if not ready:
                size = _native.lib().sig_fused_voice_consts_size(v) // 8
                buf = held[2] if held is not None and held[2].numel() >= size else torch.empty(size, dtype=CTRL_DTYPE, device=dev)
                plan = _native.fused_voice_bus_plan(self.kind, position, v, N, K, CONTEXT)
                o._steady_consts = held = (key, (tuple(ctl), pan_now), buf, self.cycles_per_frame_bound(),
                                           self.ordered_by_cutoff(ctl, pan_now, plan['voices_per_lane']))
            ctl, pan_now = held[4]
            # Past |t| = 2^26 cycles the Sine closed form (and the walker's incremental phase) hands over to the exact
            # per-row phase, wave by wave inside the launch -- correct, but the closed-form kernel's built-in fallback is
            # a plain loop meant for a few waves.  max |hertz| and max |phase| are host-side knowledge: when the launch's
            # last frame puts the fastest voice past the limit, the span walker takes the whole launch instead.
            per_frame, ph_max = held[3]
            walk = self.kind == 'Sine' and ph_max + (position + rows) * per_frame >= _native.SINE_FAST_MAX_CYCLES
            if o.timer is None or o.timer.region:
                # the per-batch host path: one pre-bound ctypes call (the binding's per-call validation cost more than the launch at
                # 256 blocks per batch); bound per set of constants, i.e. while the parameter uploads are the same tensors
                call = held[5] if len(held) > 5 else None
                if call is None or call.shape != (rows, bus_c) or call._keep[5] is not o._workspace:
                    call = _native.FusedVoiceBusCall(self.kind, self.btype, rate, N, K, CONTEXT, v, ctl[0], ctl[1], ctl[2], ctl[3], pan_now,
                                                     bus_c, o._workspace, status, held[2])
                    o._steady_consts = held = (*held[:5], call)
                if o.timer is not None:
                    return o.timer.launch(bus_name, lambda: call(position, out, ready, walk), units=rows * v)
                return call(position, out, ready, walk)
            return o._launch(bus_name, lambda: _native.fused_voice_bus(self.kind, self.btype, rate, position, N, K, CONTEXT, v,
                                                                       ctl[0], ctl[1], ctl[2], ctl[3], pan_now, out,
                                                                       workspace=o._workspace, status=status,
                                                                       consts=held[2], consts_ready=ready, walk=walk),
                             units=rows * v)

        def captured(position, ctl, pan_now):
            """hipGraph of [scan chain(position on device) -> bus -> position += N*K]; None if capture is refused"""
            keys = tuple(t.data_ptr() if t is not None else 0 for t in (*ctl, pan_now))
            cap = o._captured
            if cap is None or cap.keys != keys:
                def record(pos_t: torch.Tensor) -> torch.Tensor:
                    bus_out = torch.empty((rows, bus_c), dtype=AUDIO_DTYPE, device=dev)
                    if one_launch:
                        return _native.latency_voice_bus(self.btype, rate, pos_t, N, CONTEXT, v, ctl[0], ctl[1], ctl[2], ctl[3],
                                                         pan_now, bus_out, o._latency_ws[1], status=status)
                    vbuf = torch.empty((rows, v), dtype=AUDIO_DTYPE, device=dev)
                    _native.fused_osc_biquad(self.kind, self.btype, rate, pos_t, N, K, CONTEXT,
                                             ctl[0], ctl[1], ctl[2], ctl[3], vbuf, status=status)
                    return _native.sum_bus(vbuf, pan_now, bus_out)
                try:
                    cap = o._captured = _CapturedLaunches(record, N * K, position, keys, self_advancing=one_launch)
                except RuntimeError:
                    # capture refused (another capture in progress, a profiler that forbids it ...):
                    # keep rendering with plain launches
                    o.graph_replay, o._captured = False, None
                    return None
            return cap.run(position)

        bound = [None]                                       # (live key, LatencyVoiceBusCall) of the one-launch block

        RING = 4

        def pipelined(position: int):
            """the batch at `position` on the next of `o.pipeline` streams, or None when the fast path does not apply (first
            call, parameters re-uploaded, a position inside the first context, the walker's range): the caller falls through"""
            key = self.live_key(self.bus_node)
            held, pipe = o._steady_consts, o._pipe
            if key is None or held is None or len(held) < 5 or held[0][-1] != CONTEXT or position < CONTEXT:
                return None
            if pipe is None or pipe['held'] is not held or len(pipe['key']) != len(key) or not all(a is b for a, b in zip(key, pipe['key'])) \
                    or pipe['shape'] != (rows, bus_c):
                # set-up (once per set of parameter uploads): the constants are in held[2] (made by the plain path on the caller's
                # stream: the device is synchronised once here so that the side streams may read them and the ordered rows)
                held_ctl, held_pan = held[1]
                if len(key) != len(held_ctl) + 1 or key[-1] is not held_pan or \
                        not all(k is None or k is t for k, t in zip(key, held_ctl)):
                    return None                                                # (the constants belong to other uploads: the plain path renews them first)
                torch.cuda.synchronize()
                ctl_o, pan_o = held[4]
                ws_size = _native.lib().sig_fused_voice_bus_workspace(v, rows, bus_c) // 8
                streams = [torch.cuda.Stream() for _ in range(o.pipeline)]
                work = [torch.empty(ws_size, dtype=CTRL_DTYPE, device=dev) for _ in streams]
                pipe = o._pipe = {
                    'held': held, 'key': key, 'shape': (rows, bus_c), 'i': 0, 'streams': streams, 'work': work,
                    'handles': [s_.cuda_stream for s_ in streams],
                    'events': [torch.cuda.Event() for _ in streams],
                    'calls': [_native.FusedVoiceBusCall(self.kind, self.btype, rate, N, K, CONTEXT, v, ctl_o[0], ctl_o[1], ctl_o[2], ctl_o[3],
                                                        pan_o, bus_c, w, status, held[2]) for w in work],
                    'outs': [[torch.empty((rows, bus_c), dtype=AUDIO_DTYPE, device=dev) for _ in range(RING)] for _ in streams],
                    'slot': [0] * len(streams)}
                torch.cuda.synchronize()
            per_frame, ph_max = held[3]
            if self.kind == 'Sine' and ph_max + (position + rows) * per_frame >= _native.SINE_FAST_MAX_CYCLES:
                return None                                                    # (beyond the closed form's range: the plain path picks the walker)
            i = pipe['i']
            pipe['i'] = (i + 1) % len(pipe['streams'])
            slot = pipe['slot'][i]
            pipe['slot'][i] = (slot + 1) % RING
            out = pipe['outs'][i][slot]
            call, handle = pipe['calls'][i], pipe['handles'][i]
            if o.timer is not None:
                o.timer.launch(bus_name, lambda: call(position, out, True, False, stream=handle), units=rows * v)
            else:
                call(position, out, True, False, stream=handle)
            pipe['events'][i].record(pipe['streams'][i])                     # the caller's stream waits for this batch
            raw = _native.current_stream_handle(dev.index)                     # (torch.cuda.current_stream() costs 8 us of Python: the Stream
            cur = pipe.get('current')                                          # object is kept while the raw handle is the same)
            if cur is None or cur[0] != raw:
                cur = pipe['current'] = (raw, torch.cuda.current_stream())
            cur[1].wait_event(pipe['events'][i])
            return out

        def replay(position: int) -> torch.Tensor:
            if o.pipeline > 1 and not small and (o.timer is None or o.timer.region):
                done = pipelined(position)
                if done is not None:
                    return done
            if one_launch and o.timer is None:
                # latency mode, the per-block host path: five identity checks, one allocation, one ctypes call (a hipGraph of
                # this single launch would only add its replay cost: 19.6 us per block against 14.6)
                key, held = self.live_key(self.bus_node), bound[0]
                if key is not None and held is not None and len(key) == len(held[0]) and all(a is b for a, b in zip(key, held[0])):
                    return held[1](position, torch.empty((rows, bus_c), dtype=AUDIO_DTYPE, device=dev))
            ctl, pan_now = self.resolve(), self.bus_node.resident_gains()
            if not self._same_shapes(ctl, controls) or (pan_now is None) != (pan is None):
                o._replay = None
                return o.render(position, N, K)                  # pattern no longer holds: re-plan
            if small and o.graph_replay and not one_launch:
                out = captured(position, ctl, pan_now)
                if out is not None:
                    return out
            if one_launch and o.timer is None:
                key = self.live_key(self.bus_node)
                if key is not None and (pan_now is None or pan_now.shape[1] == v) and self.widths_ok(ctl):
                    bound[0] = (key, _native.LatencyVoiceBusCall(self.btype, rate, N, CONTEXT, v, ctl[0], ctl[1], ctl[2], ctl[3],
                                                                 pan_now, bus_c, o._latency_ws[1], status))
            return run(position, ctl, pan_now, torch.empty((rows, bus_c), dtype=AUDIO_DTYPE, device=dev))
        if node is o.node:
            o._remember_replay(N, K, replay)
        return run(b.pos, controls, pan, torch.empty((rows, bus_c), dtype=AUDIO_DTYPE, device=dev))


class _NoProgram(Exception):
    """this graph is not one voice program"""


class _ProgramRows:
    """The per-block rows of a voice program's computed control ports, laid out as sig_voice_program takes them, each port one
    (control_rows, cols) buffer filled by block-rate control programs (sig_control_program) that write straight into it:
      blocks >= context:  [the block in front | H history blocks | K blocks]   -- the K rows and the row in front of them in one
                          launch, one more launch of one row per further row in front;
      blocks <  context:  [K virtual blocks, controls at max(p - context, 0) | K blocks]; the second group evaluated at the
                          blocks' own positions (`inner` False) or where the oldest cached reply containing the block was
                          evaluated (`inner`: the ports in front of the voice's last filter; header of sig_voice_program)."""

    def __init__(self, srcs: tuple, K: int, lead: int, small: bool):
        self.srcs, self.K, self.lead, self.small = srcs, K, lead, small
        if not small:
            self.main = _ControlProgram(srcs, K, lead=lead)
            self.full = self.main.full
            self.front = [_ControlProgram(srcs, 1, into={i: t[j:j + 1] for i, t in self.full.items()}) for j in range(lead - 1)]
        else:
            probe = _ControlProgram(srcs, K, lead=K)                           # rows [K:] of (2 K, cols) buffers: the blocks themselves
            self.main, self.full = probe, probe.full
            self.virtual = _ControlProgram(srcs, K, into={i: t[:K] for i, t in self.full.items()})
            self.first = _ControlProgram(srcs, 1, into={i: t[K:K + 1] for i, t in self.full.items()})

    def current(self) -> bool:
        return self.main.current()

    def tensors(self) -> list:
        """per source: its rows -- the filled buffer, or the resident (1, cols) row of a Fixed / an unplugged port"""
        return [self.full.get(i, r if r is not None else as_control(src.resident()))
                for i, (src, r) in enumerate(zip(self.srcs, self.main.results))]


class _VoiceProgram:
    """The per-voice graph under a node as ONE launch of sig_voice_program (voice_program.hip): oscillators, LowPass / HighPass,
    Gain / Amp / Mix / RingMod, Fixed rows, ADSR, White, in any arrangement in which every voice is computed from its own
    parameters only (nothing mixes channels in front of the sink) and no inner node has a reader outside the graph.  Compiled
    here into straight-line code for the kernel's accumulator machine: a binary node parks its left operand in a temporary, a
    node with several readers is computed once and kept in one.  Control ports driven by computed block-rate signals become
    per-block rows (`_ProgramRows`).  The block history stays implicit: the launch re-walks the blocks in
    front of it from where the reference cold-started them, so no tails are kept for what it covers."""

    KERNEL_NODES = (osc.Osc, fx.SingleCritFilter, fx.Gain, fx.Amp, fx.Mix, fx.RingMod, ext.ADSR, noise.White)

    def __init__(self, batch: '_Batch', top: Emitter, voices: int):
        self.batch, self.top, self.voices = batch, top, voices
        self.code: list = []
        self.oscs: list = []                     # (hertz control index, phase control index | None)
        self.params: list = []                   # control index per parameter register
        self.filters: list = []                  # (cutoff control index, type, level, node)
        self.controls: list = []                 # (port | None, constant tensor | None, filters between the node and the sink)
        self.adsr = None
        self.seeds: list = []
        self.temps_used, self.temps_free, self.n_temps = set(), [], 0
        self.saved: dict = {}                    # node -> [temporary, readers left]
        self.depth_of: dict = {}
        self.uses: dict = {}
        self.kernel_nodes = 0
        self._count(top)
        for n, uses in self.uses.items():
            if n is not top and len(n.outputs_with_ports) != uses:
                raise _NoProgram(f'{n.cls_name()} has a reader outside the graph')
        self.depth = self._emit(top, 0)
        if len(self.code) > _native.VP_MAX_INS:
            raise _NoProgram('program too long')

    # ---- pass 1: readers of every node inside the graph
    def _count(self, n):
        if n is None or not n.get_state().enabled:
            if isinstance(n, ext.Tap):
                self._count(n.input.sig)
            return
        if isinstance(n, ext.Tap):
            return self._count(n.input.sig)
        self.uses[n] = self.uses.get(n, 0) + 1
        if self.uses[n] > 1:
            return
        if isinstance(n, self.KERNEL_NODES):
            self.kernel_nodes += 1
        elif not isinstance(n, fixed.Fixed):
            raise _NoProgram(f'no voice-program instruction for {n.cls_name()}')
        for port in _audio_ports(n):
            self._count(port.sig)

    # ---- pass 2: code
    def _control(self, port, below: int, optional: bool = False):
        src = port.sig
        if src is None or not src.get_state().enabled:
            if optional:
                return None
            self.controls.append((None, Emitter.empty_result(), below))
        else:
            self.controls.append((port, None, below))
        return len(self.controls) - 1

    def _param(self, index: int) -> int:
        if len(self.params) >= _native.VP_MAX_PARAMS:
            raise _NoProgram('more parameter registers than the machine has')
        self.params.append(index)
        return len(self.params) - 1

    def _temp(self) -> int:
        if self.temps_free:
            t = self.temps_free.pop()
        else:
            t = self.n_temps
            self.n_temps += 1
            if self.n_temps > _native.VP_MAX_TEMPS:
                raise _NoProgram('more temporaries than the machine has')
        return t

    def _zero(self, below: int) -> int:
        self.controls.append((None, Emitter.empty_result(), below))
        self.code.append(('Const', 0, self._param(len(self.controls) - 1), 0, 0))
        return 0

    def _emit(self, n, below: int) -> int:
        """code that leaves the node's sample in the accumulator; returns the filters in series up to and including it"""
        if isinstance(n, ext.Tap):
            return self._emit(n.input.sig, below)                              # a pass-through, enabled or not
        if n is None or not n.get_state().enabled:
            return self._zero(below)                                           # zeros((1, 1)) (chain/__init__.py:250-254, :297-298)
        if n in self.saved:
            slot = self.saved[n]
            self.code.append(('Load', 0, slot[0], 0, 0))
            slot[1] -= 1
            if slot[1] == 0:
                self.temps_free.append(slot[0])
                del self.saved[n]
            return self.depth_of[n]
        if isinstance(n, fixed.Fixed):
            row = n.resident()
            if row.shape[0] != 1:
                raise _NoProgram('multi-row Fixed as an audio source')
            self.controls.append((None, as_control(row), below))
            self.code.append(('Const', 0, self._param(len(self.controls) - 1), 0, 0))
            depth = 0
        elif isinstance(n, osc.Osc):
            if len(self.oscs) >= _native.VP_MAX_OSCS:
                raise _NoProgram('more oscillators than the machine has slots')
            self.oscs.append((self._control(n.hertz, below), self._control(n.phase, below, optional=True)))
            self.code.append(('Osc', _native.OSC_KINDS[n.kind()], len(self.oscs) - 1, 0, 0))
            depth = 0
        elif isinstance(n, noise.White):
            if len(self.seeds) >= 2 or n.channels != self.voices:
                raise _NoProgram('White: two per program, as wide as the voices')
            self.seeds.append(int(n.get_state().seed))
            self.code.append(('Noise', 0, len(self.seeds) - 1, 0, 0))
            depth = 0
        elif isinstance(n, ext.ADSR):
            if (self.adsr is not None and self.adsr is not n) or _modulated(n):
                raise _NoProgram('one block-invariant envelope per program')
            self.adsr = n
            self.code.append(('Adsr', 0, 0, 0, 0))
            depth = 0
        elif isinstance(n, (fx.Gain, fx.Amp)):
            depth = self._emit(n.left.sig, below)
            self.code.append(('Gain' if isinstance(n, fx.Gain) else 'Amp', 0, self._param(self._control(n.right, below)), 0, 0))
        elif isinstance(n, (fx.Mix, fx.RingMod)):
            left = self._emit(n.left.sig, below)
            kept = self.saved.get(n.right.sig) if n.right.sig is not n.left.sig else None
            if kept is not None:
                # the right operand is a node with several readers that already sits in a temporary: combine straight from it
                t, right, swapped = kept[0], self.depth_of[n.right.sig], 1                    # (the accumulator is the LEFT operand)
                kept[1] -= 1
            else:
                t, swapped = self._temp(), 0
                self.code.append(('Save', 0, t, 0, 0))
                right = self._emit(n.right.sig, below)
            if isinstance(n, fx.Mix):
                self.code.append(('Mix', 0, t, self._param(self._control(n.mix, below)), swapped))     # m L + (1 - m) R  (fx.py:40)
            else:
                self.code.append(('Mul', 0, t, 0, 0))
            if kept is None:
                self.temps_free.append(t)
            elif kept[1] == 0:
                self.temps_free.append(t)
                del self.saved[n.right.sig]
            depth = max(left, right)
        elif isinstance(n, fx.SingleCritFilter):
            src = n.input.sig
            if src is None or not src.get_state().enabled:
                raise _NoProgram('filter without an input')                    # (the per-node schedule raises the reference's error)
            depth = self._emit(src, below + 1) + 1
            if len(self.filters) >= _native.VP_MAX_FILTERS:
                raise _NoProgram('more filters than the machine has slots')
            self.filters.append((self._control(n.cutoff, below), str(n.type()), depth, n))
            self.code.append(('Filter', 0, len(self.filters) - 1, 0, 0))
        else:
            raise _NoProgram(f'no voice-program instruction for {n.cls_name()}')
        self.depth_of[n] = depth
        if self.uses.get(n, 1) > 1:                                            # several readers: computed once, kept in a temporary
            t = self._temp()
            self.code.append(('Save', 0, t, 0, 0))
            self.saved[n] = [t, self.uses[n] - 1]
        return depth

    # ---- control rows and the launch
    def _rows(self):
        """(per control: its row tensor, control_rows, history block starts, blocks rendered before) or None"""
        b, o = self.batch, self.batch.owner
        N, K, pos = b.N, b.K, b.pos
        small = self.depth > 0 and N < CONTEXT
        if small and (self.depth > 2 or N < 16):
            raise NotBatchable('short blocks: two filters in series at most, 16 frames at least')
        want = max(self.depth - 1, 0)
        if small:
            hist, front, lead = [], 0, K
        else:
            starts = o.history_starts(pos, want + 1)
            hist = starts[len(starts) - want:] if want else []
            front = starts[len(starts) - want - 1] if len(starts) > want else 0
            lead = len(hist) + 1
        control_rows = 2 * K if small else lead + K
        tensors: list = [None] * len(self.controls)
        groups: dict = {}
        for i, (port, const, below) in enumerate(self.controls):
            if const is not None:
                tensors[i] = const
            elif _ctl_const(port):
                tensors[i] = b._control_const(port, port.name)
            else:
                if small and below >= 2:
                    raise NotBatchable('short blocks: block-rate control in front of two filters in series')
                groups.setdefault(bool(small and below >= 1), []).append(i)
        for inner, members in groups.items():
            srcs = tuple(self.controls[i][0].sig for i in members)
            key = ('voice-program', tuple(id(x) for x in srcs), K, lead, small)
            held = o._ctl_programs.get(key)
            if held is None or held[0] != graph_clock.version or not held[1].current():
                if len(o._ctl_programs) > 16:
                    o._ctl_programs.clear()
                held = o._ctl_programs[key] = (graph_clock.version, _ProgramRows(srcs, K, lead, small))
            rows = held[1]
            if not small:
                ahead = [front] + hist                                         # where the rows in front of the K blocks are read
                rows.main.run(o, b.rate, pos, N, front_position=ahead[lead - 1])       # (the last of them in the same launch)
                for j, sub in enumerate(rows.front):
                    sub.run(o, b.rate, ahead[j], 0)
            else:
                rows.virtual.run(o, b.rate, pos - CONTEXT, N)
                if not inner:
                    rows.main.run(o, b.rate, pos, N)
                else:
                    # what feeds the last filter over block b was evaluated m_b = min((100 - N) / N, blocks before it - 1) blocks
                    # earlier (the oldest cached `after` reply containing the block), never before the stream's second block
                    mmax = (CONTEXT - N) // N
                    first = pos - o._stream_blocks * N
                    rows.main.run(o, b.rate, pos - mmax * N, N, min_position=first + N)
                    if o._stream_blocks == 0:
                        rows.first.run(o, b.rate, pos, 0)                      # ... the stream's very first block at its own position
            for i, t in zip(members, rows.tensors()):
                tensors[i] = as_control(t)
        return tensors, control_rows, hist, (o._stream_blocks if small else 0)

    def launch(self, out: torch.Tensor, bus_gains: torch.Tensor | None, bus: bool, label: str) -> torch.Tensor:
        b, o = self.batch, self.batch.owner
        tensors, control_rows, hist, before = self._rows()
        v = self.voices
        widths = [t.shape[1] for t in tensors]
        if any(w not in (1, v) for w in widths):
            raise _NoProgram('control rows of another width than the voices')
        for cut, _, _, _ in self.filters:
            if tensors[cut].shape[1] != v and v != 1:
                raise IndexError(f'index {tensors[cut].shape[1]} is out of bounds for axis 1 with size {tensors[cut].shape[1]}')   # fx.py:99
        adsr = None
        if self.adsr is not None:
            adsr = self.adsr.control_rows(lambda bound: b._control_const(bound, bound.name))
            if any(r.shape[1] not in (1, v) for r in adsr.values()):
                raise _NoProgram('envelope rows of another width than the voices')
        oscs = [(tensors[h], tensors[p] if p is not None else None) for h, p in self.oscs]
        params = [tensors[i] for i in self.params]
        filters = [(tensors[c], t, level) for c, t, level, _ in self.filters]
        status = o._status_word(self.filters[0][3]) if self.filters else None
        if bus:
            need = _native.lib().sig_fused_voice_bus_workspace(v, out.shape[0], out.shape[1]) // 8
            if o._workspace is None or o._workspace.numel() < need:
                o._workspace = torch.empty(need, dtype=CTRL_DTYPE, device=runtime.device())
        seeds = tuple(self.seeds + [0, 0])[:2]
        if self.depth:
            o._virtual_history.add(self.top)
        if o.specialise:
            from . import specialise
            C = out.shape[1] if bus else 0
            aligned = (4 if v % 4 == 0 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0 else
                       2 if v % 2 == 0 and out.stride(0) % 2 == 0 and out.data_ptr() % 8 == 0 else 1)
            # four voices per lane (one wave per SIMD) pay for programs without filter state or temporaries -- an oscillator bank
            # under a bus: 176 -> 144 us per 268 M voice-samples; with two filters 356 -> 396 us, with a temporary 380 -> 899 us.
            # The launch takes four only with such an image attached, so the choice is made here
            for four in ((True, False) if not filters and self.n_temps == 0 else (False,)):
                vpl, _ = _native.voice_program_geometry(v, b.N, b.K, CONTEXT, self.depth, C, aligned, specialised=four)
                make = specialise.ensure_in_background if o.specialise == 'background' else specialise.ensure
                if make(self.code, len(oscs), len(params), len(filters), self.n_temps, vpl, C):
                    label += '*specialised'
                    break
        return o._launch(label, lambda: _native.voice_program(self.code, oscs, params, filters, self.n_temps, self.depth, b.rate, b.pos,
                                                              b.N, b.K, CONTEXT, v, control_rows, hist, out, bus_gains=bus_gains, bus=bus,
                                                              adsr=adsr, noise_seeds=seeds, workspace=o._workspace if bus else None,
                                                              status=status, blocks_before=before),
                         units=out.shape[0] * v)

    @classmethod
    def compile(cls, batch: '_Batch', top: Emitter, voices: int, min_nodes: int = 2):
        """the program, or None when the graph is not one (or is a single kernel anyway: under a bus one node is enough, the
        bus being the second -- `min_nodes`)"""
        if top is None or not top.get_state().enabled:
            return None
        try:
            prog = cls(batch, top, voices)
        except _NoProgram:
            return None
        return prog if prog.kernel_nodes >= min_nodes else None

    def describe(self) -> str:
        return ','.join(op for op, *_ in self.code)

    def worthwhile(self) -> bool:
        """Does the interpreted launch beat one kernel per node?  Measured (tools/time_voice_program.py, 1024 voices): programs
        that fit the interpreter's SMALL register file -- two filters, three oscillators, four parameter registers, one
        temporary, no Amp / ADSR / White -- run at two waves per SIMD, 0.36-0.8 T voice-samples/s against 0.2-0.26 T per node;
        the full register file runs at one wave per SIMD and loses (0.14 T for three filters in series; f64 pow dominates an
        Amp either way).  Blocks shorter than the filter context have no per-node schedule at all (the alternative is the eager
        pull path, ~150 us per block)."""
        small_file = (len(self.filters) <= 2 and len(self.oscs) <= 3 and len(self.params) <= 4 and self.n_temps <= 1
                      and self.adsr is None and not self.seeds and not any(op == 'Amp' for op, *_ in self.code))
        if self.batch.owner.specialise is True:         # ('background': the interpreter renders meanwhile, so its policy decides)
            # a kernel built for this program has no interpreter to pay for: three filters in series 0.54 T against 0.26 per node,
            # an Amp behind a filter level with it (f64 pow either way)
            from . import specialise
            if specialise.hipcc() is not None:
                return True
        return small_file or (self.depth > 0 and self.batch.N < CONTEXT)


_KNOWN_TYPES = tuple(t for types, _ in _Batch._SCHEDULES for t in (types if isinstance(types, tuple) else (types,)))
