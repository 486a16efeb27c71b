#!/usr/bin/env python3
"""Headline benchmark: rendered Msamples/s (voice-samples per wall second), 48 kHz, 256-frame blocks.

Workload (BASELINE.json configs[1]): per GPU a 1024-voice
`Fixed -> Sine -> LowPass -> Gain -> SumBus(stereo)` graph built through the node API and rendered by
the batched engine, one step = one batch of `--blocks` (default 4096) consecutive 256-frame blocks of synthetic
parameters (numpy default_rng(0): hertz U(55,1760), phase U(0,1), cutoff U(200,8000), gain U(0,1)/V,
pan theta U(0,pi/2)), already resident in HBM.  With N > 1 GPUs every rank renders its own 1024 voices
(weak scaling, no data-path traffic) and the stereo bus is summed onto rank 0 with one RCCL reduce per
batch (the path's only exchange step), overlapped with the next batch's kernels.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks B] [--voices V] [--frames F]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with
  * `roofline`: the dominant kernel of the timed steps against the ceiling that bounds it (HIP events around every
    launch, on the launch stream) -- the f64 vector issue rate for the fused voice kernels (their HBM view nested
    under `hbm`), HBM bandwidth for the per-node kernels;
  * `max_abs_error`: blocks {0..7, middle, last} of the LAST TIMED batch's own output against the CPU oracle, with
    the name of the device kernel that produced them (BASELINE.json's second metric);
  * `sustained`: the same steps repeated for >= 2 s of wall time (clocks settled, visible to a busy sampler);
  * `configs`: BASELINE.json's other single-GPU configurations (C3, C5), each with its own roofline and error;
  * `cpu_baseline`: the CPU oracle, structured like the reference (per-channel butter + sosfilt per block), timed on
    a bounded sample.
"""
import argparse
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch
import torch.distributed as dist

import bench_configs as cfg

RATE = cfg.RATE
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
F64_VALU_PEAK = 39.3            # T f64 instruction-lanes/s: 256 CU x 4 SIMD x 16 f64 lanes/clk x 2.4 GHz (= 78.6 TFLOP/s as FMAs)
# algorithmic HBM bytes per voice-sample, f32 storage, every node output written once and read once
# per consumer (C2 = 24 B over four kernels, C3 = 40 B, C5's matrix node 8 B)
ALGO_BYTES = {'osc_bank': 4, 'biquad_coldstart': 8, 'elementwise': 8, 'sum_bus': 4, 'adsr': 4, 'adsr_apply': 8,
              'biquad_bus': 4, 'mix_matrix': 8,
              'fused_osc_biquad': 4,          # fused chain: only the f32 store reaches HBM
              'fused_osc_biquad_mix': 4,      # chain + matrix in one launch: only the mixed f32 rows are stored
              'fused_cascade_bus': None,      # cascade + envelope + bus: f64 tile partials + the bus (like fused_voice_bus)
              'fused_voice_bus': None}        # f64 tile partials (written, re-read) + the f32 bus: depends on the voices per lane
VALU_BOUND = ('fused_voice_bus', 'fused_cascade_bus')      # kernels whose ceiling is the f64 vector issue rate, not HBM

synth_params = cfg.c2_params
build_graph = cfg.c2_graph


def steady_applies(p, lo, hi, first_frame, last_frame, N, ctx=100):
    """host mirror of fused_voice.hip:steady_voice_ok for every voice of the shard: does sig_fused_voice_bus run its
    closed-form Sine kernel (steady-state sinusoid + homogeneous transient) rather than the row-by-row walker?"""
    hz, ph = p['hertz'][0, lo:hi], p['phase'][0, lo:hi]
    d = hz / RATE
    dr = d - np.rint(d)
    t = np.abs(np.stack([first_frame / RATE * hz + ph, last_frame / RATE * hz + ph]))
    return bool((t < 2.0 ** 26).all() and (np.abs(dr) <= 0.25).all() and (np.abs(np.sin(2 * np.pi * dr)) >= 1e-3).all()
                and (N >= ctx or first_frame >= ctx))


def closed_form_live_fraction(p, lo, hi, N, vpt, ctx=100, bus_channels=2, tol=1e-9):
    """Host mirror of what fused_steady_bus_kernel does with the bench's voices: the fraction of (voice slot, row) pairs
    whose homogeneous part is still carried.  Mirrors steady_prep_kernel's decay bound (rows from the cold start until
    the homogeneous part is below `tol` of the voice's full scale), the engine's voice order (engine.py:
    ordered_by_cutoff: groups of 64 neighbours in cutoff dealt round robin over the voice tiles, slot-major), the
    wave-maximum per slot and the kernel's row-group variants."""
    hz, cut = p['hertz'][0, lo:hi], p['cutoff'][0, lo:hi]
    k = np.tan(np.pi * (cut / (RATE / 2)) / 2); k2 = k * k; nrm = 1 / (1 + np.sqrt(2) * k + k2)
    b0, a1, a2 = k2 * nrm, 2 * (k2 - 1) * nrm, (1 - np.sqrt(2) * k + k2) * nrm
    d = hz / RATE; z = np.exp(-2j * np.pi * (d - np.rint(d)))
    H = (1 + 2 * z + z * z) / (1 + a1 * z + a2 * z * z)
    P = H - 1; Q = P / z - 2 + a1 * H
    d2 = a2 - 0.25 * a1 * a1; f2 = 1 + 0.25 * a1 * a1 + d2
    kappa = (f2 + np.sqrt(np.maximum(f2 * f2 - 4 * d2, 0))) / (2 * np.sqrt(d2))
    amp = b0 * kappa * np.sqrt(np.abs(P) ** 2 + np.abs(Q) ** 2)
    nd = np.where(amp > tol, np.ceil(np.log(tol / amp) / (0.5 * np.log(a2))) + 1, 0.0)
    v, tile = hi - lo, 64 * vpt
    if v % tile:
        return 1.0
    order = np.argsort(cut, kind='stable'); tiles = v // tile
    q = np.arange(v); group, lane = q // 64, q % 64
    perm = np.empty(v, dtype=np.int64)
    perm[((group % tiles) * 64 + lane) * vpt + group // tiles] = order[q]
    drop = np.maximum(nd[perm].reshape(tiles, 64, vpt).max(axis=1) - ctx, 0)          # (tile, slot): first row without it
    variants = {16: (16, 12, 8, 6, 4, 3, 2, 1, 0), 8: (8, 6, 4, 3, 2, 1, 0), 4: (4, 2, 1, 0), 2: (2, 1, 0), 1: (1, 0)}[vpt]
    R = 16 // bus_channels
    live = 0
    for t in range(tiles):
        for r0 in range(0, N - N % R, R):
            m = max([i + 1 for i in range(vpt) if drop[t, i] > r0], default=0)
            live += R * min(x for x in variants if x >= m)
        live += (N % R) * vpt
    return live / (tiles * N * vpt)


def fused_f64_ops_per_voice_sample(name, voices, N, K, ctx=100, bus_channels=2, steady=False, live_fraction=1.0):
    """f64-rate VALU instructions per stored voice-sample of the fused Sine kernels, counted in the ISA.
    Walker: 2 for the oscillator recurrence on every row a lane walks (span*N + c rows per span*N stored), 4 for
    the b0-normalised DF2T on (N + c)/N rows (every block is warmed up c rows), then per stored row either C bus FMAs +
    C/vpt adds of the cross-lane flush (16 adds per lane per 16/C rows), or 1 multiply + 1 conversion for the f32 store.
    Closed form (`steady`): 1 for the steady-state two-term recurrence, C bus FMAs, 17 adds per lane per 16/C rows of the
    folded cross-lane sum, and 3 (homogeneous recurrence + sum) on the `live_fraction` of (voice slot, row) pairs whose
    homogeneous part has not decayed yet; no warm-up rows."""
    from signals_amd import _native
    vpt, span = _native.fused_geometry(voices, N, K, ctx)
    sink = bus_channels + bus_channels / vpt if name == 'fused_voice_bus' else 2.0
    if steady and name == 'fused_voice_bus':
        plan = _native.fused_voice_bus_plan('Sine', ctx, voices, N, K, ctx)
        vpt, span = plan['voices_per_lane'], plan['blocks_per_lane']
        return 1.0 + bus_channels + 3.0 * live_fraction + 17.0 * bus_channels / (16 * vpt), vpt, span
    return 2.0 * (span * N + ctx) / (span * N) + 4.0 * (N + ctx) / N + sink, vpt, span


def kernel_table(summ: dict, bytes_per_unit) -> dict:
    """per engine launch name: calls, average HIP-event time, share of the timed GPU time, algorithmic GB/s"""
    total_ms = sum(e['ms'] for e in summ.values()) or 1.0
    table = {}
    for name, e in summ.items():
        bpu = bytes_per_unit(name)
        avg_ms = e['ms'] / e['calls']
        table[name] = {'calls': e['calls'], 'avg_ms': avg_ms, 'share': e['ms'] / total_ms,
                       'algo_bytes_per_voice_sample': bpu,
                       'algo_GBs': (bpu or 0) * (e['units'] / e['calls']) / (avg_ms * 1e-3) / 1e9}
    return table


def hbm_roofline(kernels: dict, dom: str, traffic) -> dict:
    k = kernels[dom]
    return {'bound': 'hbm', 'kernel': dom, 'achieved': k['algo_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': k['algo_GBs'] / HBM_PEAK_GBS, 'traffic': traffic,
            'algo_bytes_per_voice_sample': k['algo_bytes_per_voice_sample'], 'avg_launch_ms': k['avg_ms']}


def _with_practical(detail: dict, achieved: float) -> dict:
    if isinstance(detail.get('practical_peak'), dict):
        detail = dict(detail, practical_peak=dict(detail['practical_peak'], frac=achieved / detail['practical_peak']['value']))
    return detail


def valu_roofline(kernels: dict, dom: str, units_per_call: float, ops: float, traffic, detail: dict) -> dict:
    """a fused voice kernel: f64 VALU instruction-lanes per second against the chip's f64 vector issue peak; the HBM
    view of the same launch (it moves almost nothing) is nested under `hbm`"""
    k = kernels[dom]
    ach = ops * units_per_call / (k['avg_ms'] * 1e-3) / 1e12
    return {'bound': 'valu_f64', 'kernel': dom, 'achieved': ach, 'peak': F64_VALU_PEAK, 'unit': 'T f64-instr-lanes/s',
            'frac': ach / F64_VALU_PEAK, 'traffic': traffic, 'f64_ops_per_voice_sample': ops,
            'avg_launch_ms': k['avg_ms'], **_with_practical(detail, ach),
            'hbm': {'achieved': k['algo_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': k['algo_GBs'] / HBM_PEAK_GBS,
                    'algo_bytes_per_voice_sample': k['algo_bytes_per_voice_sample'],
                    'note': 'SURVEY.md 8d: the fused lower bound is ~16/V B per voice-sample, so this launch is f64-VALU-bound '
                            'by construction; the HBM-bound node-materialised schedule is under alt_schedule'}}


def pmc_traffic(family: str):
    tfile = ROOT / 'profiles' / 'traffic.json'
    if tfile.exists():
        return json.loads(tfile.read_text()).get(family)
    return None


def cpu_baseline(p, voices, frames, budget_s=12.0):
    """The reference's CPU path as restated by the oracle (pull protocol, per-channel butter+sosfilt per
    block, block caches, after-windows), single thread, on as many consecutive blocks as fit the budget."""
    from oracle import chain_ref as R
    import warnings
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    node, pan = cfg.c2_oracle(p, 0, voices)
    t0 = time.perf_counter()
    blocks = 0
    while True:
        R.sum_bus(R.render(node, blocks * frames, frames, voices, RATE), pan)
        blocks += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or blocks >= 64:
            break
    return dict(value=voices * frames * blocks / dt / 1e6, unit='Msamples/s', cores=1, kind='port',
                sample=f'{blocks} consecutive {frames}-frame blocks of the {voices}-voice C2 graph from position 0, '
                       f'{dt:.1f} s, oracle/chain_ref.py (numpy {np.__version__}, scipy butter+sosfilt per channel per block), '
                       f'host has {os.cpu_count()} logical cores')


def oracle_bus_blocks(p, voices, frames, positions):
    """float32(oracle float64) stereo bus of the C2 graph for the blocks starting at `positions` (the single-filter graph
    is position-pure: any block can be rendered on its own, fx.py:85-106)"""
    from oracle import chain_ref as R
    import warnings
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    node, pan = cfg.c2_oracle(p, 0, voices)
    return {pos: R.sum_bus(R.render(node, pos, frames, voices, RATE), pan).astype(np.float32).astype(np.float64)
            for pos in positions}


def check_batch_against_oracle(p, voices, frames, batch: torch.Tensor, batch_position: int, nblocks: int, indices):
    """max |GPU - float32(oracle)| per checked block of ONE rendered batch (rows = nblocks*frames from batch_position)"""
    indices = sorted({b for b in indices if 0 <= b < nblocks})
    ref = oracle_bus_blocks(p, voices, frames, [batch_position + b * frames for b in indices])
    got = batch.double().cpu().numpy()
    per_block = {}
    for b in indices:
        per_block[str(b)] = float(np.max(np.abs(got[b * frames:(b + 1) * frames] - ref[batch_position + b * frames])))
    scale = float(max(np.max(np.abs(r)) for r in ref.values()))
    return per_block, scale


def _cpu_worker(args):
    p, lo, hi, frames, blocks = args
    from oracle import chain_ref as R
    import warnings
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    node, pan = cfg.c2_oracle(p, lo, hi)
    acc = 0.0
    for b in range(blocks):
        acc += float(R.sum_bus(R.render(node, b * frames, frames, hi - lo, RATE), pan).sum())
    return acc


def cpu_baseline_sharded(p, voices, frames, workers, blocks=128):
    """the same oracle with the voices sharded over host cores (one process per shard, like
    the GPU path shards voices over GPUs); the per-shard stereo buses would be summed -- here only timed."""
    import multiprocessing as mp
    per = voices // workers
    jobs = [(p, w * per, (w + 1) * per, frames, blocks) for w in range(workers)]
    ctx = mp.get_context('fork')
    with ctx.Pool(workers) as pool:
        pool.map(_cpu_worker, [(p, 0, 4, frames, 1)] * workers)          # warm the workers (imports)
        t0 = time.perf_counter()
        pool.map(_cpu_worker, jobs)
        dt = time.perf_counter() - t0
    return dict(value=voices * frames * blocks / dt / 1e6, unit='Msamples/s', cores=workers, kind='port',
                sample=f'{blocks} consecutive {frames}-frame blocks, {voices} voices sharded {per} per process over '
                       f'{workers} processes, {dt:.1f} s')


# --------------------------------------------------------------------------------------------------- other configs
def run_config(name: str, steps: int = 0, prewarm_s: float = 1.0) -> dict:
    """BASELINE.json's C3 / C5 on this GPU through the engine's default schedule: throughput, per-kernel table, the
    dominant kernel's roofline, and max-abs error of the first rendered batch (the timed launch geometry) against the
    CPU oracle on whole-width blocks."""
    from oracle import chain_ref as R
    from signals_amd.engine import BatchRenderer, KernelTimer
    import warnings
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    if name == 'C3':
        V, channels, N, K = 1024, 1, 1024, 1024              # 1.07 G voice-samples per batch, like the headline's
        p = cfg.c3_params(V)
        steps = steps or 30
        node, workload = cfg.c3_graph(p), f'C3: {V}-voice Sawtooth->LowPass->LowPass->(x ADSR)->SumBus(mono), 48 kHz, {N}-frame blocks, {K} blocks per batch'
    else:
        V, channels, N, K = 4096, 4096, 256, 256             # 268 M voice-samples (1 GiB of mixed rows) per batch
        p = cfg.c5_params(V)
        steps = steps or 100
        node, workload = cfg.c5_graph(p), f'C5: {V}-voice Sine->LowPass->MixMatrix(64x64), 48 kHz, {N}-frame blocks, {K} blocks per batch'
    timer = KernelTimer(sample_every=8)                  # (an event pair around EVERY launch costs the stream ~10 % at these launch lengths)
    r = BatchRenderer(node, channels, RATE, timer=timer)
    first = r.render(0, N, K)
    if name == 'C3':
        # cascaded filters depend on the render history: the oracle renders sequentially from 0
        ref = R.sum_bus(R.render_stream(cfg.c3_oracle(p), 0, N, 2, V)).astype(np.float32).astype(np.float64)
        got = first[:2 * N].double().cpu().numpy()
        errs = {str(b): float(np.max(np.abs(got[b * N:(b + 1) * N] - ref[b * N:(b + 1) * N]))) for b in (0, 1)}
        sample = (f'mono bus, blocks 0 and 1 of the first {K}-block batch (all {V} voices) vs the CPU oracle rendered sequentially '
                  f'from 0 (cascades depend on the render history, so later blocks would cost the oracle the whole stream)')
    else:
        oracle = cfg.c5_oracle(p)
        errs = {}
        for b in (0, K - 1):
            ref = R.render(oracle, b * N, N, V, RATE).astype(np.float32).astype(np.float64)
            errs[str(b)] = float(np.max(np.abs(first[b * N:(b + 1) * N].double().cpu().numpy() - ref)))
        sample = f'all {V} mixed voices, blocks 0 and {K - 1} of the first {K}-block batch vs the CPU oracle'
    del first
    def timed(K, steps, pos):
        t_end = time.perf_counter() + prewarm_s                                # the oracle above left the GPU idle: clocks back up first
        while time.perf_counter() < t_end:
            r.render(pos, N, K); pos += N * K
            torch.cuda.synchronize()
        timer.reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render(pos, N, K); pos += N * K
        torch.cuda.synchronize()
        return time.perf_counter() - t0, pos
    dt, pos = timed(K, steps, N * K)
    summ = timer.summary()
    from signals_amd import _native

    def bytes_per_unit(kname):
        if kname.split('[')[0] == 'fused_cascade_bus':                 # one f64 partial per (voice tile, frame, channel), written and re-read, + the bus
            tiles = -(-V // (64 * _native.fused_cascade_geometry(V, K)[0]))
            return (tiles * channels * 8 * 2 + channels * 4) / V
        return ALGO_BYTES.get(kname.split('[')[0])
    kernels = kernel_table(summ, bytes_per_unit)
    dom = max(summ, key=lambda k: summ[k]['ms'])
    fam = dom.split('[')[0]
    model = _native.fused_cascade_model(V, N, K, bus_channels=channels) if fam == 'fused_cascade_bus' else None
    if model is not None:
        roof = valu_roofline(kernels, dom, summ[dom]['units'] / summ[dom]['calls'], model['f64_ops_per_voice_sample'],
                             pmc_traffic(f'{name}/{fam}'), {k: v for k, v in model.items() if k != 'f64_ops_per_voice_sample'})
    else:
        roof = hbm_roofline(kernels, dom, pmc_traffic(f'{name}/{fam}'))
        if fam == 'fused_osc_biquad_mix':
            # chain + matrix in one launch.  The only HBM traffic is the mixed float32 rows (4 B per voice-sample), and that
            # is the roof the launch is closest to since the contraction moved to bf16 MFMAs (sig_mix_tile.h: float32 =
            # three bfloat16, six exact products per k-block -> 768 executed flop per voice-sample at the 2.5 PFLOP/s bf16
            # rate instead of 128 at the 157 TFLOP/s of v_mfma_f32_32x32x2_f32).  Measured by leaving parts out
            # (DESIGN.md 7): stores ~22 us, MFMAs ~28 us, vector work (rows, float32 -> 3 x bf16) ~21 us, per-wave set-up
            # ~6 us of the launch, adding up rather than overlapping at two waves per SIMD.
            secs = kernels[dom]['avg_ms'] * 1e-3
            roof['mfma'] = {'executed_flop_per_voice_sample': 768, 'algorithmic_flop_per_voice_sample': 128,
                            'achieved': 768 * V * N * K / secs / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s (bf16, dense)',
                            'frac': 768 * V * N * K / secs / 1e12 / 2500.0,
                            'instruction': 'v_mfma_f32_32x32x16_bf16, 48 per 32-row x 64-voice tile'}
            extra_c5 = {'contraction': 'float32 rows x float32 matrix, float32 accumulators; products exact: each float32 operand is the sum of '
                                       'three bfloat16 (six v_mfma_f32_32x32x16_bf16 terms per k-block, the three below 2^-26 dropped); measured '
                                       'against the f64 product of the same rows: 2.3-4.8 ulp of the rows\' scale, v_mfma_f32_32x32x2_f32 2.8-5.3 '
                                       '(tests/test_gpu_fused_walker.py); f32_mfma_sink times the same launch on that instruction'}
            roof['store_rate_of_a_plain_fill_GBs'] = 6900.0     # torch fill_ of 256 MiB on this GPU (tools/ubench/torch_bandwidth.py): the practical write roof
    full_scale = float(np.max(np.abs(ref)))
    extra = {}
    if name == 'C3':
        # the same stream in batches as long as the headline's: one history block per 16 blocks instead of per 4
        K4 = 4096
        dt4, pos = timed(K4, 10, pos)
        extra['long_batches'] = {'blocks_per_batch': K4, 'value': V * N * K4 * 10 / dt4 / 1e6, 'unit': 'Msamples/s',
                                 'ms_per_step': dt4 / 10 * 1e3, 'steps': 10,
                                 'launch_geometry': dict(zip(('voices_per_lane', 'blocks_per_lane'), _native.fused_cascade_geometry(V, K4)))}
    if name == 'C5':
        # BASELINE names this configuration "MFMA fp32": the same launch with the sink on v_mfma_f32_32x32x2_f32 (the default
        # contracts each float32 as three bfloat16 -- exact products, float32 accumulators, sig_mix_tile.h)
        _native.set_fused_tuning(0, 0, 3, -1)
        try:
            r32 = BatchRenderer(node, channels, RATE)
            got32 = r32.render(0, N, K)[:N].double().cpu().numpy()
            ref0 = R.render(cfg.c5_oracle(p), 0, N, V, RATE).astype(np.float32).astype(np.float64)
            r, keep = r32, r
            dt32, _ = timed(K, max(steps // 2, 10), N * K)
            r = keep
            extra['f32_mfma_sink'] = {'value': V * N * K * max(steps // 2, 10) / dt32 / 1e6, 'unit': 'Msamples/s',
                                      'ms_per_step': dt32 / max(steps // 2, 10) * 1e3,
                                      'max_abs_error_block_0': float(np.max(np.abs(got32 - ref0))),
                                      'instruction': 'v_mfma_f32_32x32x2_f32, 64 per 32-row x 64-voice tile'}
        finally:
            _native.set_fused_tuning()
    if name == 'C5':
        extra.update(extra_c5)
    return {**extra, 'workload': workload, 'value': V * N * K * steps / dt / 1e6, 'unit': 'Msamples/s', 'ms_per_step': dt / steps * 1e3,
            'steps': steps, 'roofline': roof, 'kernels': kernels,
            'max_abs_error': {'per_block': errs, 'max': max(errs.values()), 'full_scale': full_scale,
                              'max_relative_to_full_scale': max(errs.values()) / max(1.0, full_scale),
                              'bar': '1e-6 of full scale: the outputs are sums over voices (full scale > 1), stored as float32 '
                                     '(one ulp at full scale = %.1e); C5 contracts exact products of the float32 rows and matrix in float32 accumulators. '
                                     'Both nodes are build-defined (SURVEY.md 8a A11): parity is against the oracle definition'
                                     % float(np.spacing(np.float32(full_scale))),
                              'sample': sample}}


def run_modulated(K: int = 1024, steps: int = 30, prewarm_s: float = 0.5) -> dict:
    """C2's 1024-voice graph with its control ports driven at block rate (vibrato, LFO-swept cutoff, tremolo: what the
    reference's forward_at_block_rate allows on top of the headline graph; not a BASELINE configuration).  The voice stays one
    launch -- the row walker with per-block hertz / cutoff / gain rows -- behind a few block-rate control launches."""
    from oracle import chain_ref as R
    from signals_amd.engine import BatchRenderer, KernelTimer
    import warnings
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    V, N = 1024, 256
    p = cfg.c2_params(V)
    out = {'workload': f'C2 voices ({V}) with block-rate modulation (vibrato / cutoff sweep / tremolo), {N}-frame blocks, {K} blocks per batch',
           'unit': 'Msamples/s', 'voices': {}}
    legs = {'Sawtooth': ('Sawtooth', True, True, True), 'Sine': ('Sine', True, True, True),
            'Sine_sweep_tremolo': ('Sine', False, True, True), 'Sine_sweep': ('Sine', False, True, False)}
    for label, (kind, vibrato, sweep, tremolo) in legs.items():
        timer = KernelTimer(sample_every=4)
        r = BatchRenderer(cfg.c2_modulated_graph(p, kind, vibrato, sweep, tremolo), 2, RATE, timer=timer)
        first = r.render(0, N, K)
        node, pan = cfg.c2_modulated_oracle(p, kind, vibrato, sweep, tremolo)
        ref = R.sum_bus(R.render_stream(node, 0, N, 2, V), pan).astype(np.float32).astype(np.float64)
        err = float(np.max(np.abs(first[:2 * N].double().cpu().numpy() - ref)))
        pos = N * K
        t_end = time.perf_counter() + prewarm_s
        while time.perf_counter() < t_end:
            r.render(pos, N, K); pos += N * K
            torch.cuda.synchronize()
        timer.reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render(pos, N, K); pos += N * K
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        summ = timer.summary()
        out['voices'][label] = {'modulation': '+'.join(n for n, on in (('vibrato', vibrato), ('cutoff sweep', sweep), ('tremolo', tremolo)) if on),
                                'value': V * N * K * steps / dt / 1e6, 'ms_per_step': dt / steps * 1e3,
                                'max_abs_error_blocks_0_1': err, 'full_scale': float(np.max(np.abs(ref))),
                                'launches_per_step': {k: round(v['ms'] / v['calls'], 4) for k, v in summ.items()}}
        # the same leg with the control program's kernel built for this graph's program (BatchRenderer(specialise=True): hipcc at
        # the first render, then the disk cache): registers in VGPRs instead of an LDS file behind an interpretive loop
        timer = KernelTimer(sample_every=4)
        r = BatchRenderer(cfg.c2_modulated_graph(p, kind, vibrato, sweep, tremolo), 2, RATE, timer=timer, specialise=True)
        first = r.render(0, N, K)
        err = float(np.max(np.abs(first[:2 * N].double().cpu().numpy() - ref)))
        pos = N * K
        t_end = time.perf_counter() + prewarm_s / 2
        while time.perf_counter() < t_end:
            r.render(pos, N, K); pos += N * K
            torch.cuda.synchronize()
        timer.reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render(pos, N, K); pos += N * K
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        summ = timer.summary()
        if any(k.endswith('*specialised') for k in summ):
            out['voices'][label]['specialised'] = {'value': V * N * K * steps / dt / 1e6, 'ms_per_step': dt / steps * 1e3,
                                                   'max_abs_error_blocks_0_1': err,
                                                   'launches_per_step': {k: round(v['ms'] / v['calls'], 4) for k, v in summ.items()}}
    return out


def run_programs(K: int = 1024, steps: int = 10) -> dict:
    """Graph shapes none of the fused kernels covers (a Mix or a RingMod behind filters, a modulated cascade ...): ONE interpreted
    launch per batch (sig_voice_program) against the same graph one kernel per node -- tools/time_voice_program.py; 1024 voices under a
    stereo bus, 256-frame blocks.  Not a BASELINE configuration: what the reference's node API allows beyond them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('time_voice_program', ROOT / 'tools' / 'time_voice_program.py')
    tvp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tvp)
    V, N = 1024, 256
    out = {'workload': f'{V} voices under a stereo bus, {N}-frame blocks, {K} blocks per batch', 'unit': 'Msamples/s', 'shapes': {}}
    for name, build in tvp.shapes(V).items():
        if name in ('amp_after_filter', 'three_filters', 'osc_gain_only'):
            continue                                     # (the interpreter's full register file loses to one kernel per node: the engine does not pick it)
        fast, launches = tvp.rate(build, V, N, K, True, steps=steps)
        slow, _ = tvp.rate(build, V, N, K, False, steps=4)
        out['shapes'][name] = {'value': fast * 1e6, 'one_kernel_per_node': slow * 1e6, 'launches_us': launches}
        # the same program as a kernel built for it (BatchRenderer(specialise=True): hipcc at the first render, then the disk cache)
        spec, spec_launches = tvp.rate(build, V, N, K, 'always', steps=steps, specialise=True)
        if any(k.endswith('*specialised') for k in spec_launches):
            out['shapes'][name]['specialised'] = {'value': spec * 1e6, 'launches_us': spec_launches}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--prewarm-ms', type=float, default=1500.0,
                    help='untimed: keep rendering batches for this long before the W warm-up steps, so that the GPU clocks '
                         'have settled under load (the first ~50 ms after idle run 15-30 % slower, and the rate keeps creeping up for ~2 s: '
                         '4.85 / 5.09 / 5.24 T voice-samples/s after 250 / 1000 / 2000 ms); 0 disables')
    ap.add_argument('--blocks', type=int, default=4096, help='256-frame blocks per batch (one step)')
    ap.add_argument('--voices', type=int, default=1024, help='voices per GPU')
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--position', type=int, default=0)
    ap.add_argument('--sustained-s', type=float, default=2.0, help='length of the sustained leg (0 disables)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the C3 / C5 legs')
    ap.add_argument('--materialised', action='store_true', help='headline = one kernel per node (no fusion)')
    ap.add_argument('--single-mode', action='store_true', help='only the headline measurement (profiling runs)')
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  RCCL prints its version banner to stdout (fd 1) when the first
    # communicator is created, so everything but the final line is sent to stderr at the fd level.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert torch.cuda.is_available(), 'bench.py needs an MI355X'

    from signals_amd import _native, runtime
    runtime.set_device(f'cuda:{local_rank % torch.cuda.device_count()}')
    _native.lib()
    from signals_amd import parallel
    from signals_amd.engine import KernelTimer
    parallel.init_process_group()                        # RCCL (backend "nccl") when WORLD_SIZE > 1
    backend = dist.get_backend() if dist.is_initialized() else None
    V, N, K = args.voices, args.frames, args.blocks
    params = synth_params(V * world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(fuse: bool, steps: int, warmup: int, sustained_s: float = 0.0) -> dict:
        """W untimed + exactly `steps` timed batches of this rank's 1024-voice graph (+ bus reduce); then, optionally,
        the same steps for `sustained_s` more seconds"""
        # fused schedule: one ~200-us launch per step -- ONE HIP event in front of the first timed launch and one behind the last
        # (nothing between the launches: the stream runs as it does untimed, so avg_launch_ms is the per-step GPU time, gaps
        # included, and cannot exceed ms_per_step); node-materialised: every launch bracketed
        timer = None if args.no_kernel_timing else (KernelTimer(region=True) if fuse else KernelTimer(sample_every=1))
        renderer = parallel.ShardedRenderer(lambda lo, hi: build_graph(params, lo, hi), V * world, bus_channels=2,
                                            rate=RATE, timer=timer, fuse=fuse)
        assert (renderer.lo, renderer.hi) == (rank * V, (rank + 1) * V)
        pos = args.position
        pending = None

        def step():
            # (K*N, 2) f32 bus: local render, then the RCCL reduce of THIS batch is left in flight on
            # RCCL's stream while the next batch renders; it is waited for one step later (and at the fence)
            nonlocal pos, pending
            bus, work = renderer.render_async(pos, N, K, dst=0)        # the sink lives on rank 0: reduce, not all-reduce
            pos += N * K
            if pending is not None:
                pending.wait()
            pending = work
            return bus

        if args.prewarm_ms > 0:                              # untimed, same work: clocks ramp up under load
            fence()
            t_pre = time.perf_counter()
            for _ in range(4):
                step()
            fence()
            t4 = torch.tensor([time.perf_counter() - t_pre], dtype=torch.float64, device='cuda')
            if world > 1:
                dist.all_reduce(t4, op=dist.ReduceOp.MAX)   # the same step count on every rank (one reduce per step)
            for _ in range(int(min(100000, args.prewarm_ms * 1e-3 / max(float(t4.item()) / 4, 1e-6)))):
                step()
            pos = args.position                             # the timed stream starts where it says
        for _ in range(warmup):
            step()
        if pending is not None:
            pending.wait()
            pending = None
        fence()
        if timer:
            timer.reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            bus = step()
        if timer:
            timer.close()
        if pending is not None:
            pending.wait()
            pending = None
        fence()
        dt = time.perf_counter() - t0
        last_position = pos - N * K                          # where the batch in `bus` starts
        runtime.check_status()
        assert torch.isfinite(bus).all()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        samples = V * world * N * K * steps
        res = {'value': samples / dt / 1e6, 'ms_per_step': dt / steps * 1e3, 'last_bus': bus, 'last_position': last_position}
        if timer:
            summ = timer.summary()
            closed = fuse and steady_applies(params, rank * V, (rank + 1) * V, args.position,
                                             args.position + (warmup + steps) * N * K - 1, N)

            def bytes_per_unit(kname):
                fam = kname.split('[')[0]
                if fam == 'fused_voice_bus':
                    vpl = fused_f64_ops_per_voice_sample('fused_voice_bus', V, N, K, steady=closed)[1]
                    tiles = -(-V // (64 * vpl))                     # one f64 partial per (voice tile, frame, channel), written and re-read
                    return (tiles * 2 * 8 * 2 + 2 * 4) / V
                return ALGO_BYTES.get(fam, 0)
            kernels = kernel_table(summ, bytes_per_unit)
            dom = max(summ, key=lambda k: summ[k]['ms'])
            fam = dom.split('[')[0]
            if fam in VALU_BOUND:
                plan = _native.fused_voice_bus_plan('Sine', args.position + N * K, V, N, K, 100)
                live = closed_form_live_fraction(params, rank * V, (rank + 1) * V, N, plan['voices_per_lane']) if closed else 1.0
                ops, vpt, span = fused_f64_ops_per_voice_sample(fam, V, N, K, steady=closed, live_fraction=live)
                res['roofline'] = valu_roofline(kernels, dom, summ[dom]['units'] / summ[dom]['calls'], ops, pmc_traffic(fam), {
                    'voices_per_lane': vpt, 'blocks_per_lane': span, 'device_kernel': plan['kernel'].replace('C>', '2>'),
                    'homogeneous_live_fraction': live,
                    'practical_peak': {'value': 30.8, 'frac': None,
                                       'note': 'tools/ubench/f64_rates.hip: back-to-back independent v_fma_f64 reach 5.1 cycles per '
                                               'wave-instruction counted at 2.4 GHz (the chip holds ~1.9 GHz under f64 load), i.e. '
                                               '30.8 T instr-lanes/s is what this instruction mix can reach at all'},
                    'path': 'closed form: steady-state sinusoid + homogeneous transient per block, no warm-up rows '
                            '(fused_steady_bus_kernel); constant-parameter Sine->LowPass|HighPass->[Gain]->SumBus takes it; other '
                            'oscillators, per-block (LFO) cutoff / gain rows and two-oscillator voices run the span walker in one '
                            'launch too (1.2-2.5 T voice-samples/s), anything else per-node kernels' if closed else 'span walker (fused_walk_kernel)',
                    'launches': 'avg_launch_ms = (one HIP event behind the last timed launch - one in front of the first) / launches: the '
                                'per-step time on the GPU timeline, launch gaps included, nothing recorded between the launches; one kernel per batch -- it adds its two voice tiles itself '
                                '(sig_bus::sum_tiles_in_workgroup) -- plus steady_prep_kernel on the calls where the per-voice '
                                "constants change; the kernel's duration under the profiler is in profiles/*_kernel_stats.csv"})
            else:
                res['roofline'] = hbm_roofline(kernels, dom, pmc_traffic(fam))
            res['kernels'] = kernels
        if sustained_s > 0 and world == 1:
            # at 1.07 G voice-samples per 0.33 ms this leg covers DAYS of audio; a real stream is nowhere near the
            # 2^26-cycle hand-over of the closed form (10.6 h at 1760 Hz), so the stream position wraps every `wrap` batches
            n = max(steps, int(sustained_s / (dt / steps)) + 1)
            wrap = max(1, min(1024, int(2.0 ** 26 / 1760.0 * RATE) // (N * K) - 1))
            renderer.renderer.timer = None                    # no HIP events: thousands of un-synchronised steps
            fence()
            t0 = time.perf_counter()
            for i in range(n):
                if i % wrap == 0:
                    pos = args.position
                step()
            fence()
            ds = time.perf_counter() - t0
            res['sustained'] = {'seconds': ds, 'steps': n, 'Msamples_per_s': V * N * K * n / ds / 1e6, 'ms_per_step': ds / n * 1e3,
                                'note': f'same steps back to back; the stream position restarts every {wrap} batches '
                                        f'({wrap * N * K / RATE / 3600:.1f} h of audio)'}
        return res

    main_mode = measure(fuse=not args.materialised, steps=args.steps, warmup=args.warmup,
                        sustained_s=0.0 if args.single_mode else args.sustained_s)
    other = None
    if not args.single_mode:
        other = measure(fuse=args.materialised, steps=max(3, args.steps // 2), warmup=min(2, args.warmup))

    by_batch = None
    if world == 1 and not args.single_mode and not args.materialised:
        # the same schedule at other batch lengths, clocks already settled.  K = 256 is SURVEY.md 8d's "throughput mode": a leg of its
        # own with a roofline entry (region-timed like the headline)
        from signals_amd.engine import BatchRenderer
        by_batch = {}
        for k in (256, 1024, 8192, 16384):                  # (the longer ones: two and four waves per SIMD instead of one)
            if k == K:
                continue
            timer = KernelTimer(region=True)
            r = BatchRenderer(build_graph(params, 0, V), 2, RATE, timer=timer)
            pos, reps = 0, max(20, 200 * 1024 // max(k, 1024))
            wrap = max(1, int(2.0 ** 26 / 1760.0 * RATE) // (N * k) - 1)         # stay inside the closed form's phase range
            for i in range(10):
                r.render(pos, N, k); pos += N * k
            torch.cuda.synchronize()
            timer.reset()
            t0 = time.perf_counter()
            for i in range(reps):
                if i % wrap == 0:
                    pos = 0
                r.render(pos, N, k); pos += N * k
            timer.close()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            entry = {'Msamples_per_s': V * N * k / dt / 1e6, 'ms_per_step': dt * 1e3}
            summ = timer.summary()
            if k == 256 and summ:
                plan = _native.fused_voice_bus_plan('Sine', N * k, V, N, k, 100)
                live = closed_form_live_fraction(params, 0, V, N, plan['voices_per_lane'])
                ops = 1.0 + 2 + 3.0 * live + 17.0 * 2 / (16 * plan['voices_per_lane'])
                dom = max(summ, key=lambda n_: summ[n_]['ms'])
                avg_ms = summ[dom]['ms'] / summ[dom]['calls']
                ach = ops * V * N * k / (avg_ms * 1e-3) / 1e12
                entry['roofline'] = {'bound': 'valu_f64', 'kernel': dom, 'achieved': ach, 'peak': F64_VALU_PEAK, 'unit': 'T f64-instr-lanes/s',
                                     'frac': ach / F64_VALU_PEAK, 'f64_ops_per_voice_sample': ops, 'avg_launch_ms': avg_ms,
                                     'voices_per_lane': plan['voices_per_lane'], 'blocks_per_lane': plan['blocks_per_lane'],
                                     'homogeneous_live_fraction': live,
                                     'note': 'SURVEY.md 8d throughput mode (K = 256): ONE round of 1024 waves of one block each, one per SIMD, '
                                             'plus a launch per 67 M voice-samples; giving a block to 2 / 4 waves changes nothing (24.0 / 31.9 us '
                                             'against 24.4), so the round is issue-bound already: ~13 us of arithmetic (12.5 us inside a long '
                                             'batch) + launch and drain (tools/time_fused_geom.py, tools/time_k256.py, DESIGN.md 7)'}
            if k in (256, 1024):
                # the same batches over two alternating HIP streams (BatchRenderer(pipeline=2): each stream its own workspace, the
                # caller's stream waits for every batch; the same bits): the tail of one launch under the head of the next
                rp = BatchRenderer(build_graph(params, 0, V), 2, RATE, pipeline=2)
                pos = N * k
                for i in range(20):
                    rp.render(pos, N, k); pos += N * k
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(reps):
                    if i % wrap == 0:
                        pos = N * k
                    rp.render(pos, N, k); pos += N * k
                torch.cuda.synchronize()
                dtp = (time.perf_counter() - t0) / reps
                entry['two_streams'] = {'Msamples_per_s': V * N * k / dtp / 1e6, 'ms_per_step': dtp * 1e3}
            by_batch[str(k)] = entry

    latency = None
    if world == 1 and not args.single_mode:
        # latency mode: ONE 256-frame block per request, as a real-time sink would pull it
        from signals_amd.chain import BlockLoc, Shape
        from signals_amd.chain.driver import BlockDriver
        from signals_amd.engine import BatchRenderer
        graph = build_graph(params, 0, V)
        eng = BatchRenderer(graph, 2, RATE)
        drv = BlockDriver(rate=RATE, blocksize=N)
        drv.get_state().channels = 2
        drv.input = build_graph(params, 0, V)
        latency = {}
        eng_graph = BatchRenderer(build_graph(params, 0, V), 2, RATE, graph_replay=True)
        for name, fn in (('engine_one_block_per_launch', lambda i: eng.render(i * N, N, 1)),
                         ('engine_hipgraph_replay', lambda i: eng_graph.render(i * N, N, 1)),
                         ('eager_pull_one_block', lambda i: drv.input.request(BlockLoc(
                             position=i * N, rate=RATE, shape=Shape(N, 2))))):
            for i in range(20):
                fn(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(20, 220):
                fn(i)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 200 * 1e6
            latency[name] = {'us_per_block': us, 'Msamples_per_s': V * N / us, 'x_realtime': (N / RATE * 1e6) / us}

    configs = None
    if world == 1 and not args.single_mode and not args.no_configs and not args.no_cpu_baseline:
        configs = {name: run_config(name) for name in ('C3', 'C5')}
        configs['C2_modulated'] = run_modulated()
        configs['voice_programs'] = run_programs()

    if rank == 0:
        def describe(fused):
            return ('fused voice chain + bus: sig_fused_voice_bus (Sine->LowPass->Gain->SumBus in ONE launch, its voice tiles '
                    'added in fixed order by the same kernel; no per-voice sample touches HBM)' if fused else
                    'node-materialised: one kernel per node (osc_bank, biquad_coldstart, elementwise[Gain], sum_bus), '
                    'every edge f32 in HBM (24 B/voice-sample, SURVEY.md 8d)')
        line = {
            'metric': 'rendered Msamples/s (48 kHz, 256-sample blocks)',
            'value': main_mode['value'],
            'unit': 'Msamples/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': main_mode['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'C2: {V}-voice Fixed->Sine->LowPass->Gain->SumBus(stereo) per GPU, 48 kHz, '
                                   f'{N}-frame blocks, {K} blocks per batch (f32 storage, f64 phase/recurrence); '
                                   f'engine schedule = {describe(not args.materialised)}',
                       'voices_per_gpu': V, 'block_frames': N, 'blocks_per_step': K, 'start_position': args.position,
                       'parallelism': f'voices sharded {V}/GPU x{world}; '
                                      + (f'torch.distributed backend {backend}{" (RCCL over xGMI)" if backend == "nccl" else ""}, '
                                         f'{world} ranks: one async reduce of the stereo bus to rank 0 per batch'
                                         if backend else 'single process, no collective'),
                       'rccl_ranks': world if backend == 'nccl' else 0, 'dist_backend': backend},
        }
        for k in ('roofline', 'kernels', 'sustained'):
            if k in main_mode:
                line[k] = main_mode[k]
        if other is not None:
            alt = {k: v for k, v in other.items() if k not in ('last_bus', 'last_position')}
            alt['schedule'] = describe(args.materialised)
            if not args.materialised:
                # SURVEY.md 8d's whole-graph view of the per-node schedule: 24 algorithmic bytes per voice-sample
                alt['hbm_frac_at_24_bytes_per_voice_sample'] = other['value'] * 1e6 * 24 / (HBM_PEAK_GBS * 1e9)
            line['alt_schedule'] = alt
        if by_batch:
            line['fused_schedule_at_other_batch_sizes'] = by_batch
        if latency is not None:
            line['latency_mode'] = latency
        if not args.no_cpu_baseline and world == 1:
            # BASELINE.json's second metric, on the output of the LAST TIMED batch of each schedule: blocks 0..7, the middle
            # one and the last one against float32(oracle float64); the batch starts (warmup + steps - 1) batches into the stream
            errors = {}
            for label, mode, picks in (('materialised' if args.materialised else 'fused', main_mode, list(range(8)) + [K // 2, K - 1]),
                                       ('fused' if args.materialised else 'materialised', other, [0, K // 2, K - 1])):
                if mode is None:
                    continue
                per_block, scale = check_batch_against_oracle(params, V, N, mode['last_bus'], mode['last_position'], K, picks)
                errors[label] = {'max': max(per_block.values()), 'per_block': per_block, 'batch_position': mode['last_position']}
                errors['full_scale'] = scale
            plan = _native.fused_voice_bus_plan('Sine', main_mode['last_position'], V, N, K, 100)
            closed = steady_applies(params, 0, V, main_mode['last_position'], main_mode['last_position'] + N * K - 1, N)
            line['max_abs_error'] = dict(
                errors, bar=1e-6,
                fused_kernel=(plan['kernel'].replace('C>', '2>') if closed or not plan['closed_form'] else 'fused_steady_bus_kernel: per-wave fallback'),
                fused_launch=f"fused_voice_bus[Sine,lp,gain]: {plan['voices_per_lane']} voices x {plan['blocks_per_lane']} blocks per lane",
                sample=f'stereo bus of the last timed {K}-block batch of the {V}-voice graph (its own output tensor), whole '
                       f'blocks against the CPU oracle rendering the same frame positions')
            line['cpu_baseline'] = cpu_baseline(params, V, N)
            workers = min(16, os.cpu_count() or 1)                      # the GPU box's CPU share for one GPU
            if workers > 1 and V % workers == 0:
                line['cpu_baseline_sharded'] = cpu_baseline_sharded(params, V, N, workers)
        if configs:
            line['configs'] = configs
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + '\n').encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
