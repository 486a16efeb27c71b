#!/usr/bin/env python3
"""Headline benchmark: rendered Msamples/s (voice-samples per wall second), 48 kHz, 256-frame blocks.

Workload (BASELINE.json configs[1]): per GPU a 1024-voice
`Fixed -> Sine -> LowPass -> Gain -> SumBus(stereo)` graph built through the node API and rendered by
the batched engine, one step = one batch of `--blocks` (default 1024) consecutive 256-frame blocks of synthetic
parameters (numpy default_rng(0): hertz U(55,1760), phase U(0,1), cutoff U(200,8000), gain U(0,1)/V,
pan theta U(0,pi/2)), already resident in HBM.  With N > 1 GPUs every rank renders its own 1024 voices
(weak scaling, no data-path traffic) and the stereo bus is summed onto rank 0 with one RCCL reduce per
batch (the path's only exchange step), overlapped with the next batch's kernels.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--blocks B] [--voices V] [--frames F]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (HIP events around every launch, on the launch stream) and `cpu_baseline` (the CPU oracle,
structured like the reference: per-channel butter + sosfilt per block, timed on a bounded sample).
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

RATE = 48000
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
# algorithmic HBM bytes per voice-sample, f32 storage, every node output written once and read once
# per consumer (SURVEY.md §8d, C2 = 24 B over the four kernels)
ALGO_BYTES = {'osc_bank': 4, 'biquad_coldstart': 8, 'elementwise': 8, 'sum_bus': 4,
              'fused_osc_biquad': 4,      # fused chain: only the f32 store reaches HBM
              'fused_voice_bus': None}    # f64 tile partials (written, re-read) + the f32 bus: depends on the voices per lane
FUSED_KERNELS = ('fused_osc_biquad', 'fused_voice_bus')


def steady_applies(p, lo, hi, first_frame, last_frame, N, ctx=100):
    """host mirror of fused_voice.hip:steady_voice_ok for every voice of the shard: does sig_fused_voice_bus run its
    closed-form Sine kernel (steady-state sinusoid + homogeneous transient) rather than the row-by-row walker?"""
    hz, ph = p['hertz'][0, lo:hi], p['phase'][0, lo:hi]
    d = hz / RATE
    dr = d - np.rint(d)
    t = np.abs(np.stack([first_frame / RATE * hz + ph, last_frame / RATE * hz + ph]))
    return bool((t < 2.0 ** 24).all() and (np.abs(dr) <= 0.25).all() and (np.abs(np.sin(2 * np.pi * dr)) >= 1e-3).all()
                and (N >= ctx or first_frame >= ctx))


def fused_f64_ops_per_voice_sample(name, voices, N, K, ctx=100, bus_channels=2, steady=False):
    """f64-rate VALU instructions per stored voice-sample of the fused Sine kernels, counted in the ISA.
    Walker: 2 for the oscillator recurrence on every row a lane walks (span*N + c rows per span*N stored), 4 for
    the b0-normalised DF2T on (N + c)/N rows (every block is warmed up c rows).  Closed form (`steady`): 2 for the
    steady-state recurrence, 2 for the homogeneous one, 1 to add them, no warm-up rows.  Then per stored row either
    C bus FMAs + C/vpt adds of the cross-lane flush (16 adds per lane per 16/C rows), or 1 multiply + 1 conversion
    for the f32 store."""
    from signals_amd import _native
    vpt, span = _native.fused_geometry(voices, N, K, ctx)
    sink = bus_channels + bus_channels / vpt if name == 'fused_voice_bus' else 2.0
    if steady and name == 'fused_voice_bus':
        # launch_voice_bus: the closed-form kernel takes 8 voices per lane when that still leaves a wave per SIMD
        if vpt == 4 and -(-voices // 512) * -(-K // span) >= 1024:
            vpt = 8
        return 5.0 + bus_channels + bus_channels / vpt, vpt, span
    return 2.0 * (span * N + ctx) / (span * N) + 4.0 * (N + ctx) / N + sink, vpt, span


def synth_params(total_voices: int):
    rng = np.random.default_rng(0)
    hertz = rng.uniform(55, 1760, size=(1, total_voices))
    phase = rng.uniform(0, 1, size=(1, total_voices))
    cutoff = rng.uniform(200, 8000, size=(1, total_voices))
    gain = rng.uniform(0, 1, size=(1, total_voices)) / total_voices
    theta = rng.uniform(0, np.pi / 2, size=total_voices)
    pan = np.stack([np.cos(theta), np.sin(theta)])
    return dict(hertz=hertz, phase=phase, cutoff=cutoff, gain=gain, pan=pan)


def build_graph(p, lo, hi):
    from signals_amd.chain.ext import SumBus
    from signals_amd.chain.fixed import Fixed
    from signals_amd.chain.fx import Gain, LowPass
    from signals_amd.chain.osc import Sine

    def fixed(v):
        f = Fixed()
        f.get_state().value = np.ascontiguousarray(v)
        return f

    osc = Sine()
    osc.hertz = fixed(p['hertz'][:, lo:hi])
    osc.phase = fixed(p['phase'][:, lo:hi])
    lp = LowPass()
    lp.input = osc
    lp.cutoff = fixed(p['cutoff'][:, lo:hi])
    g = Gain()
    g.left = lp
    g.right = fixed(p['gain'][:, lo:hi])
    bus = SumBus()
    bus.input = g
    bus.get_state().gains = np.ascontiguousarray(p['pan'][:, lo:hi])
    return bus


def cpu_baseline(p, voices, frames, budget_s=12.0):
    """The reference's CPU path as restated by the oracle (pull protocol, per-channel butter+sosfilt per
    block, block caches, after-windows), single thread, on as many consecutive blocks as fit the budget."""
    from oracle import chain_ref as R
    import warnings
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    sl = slice(0, voices)
    node = R.Binary('Gain', R.Filter('lp', R.Osc('Sine', R.Fixed(p['hertz'][:, sl]), R.Fixed(p['phase'][:, sl])),
                                     R.Fixed(p['cutoff'][:, sl])), R.Fixed(p['gain'][:, sl]))
    pan = p['pan'][:, sl]
    t0 = time.perf_counter()
    blocks = 0
    kept = []                                   # the first blocks' stereo bus, for the max-abs-error leg
    while True:
        bus = R.sum_bus(R.render(node, blocks * frames, frames, voices, RATE), pan)
        if blocks < 8:
            kept.append(bus)
        blocks += 1
        dt = time.perf_counter() - t0
        if dt > budget_s or blocks >= 64:
            break
    cpu_baseline.reference_bus = np.concatenate(kept)
    return dict(value=voices * frames * blocks / dt / 1e6, unit='Msamples/s', cores=1, kind='port',
                sample=f'{blocks} consecutive {frames}-frame blocks of the {voices}-voice C2 graph from position 0, '
                       f'{dt:.1f} s, oracle/chain_ref.py (numpy {np.__version__}, scipy butter+sosfilt per channel per block), '
                       f'host has {os.cpu_count()} logical cores')


def _cpu_worker(args):
    p, lo, hi, frames, blocks = args
    from oracle import chain_ref as R
    import warnings
    warnings.filterwarnings('ignore', category=DeprecationWarning)
    sl = slice(lo, hi)
    node = R.Binary('Gain', R.Filter('lp', R.Osc('Sine', R.Fixed(p['hertz'][:, sl]), R.Fixed(p['phase'][:, sl])),
                                     R.Fixed(p['cutoff'][:, sl])), R.Fixed(p['gain'][:, sl]))
    acc = 0.0
    for b in range(blocks):
        acc += float(R.sum_bus(R.render(node, b * frames, frames, hi - lo, RATE), p['pan'][:, sl]).sum())
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--prewarm-ms', type=float, default=250.0,
                    help='untimed: keep rendering batches for this long before the W warm-up steps, so that the GPU clocks '
                         'have settled under load (the first ~50 ms after idle run 15-30 % slower); 0 disables')
    ap.add_argument('--blocks', type=int, default=4096, help='256-frame blocks per batch (one step)')
    ap.add_argument('--voices', type=int, default=1024, help='voices per GPU')
    ap.add_argument('--frames', type=int, default=256)
    ap.add_argument('--position', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--materialised', action='store_true', help='headline = one kernel per node (no fusion)')
    ap.add_argument('--single-mode', action='store_true', help='skip the second (alternative schedule) measurement')
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
    V, N, K = args.voices, args.frames, args.blocks
    params = synth_params(V * world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(fuse: bool, steps: int, warmup: int) -> dict:
        """W untimed + exactly `steps` timed batches of this rank's 1024-voice graph (+ bus all-reduce)"""
        timer = None if args.no_kernel_timing else KernelTimer()
        renderer = parallel.ShardedRenderer(lambda lo, hi: build_graph(params, lo, hi), V * world, bus_channels=2,
                                            rate=RATE, timer=timer, fuse=fuse)
        assert (renderer.lo, renderer.hi) == (rank * V, (rank + 1) * V)
        pos = args.position
        pending = None

        def step():
            # (K*N, 2) f32 bus: local render, then the RCCL all-reduce of THIS batch is left in flight on
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
        if pending is not None:
            pending.wait()
        fence()
        dt = time.perf_counter() - t0
        runtime.check_status()
        assert torch.isfinite(bus).all()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device='cuda')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        samples = V * world * N * K * steps
        res = {'value': samples / dt / 1e6, 'ms_per_step': dt / steps * 1e3}
        if timer:
            summ = timer.summary()
            total_ms = sum(e['ms'] for e in summ.values())
            kernels = {}
            closed = fuse and steady_applies(params, rank * V, (rank + 1) * V, args.position,
                                             args.position + (warmup + steps) * N * K - 1, N)
            for name, e in summ.items():
                bpu = ALGO_BYTES.get(name.split('[')[0], 0)
                if name.split('[')[0] == 'fused_voice_bus':
                    vpl = fused_f64_ops_per_voice_sample('fused_voice_bus', V, N, K, steady=closed)[1]
                    tiles = -(-V // (64 * vpl))                     # one f64 partial per (voice tile, frame, channel), written and re-read
                    bpu = (tiles * 2 * 8 * 2 + 2 * 4) / V
                avg_ms = e['ms'] / e['calls']
                kernels[name] = {'calls': e['calls'], 'avg_ms': avg_ms, 'share': e['ms'] / total_ms,
                                 'algo_bytes_per_voice_sample': bpu,
                                 'algo_GBs': bpu * (e['units'] / e['calls']) / (avg_ms * 1e-3) / 1e9}
            dom = max(summ, key=lambda k: summ[k]['ms'])
            traffic = None
            tfile = ROOT / 'profiles' / 'traffic.json'
            if tfile.exists():
                traffic = json.loads(tfile.read_text()).get(dom.split('[')[0])
            res['roofline'] = {'bound': 'hbm', 'kernel': dom, 'achieved': kernels[dom]['algo_GBs'],
                               'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': kernels[dom]['algo_GBs'] / HBM_PEAK_GBS,
                               'traffic': traffic,
                               'algo_bytes_per_voice_sample': kernels[dom]['algo_bytes_per_voice_sample'],
                               'avg_launch_ms': kernels[dom]['avg_ms']}
            if dom.split('[')[0] in FUSED_KERNELS:
                # f64-VALU-bound by construction (the fused lower bound is 16/V B per voice-sample):
                # the meaningful ceiling is the f64 vector issue rate,
                # peak = 256 CU x 4 SIMD x 16 f64 lanes/clk x 2.4 GHz = 39.3 T instr-lanes/s (= 78.6 TFLOP/s FMA)
                ops, vpt, span = fused_f64_ops_per_voice_sample(dom.split('[')[0], V, N, K, steady=closed)
                ach = ops * (summ[dom]['units'] / summ[dom]['calls']) / (kernels[dom]['avg_ms'] * 1e-3) / 1e12
                res['roofline']['valu_f64'] = {'achieved': ach, 'peak': 39.3, 'unit': 'T f64-instr-lanes/s',
                                               'frac': ach / 39.3, 'f64_ops_per_voice_sample': ops,
                                               'voices_per_lane': vpt, 'blocks_per_lane': span,
                                               'path': 'closed form: steady-state sinusoid + homogeneous transient per '
                                                       'block, no warm-up rows (fused_steady_bus_kernel)' if closed
                                                       else 'span walker (fused_walk_kernel)'}
                res['roofline']['launches'] = ('avg_launch_ms brackets everything sig_fused_voice_bus enqueues: the chain '
                                               'kernel and the tile sum (sig_bus::partials_kernel), plus steady_prep_kernel '
                                               "on the calls where the per-voice constants change; the chain kernel's own "
                                               'duration is in profiles/*_kernel_stats.csv')
                res['roofline']['note'] = ('this kernel is f64-VALU-bound, not HBM-bound: see valu_f64; the HBM-bound '
                                           'node-materialised schedule is reported under alt_schedule')
            res['kernels'] = kernels
        return res

    main_mode = measure(fuse=not args.materialised, steps=args.steps, warmup=args.warmup)
    other = None
    if not args.single_mode:
        other = measure(fuse=args.materialised, steps=max(3, args.steps // 2), warmup=min(2, args.warmup))

    by_batch = None
    if world == 1 and not args.single_mode and not args.materialised:
        # the same schedule at smaller batches (BASELINE.md's throughput mode is K = 256), clocks already settled
        from signals_amd.engine import BatchRenderer
        by_batch = {}
        for k in (256, 1024):
            if k == K:
                continue
            r = BatchRenderer(build_graph(params, 0, V), 2, RATE)
            pos = 0
            for _ in range(10):
                r.render(pos, N, k); pos += N * k
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                r.render(pos, N, k); pos += N * k
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 200
            by_batch[str(k)] = {'Msamples_per_s': V * N * k / dt / 1e6, 'ms_per_step': dt * 1e3}

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

    if rank == 0:
        def describe(fused):
            return ('fused voice chain + bus: sig_fused_voice_bus (Sine->LowPass->Gain->SumBus in one chain launch plus a '
                    'fixed-order tile sum; no per-voice sample touches HBM)' if fused else
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
                       'parallelism': f'voices sharded {V}/GPU x{world}, RCCL reduce of the stereo bus to rank 0, async'},
        }
        for k in ('roofline', 'kernels'):
            if k in main_mode:
                line[k] = main_mode[k]
        if other is not None:
            other['schedule'] = describe(args.materialised)
            if not args.materialised:
                # SURVEY.md 8d's whole-graph view of the per-node schedule: 24 algorithmic bytes per voice-sample
                other['hbm_frac_at_24_bytes_per_voice_sample'] = other['value'] * 1e6 * 24 / (HBM_PEAK_GBS * 1e9)
            line['alt_schedule'] = other
        if by_batch:
            line['fused_schedule_at_other_batch_sizes'] = by_batch
        if latency is not None:
            line['latency_mode'] = latency
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline(params, V, N)
            # BASELINE.json's second metric: max-abs sample error of the rendered bus against float32(oracle float64)
            from signals_amd.engine import BatchRenderer
            ref = cpu_baseline.reference_bus.astype(np.float32).astype(np.float64)
            nb = ref.shape[0] // N
            errs = {}
            for label, fuse in (('fused', True), ('materialised', False)):
                got = BatchRenderer(build_graph(params, 0, V), 2, RATE, fuse=fuse).render(0, N, nb).double().cpu().numpy()
                errs[label] = float(np.max(np.abs(got - ref)))
            line['max_abs_error'] = dict(errs, bar=1e-6, full_scale=float(np.max(np.abs(ref))),
                                         sample=f'stereo bus of the first {nb} blocks of the {V}-voice graph vs the CPU oracle')
            workers = min(16, os.cpu_count() or 1)                      # the GPU box's CPU share for one GPU
            if workers > 1 and V % workers == 0:
                line['cpu_baseline_sharded'] = cpu_baseline_sharded(params, V, N, workers)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + '\n').encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
