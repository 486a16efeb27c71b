"""Full-size (BASELINE configs[1]: 1024 voices, 256-frame blocks, 256-block batches) checks through
size-independent properties, plus spot checks against the CPU oracle on sampled voices/blocks
(the oracle renders 1024 voices at ~3 blocks/s, so it cannot cover a whole batch)."""
import numpy as np
import pytest
import torch

import bench
from helpers import RATE, f32, maxerr

pytestmark = pytest.mark.gpu
V, N, K = 1024, 256, 256


@pytest.fixture(scope='module')
def params():
    assert torch.cuda.is_available()
    from signals_amd import runtime
    runtime.set_device('cuda:0')
    return bench.synth_params(V)


def render(params, fuse, position=0, k=K, lo=0, hi=V, channels=2, gain_scale=None, node=None):
    from signals_amd.engine import BatchRenderer
    p = params if gain_scale is None else dict(params, gain=params['gain'] * gain_scale)
    graph = bench.build_graph(p, lo, hi) if node is None else node
    return BatchRenderer(graph, channels, RATE, fuse=fuse).render(position, N, k)


def test_fused_and_materialised_schedules_agree(params):
    a = render(params, True).cpu().numpy()
    b = render(params, False).cpu().numpy()
    assert a.shape == (N * K, 2) and np.isfinite(a).all()
    assert maxerr(a, b) < 2e-7                       # bus values are O(0.03); each path is within 1e-7 of f64
    assert float(np.abs(a).max()) > 1e-3             # not trivially zero


def test_batching_is_invisible(params):
    """one 256-block batch == 4 consecutive 64-block batches == a batch started mid-stream: bit for bit on the
    per-node schedule, to one float32 ulp of the bus on the fused one (its Sine recurrence is seeded per span)"""
    from signals_amd.engine import BatchRenderer
    for fused in (True, False):
        whole = render(params, fused)
        r = BatchRenderer(bench.build_graph(params, 0, V), 2, RATE, fuse=fused)
        parts = torch.cat([r.render(i * 64 * N, N, 64) for i in range(4)])
        tail = render(params, fused, position=200 * N, k=56)
        if fused:
            assert float((whole.double() - parts.double()).abs().max()) < 4e-9
            assert float((whole[200 * N:].double() - tail.double()).abs().max()) < 4e-9
        else:
            assert torch.equal(whole, parts)
            assert torch.equal(whole[200 * N:], tail)


def test_bus_linearity_in_gain_and_voice_partition(params):
    """the graph is linear in the per-voice gains and additive over voice ranges (what sharding relies on)"""
    full = render(params, True, k=32).double()
    doubled = render(params, True, k=32, gain_scale=2.0).double()
    assert float((doubled - 2 * full).abs().max()) < 1e-7
    halves = render(params, True, k=32, lo=0, hi=512).double() + render(params, True, k=32, lo=512, hi=V).double()
    assert float((halves - full).abs().max()) < 1e-7


def test_spot_check_against_oracle(params):
    """8 sampled voices x blocks {0, 1, 137, 255}: per-voice Gain output vs the reference restatement"""
    from oracle import chain_ref as R
    from signals_amd.chain.fixed import Fixed
    from signals_amd.chain.fx import Gain, LowPass
    from signals_amd.chain.osc import Sine
    from signals_amd.engine import BatchRenderer

    def fixed(v):
        f = Fixed(); f.get_state().value = np.ascontiguousarray(v); return f
    o = Sine(); o.hertz = fixed(params['hertz']); o.phase = fixed(params['phase'])
    lp = LowPass(); lp.input = o; lp.cutoff = fixed(params['cutoff'])
    g = Gain(); g.left = lp; g.right = fixed(params['gain'])
    voices = np.array([0, 1, 255, 256, 511, 700, 1022, 1023])
    for fuse in (True, False):
        got = BatchRenderer(g, V, RATE, fuse=fuse).render(0, N, K)
        for b in (0, 1, 137, 255):
            sub = {k: params[k][:, voices] for k in ('hertz', 'phase', 'cutoff', 'gain')}
            ref = R.gain(R.filter_block('lp', lambda p, n: R.osc('Sine', p, n, RATE, sub['hertz'], sub['phase']),
                                        b * N, N, RATE, sub['cutoff']), sub['gain'])
            blk = got[b * N:(b + 1) * N][:, torch.from_numpy(voices).cuda()].cpu().numpy()
            assert maxerr(blk, f32(ref)) < 1e-9, (fuse, b)          # gains are ~1/1024: 1e-6 relative to full scale


def test_filter_output_is_bounded_and_blocks_are_cold_started(params):
    """every block starts from zero state 100 frames early: shifting the stream by one block shifts the output"""
    from signals_amd.chain.fixed import Fixed
    from signals_amd.chain.fx import LowPass
    from signals_amd.chain.osc import Sine
    from signals_amd.engine import BatchRenderer

    def fixed(v):
        f = Fixed(); f.get_state().value = np.ascontiguousarray(v); return f
    o = Sine(); o.hertz = fixed(params['hertz']); o.phase = fixed(params['phase'])
    lp = LowPass(); lp.input = o; lp.cutoff = fixed(params['cutoff'])
    a = BatchRenderer(lp, V, RATE).render(N, N, 64)             # blocks 1..64
    b = BatchRenderer(lp, V, RATE).render(0, N, 65)[N:]         # same blocks inside a longer batch
    assert float((a.double() - b.double()).abs().max()) < 1.2e-7   # one float32 ulp: the walker's spans differ
    a = BatchRenderer(lp, V, RATE, fuse=False).render(N, N, 64)
    b = BatchRenderer(lp, V, RATE, fuse=False).render(0, N, 65)[N:]
    assert torch.equal(a, b)                                    # the per-node path is bit-for-bit position-pure
    assert float(a.abs().max()) < 1.2                           # Butterworth overshoot only


def test_span_walker_is_bit_identical_to_the_plain_biquad_kernel(params):
    """K=512 x 1024 voices makes sig_biquad_coldstart pick the span walker (every input row read once, two
    live chains); 64-block batches use the plain kernel.  Same chains, same order: bitwise equal."""
    from signals_amd.chain.fixed import Fixed
    from signals_amd.chain.fx import HighPass
    from signals_amd.chain.osc import Sawtooth
    from signals_amd.engine import BatchRenderer

    def fixed(v):
        f = Fixed(); f.get_state().value = np.ascontiguousarray(v); return f

    def build():
        o = Sawtooth(); o.hertz = fixed(params['hertz']); o.phase = fixed(params['phase'])
        hp = HighPass(); hp.input = o; hp.cutoff = fixed(params['cutoff'])
        return hp
    for pos in (0, 37):                                   # 37: the first block's context is short (c = 37)
        whole = BatchRenderer(build(), V, RATE, fuse=False).render(pos, N, 512)
        r = BatchRenderer(build(), V, RATE, fuse=False)
        parts = torch.cat([r.render(pos + i * 64 * N, N, 64) for i in range(8)])
        assert torch.equal(whole, parts), pos


def test_c4_eight_shards_of_1024_voices_sum_to_the_unsharded_bus():
    """BASELINE config 4 on one GPU: the 8192-voice graph rendered as 8 shards of 1024 voices (what 8 ranks do,
    `shard_voices`), buses summed in rank order like the RCCL reduce, against the same graph rendered unsharded"""
    from signals_amd.engine import BatchRenderer
    from signals_amd.parallel import shard_voices
    total, world, k = 8192, 8, 16
    p = bench.synth_params(total)
    whole = BatchRenderer(bench.build_graph(p, 0, total), 2, RATE).render(0, N, k).double()
    acc = torch.zeros_like(whole)
    for rank in range(world):
        lo, hi = shard_voices(total, world, rank)
        assert hi - lo == 1024
        acc += BatchRenderer(bench.build_graph(p, lo, hi), 2, RATE).render(0, N, k).double()
    assert float((acc - whole).abs().max()) < 1e-7          # summation order differs: rounding-level agreement
    assert float(whole.abs().max()) > 1e-4


def test_one_hour_stream_stays_finite_and_position_pure(params):
    """render a whole hour of the 1024-voice graph in 1024-block batches (172.8 M frames, 177 G voice-samples);
    every batch finite, and blocks met in-stream equal the same blocks rendered from a cold renderer (to one
    float32 ulp of the bus: the fused walker seeds its Sine recurrence once per span of blocks, so the launch
    geometry shows up at the 1e-13 level before the float32 rounding)"""
    from signals_amd.engine import BatchRenderer
    k = 1024
    r = BatchRenderer(bench.build_graph(params, 0, V), 2, RATE)
    pos, checks, peak = 0, {}, 0.0
    probe_at = {0, 300, 659}
    for step in range(660):                                   # 660 * 262144 frames = 60.08 min
        bus = r.render(pos, N, k)
        if step % 60 == 0 or step in probe_at:
            assert bool(torch.isfinite(bus).all()), step
            peak = max(peak, float(bus.abs().max()))
        if step in probe_at:
            checks[pos] = bus[:4 * N].clone()
        pos += N * k
    assert pos >= 172_800_000 and 1e-3 < peak < 1.0
    for p0, want in checks.items():
        cold = BatchRenderer(bench.build_graph(params, 0, V), 2, RATE)
        cold.scan_max_chains = 0                              # same (serial, bus-fused) kernel as the big batches
        assert float((cold.render(p0, N, 4).double() - want.double()).abs().max()) < 4e-9, p0
        latency = BatchRenderer(bench.build_graph(params, 0, V), 2, RATE).render(p0, N, 4)      # scan chain + bus launch
        assert float((latency.double() - want.double()).abs().max()) < 1e-8, p0


def test_config3_full_size_schedules_agree_and_spot_check():
    """BASELINE config 3 at its own size (1024 voices, 1024-frame blocks): the engine's default schedule (the whole voice in
    one launch, sig_fused_cascade_bus) against the one-kernel-per-node schedule over a continued stream -- two GPU
    schedules, each within 1e-6 of full scale of the f64 reference, so they agree to twice that -- and 6 sampled voices of
    the enveloped cascade WITHOUT the bus (fused saw + filter, then filter x envelope) against the oracle.  The bus kernel
    itself against the oracle at this size, deep in the stream: tests/test_gpu_parity_holes.py"""
    import sys, pathlib
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / 'tools'))
    import measure_configs as mc
    from oracle import chain_ref as R
    from signals_amd.engine import BatchRenderer
    Vc, Nc, Kc = 1024, 1024, 12
    outs = {}
    for fuse in (True, False):
        bus, channels, n, _, _ = mc.c3(Vc)
        assert (channels, n) == (1, Nc)
        r = BatchRenderer(bus, 1, RATE, fuse=fuse)
        outs[fuse] = torch.cat([r.render(0, Nc, Kc), r.render(Nc * Kc, Nc, 4)]).double()       # a second batch continues
    assert bool(torch.isfinite(outs[True]).all()) and float(outs[True].abs().max()) > 1.0
    full = float(outs[False].abs().max())
    assert float((outs[True] - outs[False]).abs().max()) < 2e-6 * full    # full scale ~8.7: two schedules, 1e-6 of it each
    # per-voice spot check: rebuild the graph's parameters exactly as measure_configs.c3 draws them
    rng = np.random.default_rng(0)
    hz, ph = rng.uniform(55, 1760, (1, Vc)), rng.uniform(0, 1, (1, Vc))
    c1, c2 = rng.uniform(200, 8000, (1, Vc)), rng.uniform(200, 8000, (1, Vc))
    env = {k: rng.uniform(lo, hi, (1, Vc)) for k, (lo, hi) in dict(attack=(0.001, 0.05), decay=(0.01, 0.2), sustain=(0.2, 0.9),
                                                                  release=(0.05, 0.5), gate_on=(0.0, 0.5), gate_off=(1.0, 4.0)).items()}
    voices = np.array([0, 1, 511, 512, 1022, 1023])
    sub = lambda a: a[:, voices]
    f2 = R.Filter('lp', R.Filter('lp', R.Osc('Sawtooth', R.Fixed(sub(hz)), R.Fixed(sub(ph))), R.Fixed(sub(c1))), R.Fixed(sub(c2)))
    ref = R.render_stream(R.Binary('RingMod', f2, R.Adsr(**{k: sub(v) for k, v in env.items()})), 0, Nc, 3, len(voices))
    bus, *_ = mc.c3(Vc)
    rm = bus.input.sig
    got = BatchRenderer(rm, Vc, RATE).render(0, Nc, 3)[:, torch.from_numpy(voices).cuda()].cpu().numpy()
    assert maxerr(got, f32(ref)) < 1e-6


def test_config5_full_size_fused_equals_two_launches():
    """BASELINE config 5 at its own size (4096 voices, 256-frame blocks, 64 blocks): chain + mix matrix in one launch
    is bit-identical to the per-node schedule's chain followed by sig_mix_matrix when both see the same float32 rows;
    against the per-node engine schedule (whose chain rounds through float32 earlier) to rounding"""
    import sys, pathlib
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / 'tools'))
    import measure_configs as mc
    from signals_amd.engine import BatchRenderer
    outs = {}
    for fuse in (True, False):
        mm, channels, n, k, _ = mc.c5(4096, 64)
        outs[fuse] = BatchRenderer(mm, channels, RATE, fuse=fuse).render(0, n, k)
    assert outs[True].shape == (256 * 64, 4096) and bool(torch.isfinite(outs[True]).all())
    full = float(outs[False].abs().max())
    assert float((outs[True].double() - outs[False].double()).abs().max()) < min(2e-6, 1e-6 * full)     # (vs the oracle: test_gpu_parity_holes.py)
    assert full > 0.5


def test_bench_geometry_vs_oracle(params):
    """the EXACT launch bench.py times -- 1024 voices, 4096 blocks of 256 frames per batch: sig_fused_voice_bus picks the
    Sine closed form at 8 voices x 8 blocks per lane (fused_steady_bus_kernel<8, 2>, two voice tiles) -- compared with
    the CPU oracle on whole blocks of its own output: the first four, one mid-stream, the last; then the same geometry
    one batch further into the stream (the homogeneous state of every block comes from T_100, none from T_c0)"""
    from signals_amd import _native
    from signals_amd.engine import BatchRenderer, KernelTimer
    k = 4096
    plan = _native.fused_voice_bus_plan('Sine', 0, V, N, k, 100)
    assert plan == {'voices_per_lane': 8, 'blocks_per_lane': 8, 'closed_form': True, 'kernel': 'fused_steady_bus_kernel<8, C>'}
    timer = KernelTimer()
    r = BatchRenderer(bench.build_graph(params, 0, V), 2, RATE, timer=timer)
    for start in (0, k * N):
        assert bench.steady_applies(params, 0, V, start, start + k * N - 1, N)       # every wave takes the closed form
        bus = r.render(start, N, k)
        torch.cuda.synchronize()
        assert set(timer.summary()) == {'fused_voice_bus[Sine,lp,gain]'}, set(timer.summary())
        assert bus.shape == (k * N, 2)
        per_block, scale = bench.check_batch_against_oracle(params, V, N, bus, start, k, [0, 1, 2, 3, k // 2, k - 1])
        assert len(per_block) == 6 and scale > 1e-3
        assert max(per_block.values()) < 1e-6 * max(1.0, scale), per_block
        assert max(per_block.values()) < 5e-9, per_block            # in fact one float32 ulp of the 0.03 full-scale bus
