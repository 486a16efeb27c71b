"""Oracle checks of launches that earlier rounds only compared with other GPU schedules, or only inside bench.py:

  * the Sine closed form (fused_steady_bus_kernel) with EVERY voice between 2^24 and 2^26 oscillator cycles -- the band its
    phase-range limit was extended to (sig_osc.h: kSineFastMaxT) -- against the oracle, bus and single voices;
  * the fused cascade (sig_fused_cascade_bus) at BASELINE config 3's own size, 4 voices x 4 blocks per lane, DEEP in the
    stream and across a continuing batch: all but six bus weights are exactly zero (the kernel still walks all 1024 voices),
    and the oracle renders those six voices sequentially from position 0 (cascades depend on the render history,
    chain/__init__.py:431-442, fx.py:85-106);
  * BASELINE config 5 at its own size (4096 x 256 x 256) against the oracle on whole blocks, both matrix sinks;
  * the sharded renderer in two processes on one GPU (gloo; RCCL refuses two ranks on one device): the reduced bus of two
    512-voice shards against the unsharded render, MixMatrix groups included.
"""
import os
import pathlib
import subprocess
import sys

import numpy as np
import pytest
import torch

import bench
import bench_configs as cfg
from helpers import HOUR, RATE, f32, maxerr

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent
CTX = 100


@pytest.fixture(scope='module', autouse=True)
def _device():
    assert torch.cuda.is_available()
    from signals_amd import _native, runtime
    runtime.set_device('cuda:0')
    yield
    _native.set_fused_tuning()
    _native.set_fused_cascade_tuning()


def dev(a):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float64, device='cuda')


# ------------------------------------------------------------------------------------------------ (a) 2^24 .. 2^26 cycles
K_BAND, N_BAND = 8, 256
BAND_POSITIONS = [5 * HOUR, int(9.5 * HOUR), int(2 ** 26 / 1760 * 48000) - K_BAND * N_BAND - 1]


@pytest.mark.parametrize('pos', BAND_POSITIONS)
def test_closed_form_with_every_voice_between_2p24_and_2p26_cycles(pos):
    """hertz 1000-1759 Hz at 5 h, 9.5 h and right under the limit: |t| of every voice lies in [2^24, 2^26) over the whole
    launch, the closed form takes every wave (host mirror of the kernel's per-wave test + the launch plan), and its bus
    is within 1e-6 of full scale of the oracle; four voices singled out through one-hot bus weights within 1e-6 each"""
    from oracle import chain_ref as R
    from signals_amd import _native
    V, N, K = 256, N_BAND, K_BAND
    rng = np.random.default_rng(101 + pos % 13)
    th = rng.uniform(0, np.pi / 2, V)
    p = dict(hertz=rng.uniform(1000.0, 1759.0, (1, V)), phase=rng.uniform(0, 1, (1, V)), cutoff=rng.uniform(200, 8000, (1, V)),
             gain=rng.uniform(0.1, 1.0, (1, V)), pan=np.stack([np.cos(th), np.sin(th)]))
    singled = np.array([0, 77, 130, 255])
    p['gain'][0, singled] = 1.0                                             # unit scale for the single-voice check
    t_first, t_last = pos / RATE * p['hertz'] + p['phase'], (pos + K * N - 1) / RATE * p['hertz'] + p['phase']
    assert (t_first >= 2.0 ** 24).all() and (t_last < 2.0 ** 26).all()
    assert bench.steady_applies(p, 0, V, pos, pos + K * N - 1, N)           # every wave qualifies for the closed form
    btype = 'hp' if pos == BAND_POSITIONS[1] else 'lp'
    src = lambda q, n: R.osc('Sine', q, n, RATE, p['hertz'], p['phase'])
    ref = np.concatenate([R.gain(R.filter_block(btype, src, pos + b * N, N, RATE, p['cutoff']), p['gain']) for b in range(K)])
    ref_bus = ref @ p['pan'].T
    scale = float(np.abs(ref_bus).max())
    assert scale > 2.0                                                      # sums of 256 voices of O(1), not 1/V weights
    onehot = np.zeros((4, V))
    onehot[np.arange(4), singled] = 1.0

    def run(pan, C):
        out = torch.full((K * N, C), float('nan'), device='cuda')
        _native.fused_voice_bus('Sine', btype, RATE, pos, N, K, CTX, V, dev(p['hertz']), dev(p['phase']), dev(p['cutoff']),
                                dev(p['gain']), dev(pan), out)
        return out.cpu().numpy()
    for vpt, span in ((8, 8), (4, 2), (1, 1)):
        _native.set_fused_tuning(vpt, span, 1, 0)
        plan = _native.fused_voice_bus_plan('Sine', pos, V, N, K, CTX)
        assert plan['closed_form'] and (plan['voices_per_lane'], plan['blocks_per_lane']) == (vpt, span), plan
        bus = run(p['pan'], 2)
        assert np.isfinite(bus).all()
        assert maxerr(bus, f32(ref_bus)) < 1e-6 * scale, (pos, vpt, span)
        one = run(onehot, 4)
        assert maxerr(one, f32(ref[:, singled])) < 1e-6, (pos, vpt, span)   # single voices, full scale <= 1.2


# ------------------------------------------------------------------------------------------------ (b) C3 at full size
def test_config3_fused_cascade_full_size_deep_in_the_stream_vs_oracle():
    from oracle import chain_ref as R
    from signals_amd import _native
    from signals_amd.engine import BatchRenderer, KernelTimer
    V, N, K = 1024, 1024, 1024
    p = cfg.c3_params(V)
    # config 3's envelopes end within 4.5 s (block 211); keep voices sounding 20-40 s so that deep blocks are not silence
    rng = np.random.default_rng(5)
    p['env'] = dict(p['env'], gate_off=rng.uniform(5.0, 40.0, (1, V)))
    voices = np.array([0, 1, 511, 512, 1022, 1023])
    p['env']['gate_off'][0, voices] = [12.0, 21.5, 22.0, 25.0, 30.0, 40.0]
    weights = np.zeros((1, V))
    weights[0, voices] = 1.0
    bus = cfg.c3_graph(p)
    bus.get_state().gains = weights                                         # exactly 0 for all but six voices
    assert _native.fused_cascade_geometry(V, K) == (4, 4)
    timer = KernelTimer()
    r = BatchRenderer(bus, 1, RATE, timer=timer)
    first = r.render(0, N, K)
    second = r.render(N * K, N, K)                                          # continues: history = the previous batch's last block
    torch.cuda.synchronize()
    assert set(timer.summary()) == {'fused_cascade_bus[Sawtooth,lp,lp,env]'}, set(timer.summary())
    assert timer.summary()['fused_cascade_bus[Sawtooth,lp,lp,env]']['calls'] == 2
    sub = lambda a: np.ascontiguousarray(a[:, voices])
    q = dict(hertz=sub(p['hertz']), phase=sub(p['phase']), cut1=sub(p['cut1']), cut2=sub(p['cut2']),
             env={k: sub(v) for k, v in p['env'].items()})
    nblocks = K + 4
    ref = R.render_stream(cfg.c3_oracle(q), 0, N, nblocks, len(voices)).sum(axis=1, keepdims=True)
    scale = float(np.abs(ref).max())
    assert scale > 1.0
    got1, got2 = first.cpu().numpy(), second.cpu().numpy()
    for b in (0, 1, 2, 517, 1023):
        blk = ref[b * N:(b + 1) * N]
        assert np.abs(blk).max() > 0.05, b                                  # the voices are sounding there
        assert maxerr(got1[b * N:(b + 1) * N], f32(blk)) < 1e-6 * scale, b
    for b in (0, 1, 3):
        blk = ref[(K + b) * N:(K + b + 1) * N]
        assert np.abs(blk).max() > 0.05, b
        assert maxerr(got2[b * N:(b + 1) * N], f32(blk)) < 1e-6 * scale, ('second batch', b)


# ------------------------------------------------------------------------------------------------ (c) C5 at full size
def test_config5_full_size_vs_oracle_both_sinks():
    """4096 voices x 256 frames x 256 blocks through fused_steady_mix_kernel, whole blocks 0 and 255 against the oracle:
    the default sink (each float32 as three bfloat16, sig_mix_tile.h) and the float32 MFMA instruction"""
    from oracle import chain_ref as R
    from signals_amd import _native
    from signals_amd.engine import BatchRenderer, KernelTimer
    V, N, K = 4096, 256, 256
    p = cfg.c5_params(V)
    oracle = cfg.c5_oracle(p)
    ref = {b: R.render(oracle, b * N, N, V, RATE) for b in (0, K - 1)}
    scale = max(float(np.abs(v).max()) for v in ref.values())
    assert scale > 1.0
    for sink in (1, 3):                                                     # 3: v_mfma_f32_32x32x2_f32 (tuning hook)
        _native.set_fused_tuning(0, 0, sink, -1)
        timer = KernelTimer()
        out = BatchRenderer(cfg.c5_graph(p), V, RATE, timer=timer).render(0, N, K)
        torch.cuda.synchronize()
        assert set(timer.summary()) == {'fused_osc_biquad_mix[Sine,lp]'}, set(timer.summary())
        assert out.shape == (N * K, V)
        for b, want in ref.items():
            got = out[b * N:(b + 1) * N].cpu().numpy()
            assert np.isfinite(got).all()
            assert maxerr(got, f32(want)) < 1e-6 * scale, (sink, b)
        del out
    _native.set_fused_tuning()


# ------------------------------------------------------------------------------------------------ (e) two ranks, one GPU
SHARD_SCRIPT = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
import torch
import torch.distributed as dist
import bench
import bench_configs as cfg
from signals_amd import parallel, runtime
from signals_amd.engine import BatchRenderer
runtime.set_device('cuda:0')
rank, world = parallel.init_process_group()
assert dist.is_initialized() and dist.get_backend() == 'gloo' and world == 2
V, N, K = 1024, 256, 24
p = bench.synth_params(V)
p['gain'] = p['gain'] * V / 32.0                       # bus of O(1), so that 1e-7 is a float32 ulp of it
r = parallel.ShardedRenderer(lambda lo, hi: bench.build_graph(p, lo, hi), V, bus_channels=2)
assert (r.hi - r.lo) == V // 2 and r.lo == rank * (V // 2)
a = r.render(0, N, K, dst=0).clone()                    # reduce to rank 0
b = r.render(N * K, N, K).clone()                       # continuing batch, all-reduce

def mixed(q, lo, hi):
    """Sine -> LowPass -> MixMatrix -> SumBus(stereo): 64-voice matrix groups must not straddle a shard"""
    from signals_amd.chain import ext
    mm = cfg.c5_graph(dict(hertz=q['hertz'][:, lo:hi], phase=q['phase'][:, lo:hi], cutoff=q['cutoff'][:, lo:hi], matrix=q['matrix']))
    bus = ext.SumBus(); bus.input = mm
    bus.get_state().gains = np.ascontiguousarray(q['pan'][:, lo:hi])
    return bus
Vm = 640                                                # 10 groups of 64: 5 per rank
q = cfg.c5_params(Vm)
th = np.random.default_rng(9).uniform(0, np.pi / 2, Vm)
q['pan'] = np.stack([np.cos(th), np.sin(th)]) / 8.0
rm = parallel.ShardedRenderer(lambda lo, hi: mixed(q, lo, hi), Vm, bus_channels=2, group=64)
assert (rm.lo, rm.hi) == (rank * 320, rank * 320 + 320)
c = rm.render(0, N, 6).clone()
torch.cuda.synchronize()
if rank == 0:
    whole = BatchRenderer(bench.build_graph(p, 0, V), 2, 48000)
    wa, wb = whole.render(0, N, K), whole.render(N * K, N, K)
    wc = BatchRenderer(mixed(q, 0, Vm), 2, 48000).render(0, N, 6)
    for name, got, want in (('reduce', a, wa), ('all_reduce', b, wb), ('mix_matrix', c, wc)):
        full = float(want.abs().max())
        err = float((got.double() - want.double()).abs().max())
        assert 0.05 < full < 50.0, (name, full)
        ulp = float(np.spacing(np.float32(full)))                  # the shards' float32 buses are summed in another order: one ulp of full scale
        assert err < max(1e-7, 1.01 * ulp), (name, err, full)
        print('SHARD', name, 'err', err, 'full_scale', full)
else:
    for got in (b,):                                    # the all-reduced bus is on every rank
        assert bool(torch.isfinite(got).all())
dist.barrier()
dist.destroy_process_group()
print('SHARD_OK', rank)
'''


def test_two_ranks_render_their_shards_and_the_reduced_bus_equals_the_unsharded_render():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):                                                   # started before this process's children touch the GPU
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', SIG_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0')
        env.pop('SIG_FORCE_DIST', None)
        procs.append(subprocess.Popen([sys.executable, '-c', SHARD_SCRIPT.format(root=str(ROOT))], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=900))
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for rank, (pr, (so, se)) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0, (rank, so[-2000:], se[-4000:])
        assert f'SHARD_OK {rank}' in so
    assert outs[0][0].count('SHARD ') == 3, outs[0][0]
