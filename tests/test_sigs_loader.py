"""`.sigs` patch loader: the reference's two fixture patches load verbatim
(tests/data/*.sigs are byte-identical copies of the DATA files src/signals/{vis_test,lowpass_test}.sigs)."""
import pathlib

import numpy as np
import pytest
import torch

from signals_amd.chain import fx, osc, shape
from signals_amd.chain import sigs
from signals_amd.chain.driver import BlockDriver
from signals_amd.chain.fixed import Fixed

DATA = pathlib.Path(__file__).resolve().parent / 'data'


def test_coordinates_and_values():
    # reference doctests: map/__init__.py:80-95, :110-122
    for text, rc in (('1a', (1, 1)), ('1b', (1, 2)), ('1z', (1, 26)), ('1aa', (1, 27)), ('1az', (1, 52)),
                     ('1zz', (1, 702)), ('1234aul', (1234, 1234))):
        assert sigs.parse_coordinates(text) == rc
    with pytest.raises(sigs.PatchError):
        sigs.parse_coordinates('a1')
    assert sigs.parse_value('1') == 1 and sigs.parse_value('true') is True and sigs.parse_value('-1.0') == -1.0
    v = sigs.parse_value('[[1, 2, 3]]')
    assert isinstance(v, np.ndarray) and v.dtype == np.int64 and v.shape == (1, 3)
    assert sigs.parse_value('/tmp/lowpass_test.wav') == '/tmp/lowpass_test.wav'


def test_vis_test_patch_topology():
    p = sigs.load(DATA / 'vis_test.sigs')
    assert isinstance(p['1c'], Fixed) and p['1c'].get_state().value.dtype == np.int64
    assert isinstance(p['2c'], osc.Sine) and p['2c'].hertz.sig is p['1c']
    from signals_amd.chain import ext, vis
    assert type(p['3c']) is vis.Wave and isinstance(p['3c'], ext.Tap) and p['3c'].cls_name() == 'signals.chain.vis.Wave'
    assert p['3c'].get_state().min_amp == -1.0 and p['3c'].get_state().max_amp == 1.0
    sink = p['4c']
    assert isinstance(sink, BlockDriver) and sink.input.sig is p['3c'] and list(p.sinks.values()) == [sink]


def test_lowpass_test_patch_topology():
    p = sigs.load(DATA / 'lowpass_test.sigs')
    lp, gain, tri, merge = p['4b'], p['3a'], p['2a'], p['5a']
    assert isinstance(lp, fx.LowPass) and isinstance(gain, fx.Gain) and isinstance(tri, osc.Triangle)
    assert isinstance(merge, shape.Merge) and merge.left.sig is lp and merge.right.sig is gain and merge.channels == 2
    assert lp.input.sig is gain and lp.cutoff.sig is p['1c'] and gain.left.sig is tri and gain.right.sig is p['1b']
    assert p['7a'].input.sig is p['6a'] and p['6a'].input.sig is p['5c'] and p['5c'].input.sig is lp
    from signals_amd.chain.files import FileWriter
    assert isinstance(p['5c'], FileWriter) and p['5c'].get_state().path == '/tmp/lowpass_test.wav'


def test_loader_errors():
    with pytest.raises(sigs.PatchError):
        sigs.loads('+ 1a signals.chain.osc.Sine\n+ 1a signals.chain.osc.Sine')
    with pytest.raises(sigs.PatchError):
        sigs.loads('+ 1a signals.chain.osc.Sine\n+ 2a signals.chain.fixed.Fixed\n> 2a 1a.nope')
    with pytest.raises(sigs.PatchError):
        sigs.loads('+ 1a signals.chain.osc.Sine volume=3')
    with pytest.raises(sigs.PatchError):
        sigs.loads('rm 1a')
    from signals_amd.chain import BadStateValue
    with pytest.raises(BadStateValue):
        sigs.loads('+ 1a signals.chain.fixed.Fixed value=3')


@pytest.mark.gpu
def test_patches_render_like_the_reference(golden):
    assert torch.cuda.is_available()
    from signals_amd import runtime
    runtime.set_device('cuda:0')
    g = golden('sigs')
    f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)
    out = sigs.load(DATA / 'vis_test.sigs')['4c'].render(3)
    assert out.shape == (768, 1) and np.max(np.abs(out - f32(g['sigs/vis_test']))) < 1.5e-7
    out = sigs.load(DATA / 'lowpass_test.sigs')['7a'].render(3)
    assert np.max(np.abs(out - f32(g['sigs/lowpass_test'][:, :1]))) < 2e-7
    eager = sigs.load(DATA / 'lowpass_test.sigs')['7a'].render(3, batched=False)
    assert np.max(np.abs(eager - f32(g['sigs/lowpass_test'][:, :1]))) < 2e-7
    assert np.max(np.abs(out - eager)) < 2e-7            # the engine fuses Triangle -> Gain -> LowPass (f64 between the nodes)
    # ... into ONE launch: the Gain in front of the filter is folded into the fused chain's output weight
    from signals_amd.engine import BatchRenderer, KernelTimer
    patch = sigs.load(DATA / 'lowpass_test.sigs')
    timer = KernelTimer()
    again = BatchRenderer(patch['7a'].input.sig, 1, 48000, timer=timer).render(0, 256, 3).cpu().numpy()
    torch.cuda.synchronize()
    assert set(timer.summary()) == {'fused_osc_biquad[Triangle,lp,gain]'}, set(timer.summary())
    assert np.array_equal(again, out)
    # the patch's Merge reads the Gain too (4b -> 5a.left, 3a -> 5a.right): its rows are a launch of their own (Osc x Gain as one voice program)
    timer = KernelTimer()
    both = BatchRenderer(patch['5a'], 2, 48000, timer=timer).render(0, 256, 3).cpu().numpy()
    torch.cuda.synchronize()
    assert set(timer.summary()) == {'fused_osc_biquad[Triangle,lp,gain]', 'voice_program[Osc,Gain]'}, set(timer.summary())
    assert np.max(np.abs(both - f32(g['sigs/lowpass_test']))) < 2e-7


@pytest.mark.gpu
def test_example_sine_script():
    import importlib.util
    spec = importlib.util.spec_from_file_location('example_sine', DATA.parent.parent / 'scripts' / 'example_sine.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(['500', '-a', '0.2', '-n', '4']) < 1e-6


def test_wave_file_roundtrip_on_host(tmp_path):
    """the WAV container itself (no GPU): PCM_16 quantisation, FLOAT lossless, positional writes"""
    import wave
    from signals_amd.chain.files import _WaveFile
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, (1000, 2))
    for subtype, tol in (('PCM_16', 5e-5), ('FLOAT', 1e-7)):      # write x32767, read /32768 (libsndfile's convention)
        path = tmp_path / f'{subtype}.wav'
        f = _WaveFile(path, 'w', 48000, 2, subtype)
        f.write(500, x[500:])                      # out of order: tail first (gap is silence)
        f.write(0, x[:500])
        f.close()
        r = _WaveFile(path, 'r')
        assert (r.samplerate, r.channels, r.frames, r.subtype) == (48000, 2, 1000, subtype)
        assert np.max(np.abs(r.read(0, 1000) - x)) <= tol
        assert r.read(900, 256).shape == (100, 2)  # short read at the end
        r.close()
    with wave.open(str(tmp_path / 'PCM_16.wav')) as w:      # a standard reader agrees with the header
        assert (w.getnchannels(), w.getframerate(), w.getnframes(), w.getsampwidth()) == (2, 48000, 1000, 2)


@pytest.mark.gpu
def test_file_taps_on_gpu(tmp_path, golden):
    from signals_amd import runtime
    runtime.set_device('cuda:0')
    from signals_amd.chain.files import FileReader, FileWriter
    from helpers import mkosc, stream
    g = golden('c2')
    path = tmp_path / 'take.wav'
    w = FileWriter(); w.get_state().path = str(path); w.get_state().subtype = 'FLOAT'
    w.input = mkosc('Sine', g['c2/hertz'][:, :2], g['c2/phase'][:, :2])
    d = BlockDriver(); d.get_state().channels = 2; d.input = w
    batched = d.render(4)                                   # engine path: one write of 1024 rows
    w.destroy()
    r = FileReader(); r.get_state().path = str(path)
    back = stream(r, 0, 256, 4, 2)                          # eager path: 4 positional reads
    assert np.array_equal(back, batched)
    d2 = BlockDriver(); d2.get_state().channels = 2; d2.input = r
    assert np.array_equal(d2.render(4), batched)            # and the batched reader
    w2 = FileWriter(); w2.get_state().path = str(tmp_path / 'pcm.wav')
    w2.input = mkosc('Sine', g['c2/hertz'][:, :2], g['c2/phase'][:, :2])
    eager = stream(w2, 0, 256, 4, 2)                        # pass-through result is the input, untouched
    assert np.array_equal(eager, batched)
    w2.destroy()
    r2 = FileReader(); r2.get_state().path = str(tmp_path / 'pcm.wav')
    assert np.max(np.abs(stream(r2, 0, 256, 4, 2) - batched)) < 5e-5      # 16-bit quantisation
