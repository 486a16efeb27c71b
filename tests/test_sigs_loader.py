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
    assert isinstance(p['3c'], sigs.Tap) and p['3c'].original_cls_name == 'signals.chain.vis.Wave'
    assert p['3c'].original_state['min_amp'] == -1.0
    sink = p['4c']
    assert isinstance(sink, BlockDriver) and sink.input.sig is p['3c'] and list(p.sinks.values()) == [sink]


def test_lowpass_test_patch_topology():
    p = sigs.load(DATA / 'lowpass_test.sigs')
    lp, gain, tri, merge = p['4b'], p['3a'], p['2a'], p['5a']
    assert isinstance(lp, fx.LowPass) and isinstance(gain, fx.Gain) and isinstance(tri, osc.Triangle)
    assert isinstance(merge, shape.Merge) and merge.left.sig is lp and merge.right.sig is gain and merge.channels == 2
    assert lp.input.sig is gain and lp.cutoff.sig is p['1c'] and gain.left.sig is tri and gain.right.sig is p['1b']
    assert p['7a'].input.sig is p['6a'] and p['6a'].input.sig is p['5c'] and p['5c'].input.sig is lp
    assert p['5c'].original_state['path'] == '/tmp/lowpass_test.wav'


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
    assert np.array_equal(out, eager)


@pytest.mark.gpu
def test_example_sine_script():
    import importlib.util
    spec = importlib.util.spec_from_file_location('example_sine', DATA.parent.parent / 'scripts' / 'example_sine.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(['500', '-a', '0.2', '-n', '4']) < 1e-6
