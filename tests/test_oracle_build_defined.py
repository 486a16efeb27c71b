"""CPU checks of the build-defined nodes' definitions in the oracle (parity
unpinned by the reference; these restatements ARE the definitions the HIP kernels are held to)."""
import numpy as np

from oracle import chain_ref as R


def test_sum_bus_definition():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((50, 12))
    assert np.allclose(R.sum_bus(x), x.sum(axis=1, keepdims=True)) and R.sum_bus(x).shape == (50, 1)
    g = rng.standard_normal((2, 12))
    assert np.allclose(R.sum_bus(x, g), np.stack([(x * g[0]).sum(1), (x * g[1]).sum(1)], axis=1))


def test_mix_matrix_definition():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((10, 128))
    m = rng.standard_normal((64, 64))
    out = R.mix_matrix(x, m)
    assert np.allclose(out[:, :64], x[:, :64] @ m) and np.allclose(out[:, 64:], x[:, 64:] @ m)


def test_adsr_shape_of_envelope():
    rows = dict(attack=[[0.01]], decay=[[0.02]], sustain=[[0.5]], release=[[0.05]], gate_on=[[0.1]], gate_off=[[0.3]])
    env = R.adsr(0, 24000, 48000, **rows)[:, 0]
    t = np.arange(24000) / 48000
    assert np.all(env[t < 0.1] == 0)
    assert abs(env[np.searchsorted(t, 0.105)] - 0.5) < 1e-3                    # half-way up the attack
    assert abs(env[np.searchsorted(t, 0.11)] - 1.0) < 1e-9                     # peak
    assert np.allclose(env[(t > 0.131) & (t < 0.3)], 0.5)                      # sustain
    assert abs(env[np.searchsorted(t, 0.325)] - 0.25) < 1e-3                   # half-way down the release
    assert np.all(env[t > 0.351] == 0)
    # position-pure: a block rendered at an offset equals the slice
    assert np.array_equal(R.adsr(5000, 300, 48000, **rows)[:, 0], env[5000:5300])
    # zero-length stages count as complete
    z = dict(rows, attack=[[0.0]], decay=[[0.0]])
    assert R.adsr(0, 24000, 48000, **z)[np.searchsorted(t, 0.1001), 0] == 0.5


def test_band_closed_form_matches_scipy():
    """the arithmetic design_band2 runs on the GPU, restated in Python, against scipy.signal.butter"""
    import scipy.signal
    rng = np.random.default_rng(0)
    worst = 0.0
    for btype in ('bp', 'bs'):
        for _ in range(1500):
            lo = rng.uniform(20, 20000)
            hi = rng.uniform(lo * 1.001, 23900)
            ref = scipy.signal.butter(2, [lo / 24000, hi / 24000], btype, output='sos')
            got = R.band2_sos(lo / 24000, hi / 24000, btype)
            assert got.shape == ref.shape == (2, 6)
            worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300))))
    assert worst < 1e-11, worst
    import pytest
    with pytest.raises(ValueError):
        R.band2_sos(0.5, 0.25, 'bp')
    with pytest.raises(ValueError):
        R.band2_sos(0.0, 0.25, 'bs')
