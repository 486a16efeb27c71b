"""signals_amd/specialise.py without a GPU: the specialised build of voice_program.hip (hipcc cross-compiles gfx950 here) --
the macros, the image, the disk cache.  Attaching and running it is tests/test_gpu_specialise.py."""
import time

import pytest

from signals_amd import _native, specialise

FLAGS = specialise.flags
CODE = [('Osc', 2, 0, 0, 0), ('Filter', 0, 0, 0, 0), ('Save', 0, 0, 0, 0), ('Osc', 3, 1, 0, 0), ('Filter', 0, 1, 0, 0), ('Mul', 0, 0, 0, 0)]


def test_flags_carry_the_program_and_an_exact_register_file():
    f = specialise.flags(CODE, 2, 0, 2, 1, 2, 2)
    words = _native.voice_program_words(CODE)
    assert words == [0x40, 0x1, 0x5, 0x160, 0x101, 0x3]                       # op | kind << 5 | a << 8 (include/signals_amd.h)
    assert '-DSIG_VP_STATIC_CODE={0x40,0x1,0x5,0x160,0x101,0x3}' in f
    assert {'-DSIG_VP_S_NF=2', '-DSIG_VP_S_NO=2', '-DSIG_VP_S_NP=1', '-DSIG_VP_S_NT=1', '-DSIG_VP_S_EXT=0',
            '-DSIG_VP_STATIC_VPT=2', '-DSIG_VP_STATIC_C=2', '-DSIG_VP_STATIC_WAVES=2'} <= set(f)
    for op in ('Amp', 'Adsr', 'Noise'):                                       # the extended handlers are compiled in when the program has one
        assert '-DSIG_VP_S_EXT=1' in specialise.flags([('Osc', 0, 0, 0, 0), (op, 0, 0, 0, 0)], 1, 1, 0, 0, 1, 0), op


@pytest.mark.skipif(specialise.hipcc() is None, reason='no hipcc in this environment')
def test_the_image_builds_without_a_gpu_and_is_cached(tmp_path, monkeypatch):
    monkeypatch.setattr(specialise, 'CACHE', tmp_path)
    t0 = time.perf_counter()
    image = specialise.build(CODE, 2, 0, 2, 1, 2, 2)
    first = time.perf_counter() - t0
    assert b'sig_vp_specialised' in image and b'sig_vp_specialised_info' in image
    assert b'voice_program_kernel' not in image                               # the interpreter's instantiations are not part of it
    assert len(list(tmp_path.glob('vp_*.hsaco'))) == 1
    t0 = time.perf_counter()
    again = specialise.build(CODE, 2, 0, 2, 1, 2, 2)
    assert again == image and time.perf_counter() - t0 < min(first, 0.5)     # from the cache
    other = specialise.build(CODE, 2, 0, 2, 1, 1, 0)                          # another geometry / sink: another image
    assert other != image and len(list(tmp_path.glob('vp_*.hsaco'))) == 2


@pytest.mark.skipif(specialise.hipcc() is None, reason='no hipcc in this environment')
def test_a_program_with_an_envelope_cannot_be_built_without_the_extended_handlers(tmp_path, monkeypatch):
    """(a first version spelt the op name wrong, built such programs without their ADSR handler and rendered garbage)"""
    monkeypatch.setattr(specialise, 'CACHE', tmp_path)
    code = [('Adsr', 0, 0, 0, 0), ('Save', 0, 0, 0, 0), ('Osc', 2, 0, 0, 0), ('Mul', 0, 0, 0, 0)]
    assert b'sig_vp_specialised' in specialise.build(code, 1, 0, 0, 1, 1, 0)
    monkeypatch.setattr(specialise, 'flags', lambda *a: [f.replace('SIG_VP_S_EXT=1', 'SIG_VP_S_EXT=0') for f in FLAGS(*a)])
    with pytest.raises(specialise.SpecialiseError, match='SIG_VP_S_EXT'):
        specialise.build(code, 1, 0, 0, 1, 2, 0)


def test_a_failing_compiler_is_reported(tmp_path, monkeypatch):
    monkeypatch.setattr(specialise, 'CACHE', tmp_path)
    monkeypatch.setenv('HIPCC', '/bin/false')
    with pytest.raises(specialise.SpecialiseError):
        specialise.build(CODE, 2, 0, 2, 1, 2, 2)
