"""The C ABI used with no Python and no torch in the process: examples/c2_direct (hipMalloc + two calls into
libsignals_amd.so) must produce the same stereo bus as the Python engine for the same parameters."""
import json
import pathlib
import subprocess

import numpy as np
import pytest
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
EXE = ROOT / 'examples' / 'c2_direct'


def lcg_params(V):
    """the generator of examples/c2_direct.cpp, in Python integers"""
    s, mask = 12345, (1 << 64) - 1

    def nxt():
        nonlocal s
        s = (s * 6364136223846793005 + 1442695040888963407) & mask
        return (s >> 11) / 9007199254740992.0
    hz, ph, cut, gain = (np.empty((1, V)) for _ in range(4))
    pan = np.empty((2, V))
    for v in range(V):
        hz[0, v] = 55.0 + 1705.0 * nxt()
        ph[0, v] = nxt()
        cut[0, v] = 200.0 + 7800.0 * nxt()
        gain[0, v] = nxt() / V
        th = 1.5707963267948966 * nxt()
        pan[0, v], pan[1, v] = np.cos(th), np.sin(th)
    return dict(hertz=hz, phase=ph, cutoff=cut, gain=gain, pan=pan)


def test_example_links_only_the_kernel_library():
    if not EXE.exists():
        pytest.skip('examples/c2_direct not built')
    libs = subprocess.run(['ldd', str(EXE)], capture_output=True, text=True).stdout
    assert 'libsignals_amd.so' in libs and 'torch' not in libs and 'python' not in libs


@pytest.mark.gpu
@pytest.mark.parametrize('blocks,position', ((64, 0), (32, 172_800_000)))
def test_c_program_matches_python_engine(blocks, position):
    assert torch.cuda.is_available()
    if not EXE.exists():                                     # normally built by __graft_entry__.build()
        subprocess.run(['bash', str(ROOT / 'signals_amd' / 'csrc' / 'build.sh')], check=True)
    out = subprocess.run([str(EXE), str(blocks), str(position)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    c = json.loads(out.stdout)
    assert c['status'] == 0 and c['blocks'] == blocks

    import bench
    from signals_amd import runtime
    from signals_amd.engine import BatchRenderer
    runtime.set_device('cuda:0')
    r = BatchRenderer(bench.build_graph(lcg_params(1024), 0, 1024), 2, 48000)
    r.scan_max_chains = 0
    bus = r.render(position, 256, blocks).cpu().numpy().astype(np.float64)
    # the engine hands the kernel its voices ordered by cutoff (engine.py: ordered_by_cutoff), the C program in the
    # order they were drawn: the same bus up to the rounding of the voice sum, i.e. one float32 ulp of a ~0.03 bus
    assert np.abs(bus[0] - np.array(c['first'], dtype=np.float64)).max() <= 4e-9
    assert np.abs(bus[-1] - np.array(c['last'], dtype=np.float64)).max() <= 4e-9
    assert abs(bus.sum() - c['sum']) <= 1e-9 * max(1.0, abs(c['sum']))
    assert abs((bus * bus).sum() - c['sumsq']) <= 1e-9 * c['sumsq']
    assert abs(np.abs(bus).max() - c['peak']) < 1e-9 * c['peak']          # printed with 10 significant digits
