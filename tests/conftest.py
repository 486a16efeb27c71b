import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / 'tests' / 'golden'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


class Golden:
    """Read-only view of one tests/golden/<name>.npz written by gen_golden.py
    (keys use '/' in the generator and '__' in the file)."""

    def __init__(self, name):
        self._z = np.load(GOLDEN / f'{name}.npz', allow_pickle=False)

    def __getitem__(self, key):
        return self._z[key.replace('/', '__')]

    def keys(self):
        return [k.replace('__', '/') for k in self._z.files]


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get
