import gzip
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def nms_golden():
    z = np.load(os.path.join(GOLDEN, 'nms_golden.npz'))
    with open(os.path.join(GOLDEN, 'nms_golden_index.json')) as f:
        index = json.load(f)
    return z, index


@pytest.fixture(scope="session")
def exotic_golden():
    """{case name: keep list, or the exception class the reference raised} (make_golden.py --exotic-only)."""
    z = np.load(os.path.join(GOLDEN, 'exotic_golden.npz'))
    out = {}
    for c in __import__('synth').EXOTIC_CASES:
        raised = str(z['exotic_%s_raised' % c['name']])
        out[c['name']] = ZeroDivisionError if raised == 'ZeroDivisionError' else z['exotic_' + c['name']].tolist()
    return out


@pytest.fixture(scope="session")
def proto_golden():
    with gzip.open(os.path.join(GOLDEN, 'proto_golden.json.gz'), 'rt') as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
