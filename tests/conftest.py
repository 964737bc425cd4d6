import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        # the GPU box's host has 256 cores; the CPU oracle legs of the parity tests (chains of mid-size ops) run several times
        # FASTER on 16 intra-op threads than on torch's default of one per core (bench.py's cpu_baseline probes the same thing)
        if (os.cpu_count() or 1) > 32:
            torch.set_num_threads(16)
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
