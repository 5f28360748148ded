import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")
    config.addinivalue_line("markers", "multiproc: spawns ranks / subprocesses (sockets, rendezvous); collected LAST so that a "
                                       "flake there under -x cannot hide kernel / parity tests")


def pytest_collection_modifyitems(config, items):
    import torch

    # stable partition: everything that is not multi-process first, in its collection order
    items[:] = [i for i in items if "multiproc" not in i.keywords] + [i for i in items if "multiproc" in i.keywords]

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
