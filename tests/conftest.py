import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


def pytest_sessionstart(session):
    """The C-ABI library is a build product (git-ignored): build it when a fresh checkout runs the
    tests before __graft_entry__.build() (hipcc cross-compiles without a GPU; ~20 s)."""
    lib = os.path.join(ROOT, 'cwn_amd', 'libcwn_hip.so')
    if not os.path.exists(lib) and os.path.exists('/opt/rocm/bin/hipcc'):
        import subprocess
        subprocess.run(['make', '-C', os.path.join(ROOT, 'cwn_amd', 'csrc'), '-j4'], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests require a GPU; everything else must pass on CPU."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)
