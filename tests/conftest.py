import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The CPU oracle's tensors are small (tiny-geometry fixtures: 128-wide, 2 layers): on the GPU box torch starts 128 OpenMP threads on 256
    # host CPUs and a 3-second evaluation takes 36 (measured, round 4: the 24-step trajectory test 145 s -> the oracle's share 107 s of it).
    import torch
    if torch.get_num_threads() > 16:
        torch.set_num_threads(16)


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible and -m gpu was not asked for.
    The two entry-point tests (run.py / run_adapter.py end to end: DataLoader worker processes are FORKED from the test process) go first:
    forked from a process that has already run the kernel suites (thousands of device allocations mapped) the same test took 43 - 161 s
    instead of 5 (measured, profiles/LOG.md round 4) -- the cost of fork() under a large ROCm address space, not of anything it tests."""
    first = [it for it in items if it.fspath.basename in ('test_text_run.py', 'test_cv_run.py') and 'gpu' in it.keywords]
    if first:
        rest = [it for it in items if it not in first]
        items[:] = first + rest
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)
