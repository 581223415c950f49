import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')
    # The multi-process GPU test forks its ranks from a fork server. Start that server NOW, before
    # anything in this process initialises HIP: a process that has touched the GPU must never be
    # forked into ranks or exec another program (the pool's machines do not survive it).
    import multiprocessing.forkserver as forkserver
    forkserver.ensure_running()


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) where no device exists, e.g. the build container."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Keep the strict-parity margins of a GPU session (copied to profiles/rNN_parity_margins.txt)."""
    try:
        from tests._golden import margins_report
        text = margins_report()
        if text:
            out = ROOT / 'gpurun_out'
            out.mkdir(exist_ok=True)
            (out / 'parity_margins.txt').write_text(text)
    except Exception as exc:   # a report must never turn a green session red
        print(f'parity margins not written: {exc}')
