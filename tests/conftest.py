import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-process GPU tests take their ranks from a fork server that is started HERE, before anything in this process
    # touches the GPU: the ranks are then forked from a clean process (a process that has initialised HIP must neither
    # fork GPU work nor exec another program on the GPU boxes).
    import multiprocessing
    from multiprocessing import forkserver

    multiprocessing.get_context("forkserver")
    forkserver.ensure_running()


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
