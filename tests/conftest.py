import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the real reference binary oracle/_ref/kpeg_ref")


@pytest.fixture(scope="session", autouse=True)
def _built_checkers():
    """Oracle + synthetic generator are test infrastructure; build them on first use."""
    import kpeg_testlib
    kpeg_testlib.ensure_built()


@pytest.fixture(scope="session", autouse=True)
def _torch_side_stream():
    """On a GPU box every test's torch ops run on one created stream: PyTorch's default stream has handle 0, which the C ABI
    reads as "the context's own stream", so work on it would not be ordered with a context told to use
    `torch.cuda.current_stream()` (libkpeg_amd.Context.set_stream refuses handle 0 for that reason)."""
    try:
        import torch
    except ImportError:
        yield
        return
    if torch.cuda.is_available():
        torch.cuda.set_stream(torch.cuda.Stream())
    yield
