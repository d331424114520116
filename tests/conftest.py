import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_SPAWNING = ("test_cli_gpu.py", "test_distributed_cpu.py")    # their GPU work happens in child processes only
_torch_up = []


@pytest.fixture(autouse=True)
def _torch_runtime_first(request):
    """In-process GPU tests: bring PyTorch's HIP runtime up once, before the test touches the GPU through
    libreal_hip.so.  Both use the same libamdhip64 (same SONAME: whichever is loaded first serves both); a torch.cuda
    initialisation that came AFTER the library had been busy on the device was seen to report "No HIP GPUs are
    available" on one box.  (Not for the test files that only start child processes: a parent that has initialised the
    GPU should not be the one that spawns them.)"""
    if "gpu" in request.node.keywords and os.path.basename(str(request.node.fspath)) not in _SPAWNING and not _torch_up:
        _torch_up.append(True)
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
    yield


@pytest.fixture(scope="session")
def ora():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib
