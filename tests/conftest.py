import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    # every kernel workspace is NaN-filled before use for the whole GPU suite: a partial result that is never written,
    # or read before it is written, fails the run it happens in instead of hiding behind stale finite bytes
    from vfd_gan_amd import functional as F
    F.set_workspace_poison(True)
    return torch.device("cuda", 0)
