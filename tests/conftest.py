import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_aug():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "aug_golden.npz"))


@pytest.fixture(scope="session")
def golden_aug_extra():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "aug_extra_golden.npz"))


@pytest.fixture(scope="session")
def golden_eval():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "eval_golden.npz"))


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but torch.cuda.is_available() is False")
    from augcl_amd import _lib
    lib = _lib.load()
    assert lib.augcl_device_count() >= 1, lib.augcl_last_error()
    return torch.device("cuda:0")
