import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "range_overflow_expected: the test drives the half-piece arithmetic out of range on purpose")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        return {k: torch.from_numpy(z[k]) for k in z.files if z[k].dtype.kind in "fiub"}
    return load


def seeded(shape, seed, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=dtype) * scale


@pytest.fixture(autouse=True)
def _release_gpu_state(request):
    """After every GPU test: collect the model <-> engine reference cycles (an inference engine owns tens of GB of static
    buffers and a hipGraph; cyclic garbage is otherwise freed whenever Python gets round to it) and hand cached blocks back,
    so that the ~400 tests of the suite do not pile engines, graphs and their memory pools up in one process."""
    yield
    if "gpu" in request.keywords and torch.cuda.is_available():
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        # the range guard of the half-piece kernels (csrc/range.hip) is process-wide and sticky: no test may leave it set, and a
        # test that did not ask for an overflow must not have raised it (a false positive of the guard would show up here)
        from otpose_amd import hip
        code = hip.lib().otp_range_flag_read(1)
        if code and "range_overflow_expected" not in request.keywords:
            pytest.fail(f"the range guard fired (kernel family {code}) in a test that stays inside the half range")
