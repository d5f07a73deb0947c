"""conv3x3_small launch times at the RSB staircase shapes (16 clips x 96x72; 6 / 13 / 20 channels, with the pre-added input)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402

for c in (6, 13, 20):
    x = torch.randn(16, c, 96, 72, device="cuda")
    x2 = torch.randn(16, c, 96, 72, device="cuda")
    w = torch.randn(c, c, 3, 3, device="cuda") * 0.1
    sc, sh = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    f = lambda: ops.conv3x3_small(x, w, sc, sh, ops.ACT_RELU, x2)   # noqa: E731
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        f()
    e.record()
    torch.cuda.synchronize()
    print("%2d -> %2d channels: %.1f us per call (incl. the weight re-layout launch)" % (c, c, a.elapsed_time(e) / 50 * 1e3))
