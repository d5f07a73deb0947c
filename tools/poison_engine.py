"""Development tool (GPU box): find engine memory that some kernel reads before anything wrote it.
  (a) OTPOSE_POISON=lo:hi NaN-fills the activation buffers InferenceEngine.new() hands out (bisected below);
  (b) the caching allocator's free blocks are filled with NaN (or a large finite value) before the engine is built, so that
      every torch.empty() of the build - packed weights, tables, workspaces - starts from garbage."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import OTPose, cfg2
from otpose_amd import synthetic as S
cfg = cfg2()
NAMES = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
BATCH = int(os.environ.get("POISON_BATCH", "1"))
x, mg = S.synthetic_clip(BATCH, cfg.MODEL.IMAGE_SIZE)

def poison_allocator(val):
    torch.cuda.synchronize()
    ts = [torch.full((64 << 20,), val, device="cuda") for _ in range(40)]          # 10 GB in 256 MB blocks
    small = [torch.full((n,), val, device="cuda") for n in (64, 256, 4096, 65536, 1 << 18, 1 << 20, 4 << 20, 16 << 20) for _ in range(16)]
    del ts, small
    torch.cuda.synchronize()

def forward(env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    m = OTPose(cfg); S.fill_synthetic_(m); m = m.cuda().eval()
    with torch.no_grad():
        o = [t.clone() for t in m(x.cuda(), margin=mg.cuda())]
    for k in (env or {}):
        os.environ.pop(k)
    nb = len(m._engine._bufs)
    del m
    torch.cuda.empty_cache()
    return o, nb

ref, nb = forward()
for val, tag in ((float("nan"), "NaN"), (1e30, "1e30"), (0.0, "zeros")):
    poison_allocator(val)
    got, _ = forward()
    print("allocator filled with %-5s: " % tag + ", ".join("%s %s" % (n, "same" if torch.equal(a, b) else ("DIFF %.2e%s" % (float((a - b).abs().nan_to_num(9e9).max()), " nan" if not bool(torch.isfinite(b).all()) else "")))
                                                          for n, a, b in zip(NAMES, ref, got)))
got, _ = forward({"OTPOSE_POISON": "0:%d" % (1 << 30)})
print("engine buffers NaN-filled:   " + ", ".join("%s %s" % (n, "same" if torch.equal(a, b) else "DIFF") for n, a, b in zip(NAMES, ref, got)))
