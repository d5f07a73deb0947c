"""Development tool (GPU box): re-run f32 training ops with the caching allocator's free blocks poisoned with NaN; an
operator that reads memory it did not write (its own uninitialised scratch, padding of a packed buffer) changes result."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import train_ops as T
dev = torch.device("cuda", 0)

def poison():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    ts = [torch.full((64 << 20,), float("nan"), device=dev) for _ in range(24)]   # 6 GB of NaN in 256 MB blocks
    small = [torch.full((n,), float("nan"), device=dev) for n in (256, 4096, 65536, 1 << 20, 4 << 20) for _ in range(8)]
    del ts, small
    torch.cuda.synchronize()

def check(name, fn):
    ref = [t.clone() for t in fn()]
    poison()
    got = fn()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(ref, got)):
        same = torch.equal(a, b) or bool(((a - b).abs() <= 1e-5 * a.abs().max()).all())
        print("%-40s out %d: %s  (nan in result: %s)" % (name, i, "same" if same else "DIFFERENT", bool(torch.isnan(b).any())))

g = torch.Generator().manual_seed(1)
for (cin, cout, k, s, pad, h, w, n) in [(256, 96, 3, 2, 1, 96, 72, 80), (64, 64, 3, 2, 1, 192, 144, 16), (48, 96, 3, 2, 1, 96, 72, 80), (48, 48, 3, 1, 1, 96, 72, 80), (96, 48, 1, 1, 0, 48, 36, 80)]:
    x = torch.randn(n, cin, h, w, generator=g).to(dev)
    wt = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev)
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    gy = torch.randn(n, cout, ho, wo, generator=g).to(dev)
    tag = "%d->%d k%d s%d %dx%d" % (cin, cout, k, s, h, w)
    check(tag + " fwd", lambda: [T.conv2d_forward(x, wt, None, s, pad, 1)])
    check(tag + " dgrad", lambda: [T.conv2d_grad_input(gy, wt, x.shape, s, pad, 1)])
    check(tag + " wgrad", lambda: [T.conv2d_grad_weight(x, gy, wt.shape, s, pad, 1)])
    gam, bet = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    yy = gy.clone().requires_grad_()
    def bn():
        yy.grad = None
        o = T.batch_norm_relu(yy, gam, bet, gy, rm, rv, 0.1, 1e-5, True)
        o.backward(gy)
        return [o.detach(), yy.grad]
    check(tag + " bn fwd+bwd", bn)
