import os, sys, ctypes
sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from otpose_amd import bf16_ops as B
BF = torch.bfloat16
def nhwc(t):
    n, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).contiguous().to(BF).cuda()
def nchw(t):
    return t.float().permute(0, 3, 1, 2).cpu()
for (cin, cout, s, n, h, w) in [(16,16,1,10,8,12),(32,32,1,10,4,6),(16,32,2,10,8,12),(32,16,1,10,8,12),(16,16,1,10,16,24),(64,64,1,10,4,6),(16,8,1,10,8,12),(16,24,1,3,8,12)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g).to(BF).float()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (cin*9)**0.5).to(BF).float()
    ref = F.conv2d(x.double(), wt.double(), None, s, 1, 1).float()
    gy = torch.randn(ref.shape, generator=g).to(BF).float()
    gref = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), s, 1, 1).float()
    res = torch.randn(x.shape, generator=g).to(BF).float()
    for hb in ("2", "0"):
        os.environ["OTPOSE_NHWC_HB"] = hb
        out, stats, rows = B.conv_forward(nhwc(x), wt.cuda(), None, s, 1, 1)
        gx = B.conv_dgrad(nhwc(gy), wt.cuda(), (h, w), s, 1, 1)
        gxr = B.conv_dgrad(nhwc(gy), wt.cuda(), (h, w), s, 1, 1, res=nhwc(res))
        torch.cuda.synchronize()
        o = nchw(out); e1 = float((o - ref).abs().max() / ref.abs().max())
        e2 = float((nchw(gx) - gref).abs().max() / gref.abs().max())
        st = stats.sum(0).cpu(); of = out.float().cpu().reshape(-1, cout)
        e3 = float((st[0] - of.sum(0)).abs().max()); e4 = float((st[1] - (of*of).sum(0)).abs().max())
        e5 = bool(torch.equal(gxr, gx + nhwc(res)))
        print((cin,cout,s,n,h,w), "hb" + hb, "fwd %.2e dgrad %.2e stats %.2e %.2e res_eq %s rows %d" % (e1, e2, e3, e4, e5, rows))
