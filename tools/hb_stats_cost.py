"""csrc/hb.hip's window kernel with and without the statistics epilogue (development probe)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from otpose_amd import bf16_ops as B, hip
from tools.bf16_conv_bench import timeit
L = hip.lib()
for cin, cout, h, w in ((48, 48, 96, 72), (96, 96, 48, 36), (192, 192, 24, 18)):
    n = 80
    x = torch.randn(n, h, w, cin, device="cuda").to(B.BF16)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    d = B._desc(n, h, w, cin, cout, 3, 3, 1, 1, 1)
    wp = B._pack(wt, d, 0)
    out = torch.empty(n, h, w, cout, dtype=B.BF16, device="cuda")
    rows = L.otp_nhwc_conv_stats_rows(ctypes.byref(d))
    stats = torch.empty(rows, 2, cout, device="cuda")
    res = torch.randn(n, h, w, cout, device="cuda").to(B.BF16)
    f = lambda st, rs: hip.check(L.otp_nhwc_conv_bf16_res(hip.ptr(x), hip.ptr(wp), None, hip.ptr(rs), hip.ptr(out), hip.ptr(st), ctypes.byref(d), hip.stream_of(x)), "c")
    print("%d->%d @%dx%d: with statistics %.1f us, without %.1f us, with residual %.1f us" % (cin, cout, h, w, timeit(lambda: f(stats, None), 20), timeit(lambda: f(None, None), 20), timeit(lambda: f(None, res), 20)))
