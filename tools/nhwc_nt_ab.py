"""Development tool (GPU box): A / B of two builds of the library on nhwc_conv_kernel - forward conv of the HRNet branch shapes
and a few others, 80 frames, microseconds per launch (HIP events, 30 launches) and agreement of the two results.  The tool starts
itself once per build (OTPOSE_HIP_LIB) BEFORE anything touches the GPU.  Round 4 used it for two experiments:
profiles/r04_nhwc_two_tiles_ab.txt (two pixel tiles per workgroup, a build switch that is no longer in the tree) and
profiles/r04_nhwc_minwg_ab.txt (-DOTP_NHWC_MINWG=3: register budget for three workgroups per CU).
usage: python tools/nhwc_nt_ab.py <libA.so> <libB.so>"""
import os
import subprocess
import sys

SHAPES = [(48, 48, 96, 72, 3), (96, 96, 48, 36, 3), (192, 192, 24, 18, 3), (384, 384, 12, 9, 3), (64, 64, 96, 72, 3),
          (32, 306, 96, 72, 3), (48, 96, 96, 72, 1), (256, 64, 96, 72, 1)]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, '.')
    from otpose_amd import bf16_ops as B
    torch.manual_seed(0)
    outs = []
    for cin, cout, h, w, k in SHAPES:
        n = 80 if cin != 32 else 16
        x = torch.randn(n, h, w, B.cs(cin), device="cuda").to(B.BF16)
        wt = torch.randn(cout, cin, k, k, device="cuda") * 0.05
        for _ in range(3):
            out, stats, rows = B.conv_forward(x, wt, None, 1, k // 2, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            out, stats, rows = B.conv_forward(x, wt, None, 1, k // 2, 1)
        e1.record()
        torch.cuda.synchronize()
        print("%d %d %d %d %d %.2f %.6e %.6e" % (cin, cout, h, w, k, 1e3 * e0.elapsed_time(e1) / 30,
                                                 float(out.float().abs().sum()), float(stats.float().abs().sum())), flush=True)
    sys.exit(0)

res = {}
libs = {"1": os.path.abspath(sys.argv[1]), "2": os.path.abspath(sys.argv[2])}
for nt in ("1", "2"):
    env = dict(os.environ, OTPOSE_HIP_LIB=libs[nt])
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=600)
    if r.returncode:
        print(r.stderr[-2000:])
        sys.exit(1)
    res[nt] = [ln.split() for ln in r.stdout.splitlines() if ln and ln[0].isdigit()]
print("A = %s\nB = %s" % (sys.argv[1], sys.argv[2]))
print("%-28s %10s %10s   %s" % ("shape (80 frames, forward)", "A us", "B us", "same result"))
for a, b in zip(res["1"], res["2"]):
    print("%-28s %10s %10s   %s" % ("%s->%s k%s @%sx%s" % (a[0], a[1], a[4], a[2], a[3]), a[5], b[5], a[6:] == b[6:]))
