"""Development aid (GPU box): ONE victim launch (the C -> C projection of csrc/densex.hip) on a stream, one neighbour kernel
family on another - for which neighbours does the victim's output stop being bit-identical to its quiet run?
usage: python tools/victim_neighbour_matrix.py [rounds=40]     (OTP_MLP_NT1=1 makes the MLP neighbour two workgroups per CU)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, C, HID, T = 16, 136, 544, 6912
g = torch.Generator().manual_seed(3)
rnd = lambda *s: torch.randn(*s, generator=g).cuda()   # noqa: E731
xa, res = rnd(B, C, T), rnd(B, C, T)
pk = ops.pack_dense_cc(rnd(C, C, 1) / C ** 0.5, rnd(C), rnd(C), x3=True)
vout = torch.empty_like(xa)
victim = lambda st: ops.dense_cc([xa], [pk], [res], [vout], stream=st.cuda_stream, x3=True)   # noqa: E731

xm = rnd(B, C, T)
px = ops.pack_mlp_x3_weights(rnd(HID, C, 1) / C ** 0.5, rnd(HID) * 0.1, rnd(C, HID, 1) / HID ** 0.5)
one, zero = torch.ones(C).cuda(), torch.zeros(C).cuda()
mo = torch.empty_like(xm)
xc = rnd(40, 96, 48, 36)
wc = ops.pack_x3_weight(rnd(96, 96, 3, 3) * 0.03, None, 1)
yc = torch.empty_like(xc)
iv, ov = ops.View(xc), ops.View(yc)
dd = ops.conv_desc(iv, ov, 96, 3, 3, 1, 1, 1, ops.ACT_RELU)
xs8 = ops.s8_pack(rnd(40, 48, 96, 72))
ws8 = ops.pack_s8_weight(rnd(48, 48, 3, 3) * 0.05)
ys8 = ops.s8_empty(40, 48, 96, 72, "cuda")
ds8 = ops.s8_conv_desc(40, 48, 48, 96, 72, ops.ACT_RELU)
big_a, big_b = rnd(64, 1024, 1024), rnd(64, 1024, 1024)
x2, r2, o2 = rnd(B, C, T), rnd(B, C, T), torch.empty(B, C, T, device="cuda")


def n_torch(st):
    with torch.cuda.stream(st):
        torch.add(big_a, big_b, out=big_a)
        torch.mul(big_a, 0.5, out=big_a)


neighbours = {
    "nothing": lambda st: None,
    "ln2 + MLP (csrc/mlpx.hip)": lambda st: ops.ln_mlp_x3(xm, one, zero, 1e-5, px, one, zero, out=mo, stream=st.cuda_stream),
    "another projection (csrc/densex.hip)": lambda st: ops.dense_cc([x2], [pk], [r2], [o2], stream=st.cuda_stream, x3=True),
    "conv 3x3 96->96 (csrc/convx.hip)": lambda st: ops.conv2d_x3_launch(iv, wc, None, ov, dd, None, stream=st.cuda_stream),
    "S8 conv 48->48 (csrc/convs.hip)": lambda st: ops.conv3x3_s8_launch(xs8, ws8, None, ds8, None, None, ops.S8_F32_C4, ys8, stream=st.cuda_stream),
    "torch add / mul (64 M elements)": n_torch,
}
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()                        # operands and packed weights were made on the default stream
victim(s0)
torch.cuda.synchronize()
ref = vout.clone()
for name, fn in neighbours.items():
    bad = 0
    for it in range(rounds):
        vout.zero_()
        torch.cuda.synchronize()
        for _ in range(3):
            fn(s1)
            victim(s0)
        torch.cuda.synchronize()
        bad += int(not torch.equal(vout, ref))
    print("victim next to %-40s: differs in %d of %d rounds" % (name, bad, rounds))
