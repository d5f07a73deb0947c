"""Development aid (GPU box): the ln2 + MLP launch (OTP_MLP_NT1=1: two workgroups per CU) on one stream next to ONE other
temporal-encoder kernel on another stream - which pair stops being bit-stable?  usage: OTP_MLP_NT1=1 python
tools/encoder_pair_stress.py [rounds=40]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otpose_amd import ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B, C, HID, T = 16, 136, 544, 6912
g = torch.Generator().manual_seed(3)
rnd = lambda *s: torch.randn(*s, generator=g).cuda()   # noqa: E731
xm = rnd(B, C, T)
w1, w2 = rnd(HID, C, 1) / C ** 0.5, rnd(C, HID, 1) / HID ** 0.5
b1, one, zero = rnd(HID) * 0.1, torch.ones(C).cuda(), torch.zeros(C).cuda()
px = ops.pack_mlp_x3_weights(w1, b1, w2)
mlp_out = torch.empty_like(xm)
mlp = lambda st: ops.ln_mlp_x3(xm, one, zero, 1e-5, px, one, zero, out=mlp_out, stream=st.cuda_stream)   # noqa: E731

# the other kernels, each writing into its own outputs
xa, res = rnd(B, C, T), rnd(B, C, T)
wp = rnd(C, C, 1) / C ** 0.5
pk = ops.pack_dense_cc(wp, rnd(C), rnd(C), x3=True)
proj_out = torch.empty_like(xa)
ws = [rnd(C, C, 1) / C ** 0.5 for _ in range(3)]
packs = [ops.pack_dense_cc(w, None, rnd(C), x3=True) for w in ws]
dws = [rnd(C, 1, 3) * 0.6 for _ in range(3)]
table = ops.pack_qkv_table(dws[0], dws[1], dws[2], one, zero, one, zero, one, zero)
qkv_outs = [torch.empty_like(xa) for _ in range(3)]
q, k, v = rnd(B, C, T), rnd(B, C, T), rnd(B, C, T)
others = {
    "densex proj": (lambda st: ops.dense_cc([xa], [pk], [res], [proj_out], stream=st.cuda_stream, x3=True), lambda: [proj_out]),
    "qkv front": (lambda st: ops.qkv_front(xa, table, packs, 1e-5, outs=qkv_outs, stream=st.cuda_stream, x3=True), lambda: qkv_outs),
}
try:
    att_holder = {}

    def run_att(st):
        with torch.cuda.stream(st):
            att_holder["o"] = ops.chan_attn(q, k, v, 2, 68 ** -0.5)
    others["chan attn (scores, softmax, P.v)"] = (run_att, lambda: [att_holder["o"]])
except Exception:
    pass
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()                        # operands and packed weights were made on the default stream
mlp(s0)
torch.cuda.synchronize()
mlp_ref = mlp_out.clone()
for name, (fn, outs) in others.items():
    fn(s1)
    torch.cuda.synchronize()
    refs = [o.clone() for o in outs()]
    bad_mlp = bad_other = 0
    for it in range(rounds):
        mlp_out.zero_()
        for o in outs():
            o.zero_()
        torch.cuda.synchronize()
        for _ in range(3):
            fn(s1)
            mlp(s0)
        torch.cuda.synchronize()
        if not torch.equal(mlp_out, mlp_ref):
            bad_mlp += 1
        if any(not torch.equal(a, b) for a, b in zip(outs(), refs)):
            bad_other += 1
            for oi, (a, b) in enumerate(zip(outs(), refs)):
                if not torch.equal(a, b):
                    d = (a != b).nonzero()
                    ch, tk = d[:, 1].unique(), d[:, 2].unique()
                    print("     round %d output %d: %d elements, max |d| %.3e (|ref| max %.2f); clips %s; channels %d..%d (%d distinct: %s); "
                          "tokens %d..%d (%d distinct)" % (it, oi, len(d), float((a - b).abs().max()), float(b.abs().max()),
                                                          d[:, 0].unique().tolist(), int(ch.min()), int(ch.max()), len(ch), ch[:20].tolist(),
                                                          int(tk.min()), int(tk.max()), len(tk)))
    print("%-36s next to the MLP launch: MLP output differs in %d of %d rounds, its own in %d" % (name, bad_mlp, rounds, bad_other))
