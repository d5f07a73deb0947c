"""per-layer gx / gw errors of the bf16 backbone (tests/test_gpu_train_bf16_yardstick.py) - development probe"""
import os, sys
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import test_gpu_train_bf16_yardstick as Y
from tests.conftest import seeded
from otpose_amd import train as TR
from oracle import otpose_oracle as O
cfg = Y.tiny_cfg(16, (128, 192))
model, x = Y._backbone(cfg)
pre = "rough_pose_estimation_net"
model = model.cuda().train()
graph = TR.TrainGraphBF16(model)
layers = []
graph.taps = {"layers": layers}
out = graph.hrnet(pre, graph.hrnet_input(x.cuda()))
tgt = seeded(tuple(out.shape), 21).abs() * 0.3
(0.5 * ((out - tgt.cuda()) ** 2).mean()).backward()
torch.cuda.synchronize()
P = {n: p for n, p in model.named_parameters()}
for rec in layers:
    conv, bn = rec["conv"], rec["bn"]
    w = P[conv + ".weight"]
    cout, cin = w.shape[:2]
    sd = {"c.weight": w.detach().cpu().double().requires_grad_(), "b.weight": P[bn + ".weight"].detach().cpu().double().requires_grad_(),
          "b.bias": P[bn + ".bias"].detach().cpu().double().requires_grad_()}
    xin = Y._nchw(rec["x"], cin).requires_grad_()
    res = Y._nchw(rec["res"], cout).requires_grad_() if rec["res"] is not None else None
    with O.bf16_points():
        y = O._conv_bn(sd, "c", "b", xin, rec["stride"], rec["pad"], rec["relu"], True, res=res)
        y.backward(Y._nchw(rec["gy"], cout))
    e = {"y": Y._rel(Y._nchw(rec["y"], cout), y.detach()), "gw": Y._rel(w.grad.cpu().double(), sd["c.weight"].grad)}
    if "gx" in rec:
        e["gx"] = Y._rel(Y._nchw(rec["gx"], cin), xin.grad)
    if max(e.values()) > 1e-3:
        print(conv, tuple(w.shape), "stride", rec["stride"], "x", tuple(rec["x"].shape), {k: "%.2e" % v for k, v in e.items()}, "res" if res is not None else "")
print("layers", len(layers))
# ---- replay the failing layer stand-alone under both kernels ------------------------------------------------------------------
from otpose_amd import bf16_ops as B
for rec in layers:
    if not rec["conv"].endswith("stage3.0.branches.1.2.conv1"):
        continue
    conv, bn = rec["conv"], rec["bn"]
    outs = {}
    for hb in ("2", "0"):
        os.environ["OTPOSE_NHWC_HB"] = hb
        xi = rec["x"].clone().requires_grad_()
        w = P[conv + ".weight"].detach().clone().requires_grad_()
        ga = P[bn + ".weight"].detach().clone().requires_grad_()
        be = P[bn + ".bias"].detach().clone().requires_grad_()
        rm, rv = torch.zeros_like(ga), torch.ones_like(ga)
        y = B.conv_bn(xi, w, ga, be, None, rm, rv, rec["stride"], rec["pad"], rec["relu"], 0.1, 1e-5)
        y.backward(rec["gy"])
        torch.cuda.synchronize()
        outs[hb] = (y.detach().float(), xi.grad.float(), w.grad.float())
        print("replay hb%s: y vs recorded %.3e, gx vs recorded %.3e" % (hb, float((outs[hb][0] - rec["y"].float()).norm() / rec["y"].float().norm()),
              float((outs[hb][1] - rec["gx"].float()).norm() / rec["gx"].float().norm())))
    print("replay hb2 vs hb0: y %.3e gx %.3e gw %.3e" % tuple(float((a - b).norm() / b.norm()) for a, b in zip(outs["2"], outs["0"])))
    # ---- ReLU decisions of this layer: the oracle's own y against the recorded y ------------------------------------------------
    w = P[conv + ".weight"]
    cout, cin = w.shape[:2]
    sd = {"c.weight": w.detach().cpu().double(), "b.weight": P[bn + ".weight"].detach().cpu().double(), "b.bias": P[bn + ".bias"].detach().cpu().double()}
    with O.bf16_points(), torch.no_grad():
        yo = O._conv_bn(sd, "c", "b", Y._nchw(rec["x"], cin), rec["stride"], rec["pad"], rec["relu"], True, res=None)
    yg = Y._nchw(rec["y"], cout)
    gy = Y._nchw(rec["gy"], cout)
    flip = (yo > 0) != (yg > 0)
    print("relu decisions that differ: %d of %d; |gy| there / |gy| %.3e; per channel:" % (int(flip.sum()), flip.numel(), float((gy * flip).norm() / gy.norm())),
          flip.sum((0, 2, 3)).tolist())
    print("gamma", [round(float(v), 4) for v in sd["b.weight"]], "\nbeta", [round(float(v), 4) for v in sd["b.bias"]])
