"""Development tool (GPU box): is the training step bound by the host?  Times one bf16 step of BASELINE configs[2]
(a) as bench.py runs it, (b) the host side alone (until the last launch is enqueued), (c) with zero_grad + forward + loss +
backward captured into one hipGraph (torch.cuda.graph) and the optimizer step eager behind the replay.

The capture is taken with the step on ONE stream (OTPOSE_TRAIN_STREAMS=0, OTPOSE_WGRAD_STREAM=0): with the branch streams on,
torch 2.10 / ROCm 7.2 segfaults inside capture_end for this ~9000-node graph (a recursion that does not end: raising the stack
limit turned the segfault into a machine that ran out of memory - do not try that again on a shared box).  Measured (round 4,
profiles/r04_train_graph_probe.txt): eager multi-stream step 145 ms with the host side alone 95-100 ms; the one-stream graph
replays in 168 ms, the same as the one-stream eager step (172 ms) - the step is bound by the GPU, not by its launches.
usage: python tools/train_graph_probe.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, '.')
from otpose_amd import OTPose, cfg2                        # noqa: E402
from otpose_amd import parallel as PAR                     # noqa: E402
from otpose_amd import synthetic as S                      # noqa: E402
from otpose_amd import train as TR                         # noqa: E402
from otpose_amd.bf16_ops import join_wgrad_streams         # noqa: E402
from otpose_amd.optim import FusedAdamW                    # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
cfg = cfg2()
model = OTPose(cfg)
S.fill_synthetic_(model)
model = model.to(dev).train()
model.train_dtype = "bf16"
x, margin = S.synthetic_clip(batch, cfg.MODEL.IMAGE_SIZE)
x, margin = x.to(dev), margin.to(dev)
J = cfg.MODEL.NUM_JOINTS
w, h = cfg.MODEL.HEATMAP_SIZE
gen = torch.Generator().manual_seed(11)
g = (torch.rand(batch, J, h, w, generator=gen) * 0.2).to(dev)
g[:, ::2, 3, 4] = 1.0
wt = (torch.rand(batch, J, 1, generator=gen) > 0.15).float().to(dev)
opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)


def eager_step():
    return PAR.train_step_dp(model, opt, x, margin, g, wt)


for _ in range(2):
    eager_step()
torch.cuda.synchronize()
tt, th = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = eager_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    tt.append(t2 - t0)
    th.append(t1 - t0)
print("eager step: %.1f ms wall, host side alone %.1f ms  (loss %.6f)" % (1e3 * sorted(tt)[2], 1e3 * sorted(th)[2], float(loss)), flush=True)


os.environ["OTPOSE_TRAIN_STREAMS"] = "0"            # see the module docstring
os.environ["OTPOSE_WGRAD_STREAM"] = "0"


def fwd_bwd():
    opt.zero_grad()
    outs = TR.forward_train(model, x, margin)
    flags = PAR.allreduce_joint_flags(PAR.joint_flags(g))
    loss = TR.criterion(outs, g, wt, flags)
    loss.backward()
    join_wgrad_streams()
    return loss.detach()


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        fwd_bwd()
        opt.step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("side-stream warm-up done", flush=True)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=s):
    loss_static = fwd_bwd()
torch.cuda.synchronize()
print("captured", flush=True)
tt = []
for _ in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    graph.replay()
    opt.step()
    torch.cuda.synchronize()
    tt.append(time.perf_counter() - t0)
print("graph replay + eager optimizer step: %.1f ms  (all: %s)  loss %.6f" % (1e3 * sorted(tt)[4], " ".join("%.1f" % (1e3 * t) for t in tt), float(loss_static)))
print("peak memory %.1f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30))
