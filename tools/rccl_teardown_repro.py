#!/usr/bin/env python
"""Reproducer hunt for the one-rank RCCL teardown abort (tests/test_gpu_zz_rccl_world1.py works around it
with child processes): in a long-lived process with captured hipGraphs behind it, `dist.barrier()` + `destroy_process_group()`
of a ONE-rank "nccl" group aborted about once in five runs of the full suite.  This script runs the suspected ingredients in
fresh child processes and counts how each variant ends:

    python tools/rccl_teardown_repro.py [children per variant]

variants: graphs captured before the group exists (0 / 40), a barrier before the teardown or not, a device synchronise before it or
not.  A child = capture graphs -> init group -> all_reduce -> (barrier) -> (synchronise) -> destroy -> replay nothing -> exit."""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, socket, torch, torch.distributed as dist
graphs, barrier, sync = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
keep = []
x = torch.randn(1 << 20, device=dev)
for i in range(graphs):                       # captured graphs with a side stream each, like the inference engines of the suite
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            y.copy_(x * 2)
        torch.cuda.current_stream().wait_stream(side)
        y.add_(1)
    g.replay()
    keep.append((g, y, side))
torch.cuda.synchronize()
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(1 << 22, device=dev)
for _ in range(3):
    dist.all_reduce(t)
if barrier:
    dist.barrier()
if sync:
    torch.cuda.synchronize()
dist.destroy_process_group()
print("child ok", float(t[0]))
'''


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("variant (graphs, barrier, sync) -> exit codes of %d children" % n)
    for graphs in (0, 40):
        for barrier in (1, 0):
            for sync in (1, 0):
                codes = []
                for _ in range(n):
                    r = subprocess.run([sys.executable, "-c", CHILD, str(graphs), str(barrier), str(sync)], env=env,
                                       capture_output=True, text=True, timeout=300)
                    codes.append(r.returncode)
                    if r.returncode != 0:
                        print("   stderr tail:", r.stderr.strip().splitlines()[-3:])
                print("graphs=%2d barrier=%d sync=%d -> %s  (%d of %d failed)" % (graphs, barrier, sync, codes,
                                                                                sum(c != 0 for c in codes), n), flush=True)


if __name__ == "__main__":
    main()
