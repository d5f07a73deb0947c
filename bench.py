#!/usr/bin/env python
"""OTPose forward throughput on MI355X: frames/s at 384x288, 5-frame window, batch 16 per GPU.

``python bench.py --gpus N --steps K --warmup W`` (N > 1 is launched by the driver through
torch.distributed.run, one rank per GPU).  A step = one OTPose forward (HRNet-W48 + temporal encoders +
RSB heads + 5x modulated DCN, fp32, eval) over 16 synthetic clips = 80 frames per rank, inputs resident
in HBM.  Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-step", action="store_true", help="skip the training-step probe (extra field of the line)")
    ap.add_argument("--train-f32", action="store_true", help="also time the fp32 training step (1 GPU)")
    ap.add_argument("--no-exact-fp32", action="store_true", help="skip the exact-fp32-kernel forward reported beside the headline")
    ap.add_argument("--no-eager-baseline", action="store_true", help="skip the eager PyTorch-ROCm forward timed beside the headline")
    ap.add_argument("--no-config5", action="store_true", help="skip the 7-frame-window forward (BASELINE configs[4]) reported beside the headline")
    return ap.parse_args(argv)


def launch_ranks(a):
    """``python bench.py --gpus N`` with N > 1 and no launcher environment: start one rank per GPU through
    ``torch.distributed.run`` (the reference's counterpart is the single-process ``nn.DataParallel`` of train.py:78-79 /
    eval.py:112-113).  Runs BEFORE anything touches the GPU - the parent only counts devices, starts the children, relays
    their exit code; rank 0 of the children prints the JSON line on the stdout it inherits.  Fails loudly when the box has
    fewer GPUs than asked for (never a silent 1-rank run)."""
    import socket
    import subprocess
    import torch                                       # device_count() does not initialise the GPU runtime
    have = torch.cuda.device_count()
    if have < a.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this machine - refusing to run fewer ranks\n"
                         % (a.gpus, have))
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % a.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this host driver
    sys.exit(subprocess.call(cmd, env=env))


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse_args()
    if _a.gpus > 1:
        launch_ranks(_a)                               # never returns

import torch                                           # noqa: E402

from otpose_amd import OTPose, cfg2, hip, ops          # noqa: E402
from otpose_amd import synthetic as S                  # noqa: E402

FLOP_PER_CLIP = 409.0e9          # conv + bmm FLOPs of one 5-frame 384x288 W48 clip
DCN_BYTES_PER_CLIP_DIL = 13_630_464   # x + offset + mask read, out written, fp32
PEAK_F32_MATRIX = 157.3e12       # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM = 8.0e12                # MI355X_MICROARCH.md: HBM3E spec peak
PEAK_BF16_MATRIX = 2.5e15        # MI355X_MICROARCH.md: dense bf16 MFMA peak


T_START = time.perf_counter()


def log(msg):
    """Progress to stderr (stdout carries only the JSON line)."""
    print("[bench %7.1fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def host_threads():
    """Threads for the CPU baseline: the cores this process may run on, capped at the GPU box's per-GPU CPU
    share (16); OTPOSE_CPU_THREADS overrides."""
    env = os.environ.get("OTPOSE_CPU_THREADS")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def event_time_ms(fn, iters, stream):
    """Average duration of ``fn`` over ``iters`` back-to-back launches, HIP events on the launch stream."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


def measured_traffic(key, with_source=False):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE / WRITE_SIZE collected
    in separate passes and corrected as MI355X_MICROARCH.md prescribes; see profiles/r02_traffic.json for the split-bf16
    kernels and profiles/r01_traffic.json for the f32 ones).  None when the files or the key are absent: PMC counters cannot
    be read from inside this process."""
    for name in ("r05_traffic.json", "r05_dcn_traffic.json", "r04f_traffic.json", "r04c_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                v = json.load(f).get(key, {}).get("hbm_bytes_per_launch")
        except (OSError, ValueError):
            v = None
        if v is not None:
            return (v, "profiles/" + name) if with_source else v
    return (None, None) if with_source else None


def eager_baseline(model, cfg, x, margin, dev, iters=10):
    """The north star's denominator, "the reference single-GPU PyTorch forward", MEASURED IN THIS RUN: the restated eager graph
    (the oracle's functional restatement of model/OTPose.py:307-394 over the same weights, every op a stock PyTorch-ROCm
    call - MIOpen convolutions with its per-shape search on, rocBLAS matmuls, the DCN as gather ops) at the same batch,
    fp32, HIP events, median of ``iters`` after 3 warm-ups (the first one runs MIOpen's search).  Like ``cpu_baseline`` this
    is a baseline leg: the oracle is timed here as the thing to beat, never used by the product path."""
    from oracle import otpose_oracle as O
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    prev = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = True
    try:
        with torch.no_grad():
            t0 = time.perf_counter()
            for _ in range(3):
                O.otpose_forward(sd, cfg, x, margin)
            torch.cuda.synchronize(dev)
            warm = time.perf_counter() - t0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            times = []
            for _ in range(iters):
                e0.record()
                O.otpose_forward(sd, cfg, x, margin)
                e1.record()
                e1.synchronize()
                times.append(e0.elapsed_time(e1))
    finally:
        torch.backends.cudnn.benchmark = prev
    torch.cuda.empty_cache()
    times.sort()
    return {"eager_forward_ms": times[len(times) // 2], "eager_forward_min_ms": times[0], "warmup_s": warm,
            "source": "measured in this run: oracle graph on stock PyTorch-ROCm ops (MIOpen search on / rocBLAS, DCN as gather ops), "
                      "fp32, batch %d, HIP events, median of %d after 3 warm-ups" % (x.shape[0], len(times))}


def eager_ratio_committed(batch):
    """Fallback when the in-run measurement failed: the figure of tools/eager_baseline.py committed under profiles/."""
    for name in ("r03_eager_baseline.json", "r02_eager_baseline.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                e = json.load(f)
            if int(e.get("batch", -1)) != batch:
                continue
            return {"eager_forward_ms": e["forward_ms"],
                    "source": "committed profile profiles/%s (tools/eager_baseline.py on MI355X, batch %d) - NOT measured in "
                              "this run" % (name, batch)}
        except (OSError, ValueError, KeyError):
            continue
    return None


def kernel_rooflines(dev, batch):
    """Per-kernel roofline points measured live with HIP events on the launch stream: the dominant conv launch
    (MFMA-bound; 48->48 3x3 on the 96x72 branch, 64 of the 553 conv launches of a forward and the largest single
    share of its MACs together with the 96->96 twin) and one DCN call (HBM-bound), both at the shapes the forward uses."""
    st = torch.cuda.current_stream(dev)
    g = torch.Generator().manual_seed(7)
    n = 5 * batch
    # HRNet branch-0 BasicBlock conv: 48 -> 48, 3x3, 96x72, all frames of the batch (22.6 % of the MACs)
    x = torch.randn(n, 48, 96, 72, generator=g).to(dev)
    w = (torch.randn(48, 48, 3, 3, generator=g) * 0.05).to(dev)
    sc, sh = torch.ones(48, device=dev), torch.zeros(48, device=dev)
    out = torch.empty_like(x)
    iv, ov = ops.View(x), ops.View(out)
    wp = ops.pack_conv_weight(w)
    d = ops.conv_desc(iv, ov, 48, 3, 3, 1, 1, 1, act=ops.ACT_RELU)
    conv_flop = 2.0 * 48 * 48 * 9 * 96 * 72 * n
    from otpose_amd.engine import InferenceEngine
    use_x3 = os.environ.get("OTPOSE_CONV_MATH", "x3") != "f32" and ops.x3_supported(d)
    use_wino = (not use_x3 and os.environ.get("OTPOSE_WINOGRAD", "1") != "0" and InferenceEngine.winograd_pays(48, 48)
                and ops.wino_supported(d))
    peak = PEAK_F32_MATRIX
    extra = {}
    use_s8 = (use_x3 and os.environ.get("OTPOSE_S8", "1") != "0"
              and ops.s8_conv_supported(ops.s8_conv_desc(n, 48, 48, 96, 72, ops.ACT_RELU)))
    if use_s8:
        # the kernel the engine runs for this layer inside an HRNet branch (csrc/convs.hip): split-half products on activations
        # kept as MFMA operand records (S8), staged by the LDS-DMA.  A BasicBlock runs it twice: conv1 S8 -> S8 (timed here as
        # `roofline`) and conv2 S8 + C4 residual -> C4 + S8 (`conv2_form`).  Every fp32 product is three f16 MFMA products (same rate as bf16) and
        # a chunk's 9 taps occupy 10 tap slots: the pipe executes 3 * 10/9 of the algorithmic FLOPs, priced at the dense 16-bit peak.
        xs = ops.s8_pack(x)
        ys = ops.s8_empty(n, 48, 96, 72, dev)
        ds = ops.s8_conv_desc(n, 48, 48, 96, 72, ops.ACT_RELU)
        we = ops.x3_weight_exponent(w, sc)            # the weights carry 2^we, the kernel multiplies its sums by 2^-we (engine default)
        ds.out_scale = 2.0 ** -we
        ws = ops.pack_s8_weight(w, sc, we)
        t_conv = event_time_ms(lambda: ops.conv3x3_s8_launch(xs, ws, sh, ds, None, None, ops.S8_F32_C4, ys), 20, st)
        # conv2 of a BasicBlock as the engine runs it since round 4: the block input's S8 records are the residual, S8 out
        rs8 = ops.s8_pack(torch.randn(n, 48, 96, 72, generator=g).to(dev))
        ds2 = ops.s8_conv_desc(n, 48, 48, 96, 72, ops.ACT_RELU)
        ds2.out_scale, ds2.res_layout = ds.out_scale, 1
        t_conv2 = event_time_ms(lambda: ops.conv3x3_s8_launch(xs, ws, sh, ds2, rs8, None, ops.S8_F32_C4, ys), 20, st)
        tr2, _ = measured_traffic("convs_48_48_3x3_96x72_x80_conv2", True)
        kname = ("convs_kernel<3, false, 4> (f16x3 split products, S8 operand records, LDS-DMA, 3 workgroups / CU) 48->48 3x3 @96x72 x%d frames, S8 -> S8 "
                 "(grid %d x 256 threads)" % (n, ((n * 96 * 72 // 256 + 7) // 8) * 8))
        executed = conv_flop * 3.0 * 10.0 / 9.0
        peak = PEAK_BF16_MATRIX
        extra = {"arithmetic": "fp32 accumulate; operands stored as IEEE-half hi | lo records (hi = rne(x), lo = rne(x - hi)), products "
                               "lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x32_f16; weights stored times a per-layer power of two so that both "
                               "pieces are normal numbers (csrc/convs.hip, otp_conv_desc.out_scale)",
                 "pmc": "committed profile, not collected in this run - profiles/r04f_convs_pmc_fold.txt: SQ_INSTS_VALU 11.18 M of which 4.67 M MFMA = 1.40 other vector "
                        "instructions per MFMA (prologue / epilogue), SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 6 %",
                 "conv2_form": {"what": "S8 + S8 residual -> S8 (BasicBlock conv2; round 3: + C4 residual -> C4 + S8, 428 MB)", "ms_per_launch": t_conv2,
                                "achieved": conv_flop / (t_conv2 * 1e-3) / 1e12, "frac": conv_flop / (t_conv2 * 1e-3) / peak,
                                "mfma_pipe_frac": executed / (t_conv2 * 1e-3) / peak,
                                "traffic": tr2, "traffic_source": "committed profile (profiles/r04f_traffic.json: separate --pmc passes of tools/convs_one.py ... conv2s), not collected in this run",
                                "algorithmic_bytes_per_launch": 3.0 * 4 * 48 * 96 * 72 * n,
                                "hbm_frac": (tr2 if tr2 is not None else 3.0 * 4 * 48 * 96 * 72 * n) / (t_conv2 * 1e-3) / PEAK_HBM}}
    elif use_x3:
        # the kernel the engine runs for this layer outside the S8 path: split-half products on the 16-bit matrix cores from fp32
        # NCHW input (csrc/convx.hip).  Every fp32 product is three f16 MFMA products and a chunk's 9 taps occupy 10 tap
        # slots, so the pipe executes 3 * 10/9 of the algorithmic FLOPs - against the dense bf16 peak.
        we = ops.x3_weight_exponent(w, sc)
        d.out_scale = 2.0 ** -we
        xp = ops.pack_x3_weight(w, sc, 1, we)
        t_conv = event_time_ms(lambda: ops.conv2d_x3_launch(iv, xp, sh, ov, d), 20, st)
        kname = "convx_kernel<16,4,1,3> (f16x3 split products) 48->48 3x3 @96x72 x%d frames (grid %d x 256 threads)" % (
            n, ((n * 96 * 72 // 256 + 7) // 8) * 8)
        executed = conv_flop * 3.0 * 10.0 / 9.0
        peak = PEAK_BF16_MATRIX
        extra = {"arithmetic": "fp32 storage / accumulate, products as half hi*hi + hi*lo + lo*hi (csrc/convx.hip)"}
    elif use_wino:
        # the kernel the engine runs for this layer: Winograd F(2x2,3x3) (csrc/wino.hip).  `achieved` stays ALGORITHMIC
        # (direct-convolution) FLOPs per second; the kernel itself executes 16/36 of them on the MFMA pipe.
        up = ops.pack_wino_weight(w)
        t_conv = event_time_ms(lambda: ops.conv2d_wino_launch(iv, up, sc, sh, ov, d), 20, st)
        tiles = n * 48 * 36
        import ctypes
        wplan = (ctypes.c_int * 4)()
        hip.lib().otp_conv2d_wino_last_plan(wplan)
        kname = ("conv_wino_kernel<%d,%d,true> 48->48 3x3 @96x72 x%d frames (grid %d x 256 threads)"
                 % (wplan[0], wplan[1], n, wplan[2]))
        executed = 2.0 * 48 * 48 * 16 * tiles
    else:
        t_conv = event_time_ms(lambda: ops.conv2d_launch(iv, wp, sc, sh, ov, d), 20, st)
        import ctypes
        plan = (ctypes.c_int * 8)()
        hip.lib().otp_conv2d_last_plan(plan)
        kname = ("conv_win_kernel<%d,%d,3,true> 48->48 3x3 @96x72 x%d frames (grid %d x %d threads)"
                 % (plan[0], plan[1], n, plan[5], 64 * plan[2] * plan[3]))
        executed = conv_flop
    traffic, tsrc = measured_traffic("convs_48_48_3x3_96x72_x80" if use_s8 else "convx_48_48_3x3_96x72_x80" if use_x3 else
                                     ("conv_wino_48_48_3x3_96x72_x80" if use_wino else "conv_48_48_3x3_96x72_x80"), True)
    # `achieved` is ALGORITHMIC (direct-convolution) FLOP/s per launch as the bench contract defines it and `frac` is
    # achieved / peak - nothing else.  What the matrix pipe itself executes (split products: 3 MFMA products per fp32 product
    # and 10 tap slots for 9 taps; Winograd: 16/36) is `mfma_pipe_frac` = executed MFMA FLOPs / peak: pipe occupancy, not a
    # roofline fraction.  `hbm_frac` prices the HBM traffic against 8 TB/s.
    conv_bytes = 2.0 * 4 * 48 * 96 * 72 * n                      # algorithmic: input read once, output written once
    conv = {"kernel": kname,
            "bound": "mfma", "achieved": conv_flop / (t_conv * 1e-3) / 1e12, "peak": peak / 1e12,
            "unit": "TFLOP/s", "frac": conv_flop / (t_conv * 1e-3) / peak,
            "mfma_pipe_frac": executed / (t_conv * 1e-3) / peak,
            "hbm_frac": (traffic if traffic is not None else conv_bytes) / (t_conv * 1e-3) / PEAK_HBM,
            "hbm_frac_basis": ("%s (rocprofv3 PMC passes replayed from the committed file, not collected in this run)" % tsrc)
            if traffic is not None else "algorithmic bytes",
            "traffic": traffic,
            "traffic_source": ("committed profile (%s): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel, "
                               "FETCH_SIZE doubled for the 16-byte streams as MI355X_MICROARCH.md prescribes; not collected in "
                               "this run" % tsrc) if traffic is not None else None,
            "ms_per_launch": t_conv, "ms_source": "measured in this run (HIP events on the launch stream, 20 back-to-back launches)",
            "algorithmic_flop_per_launch": conv_flop, "executed_mfma_flop_per_launch": executed}
    conv.update(extra)
    # one DCN call (one dilation) over the batch
    xd = torch.randn(batch, 17, 96, 72, generator=g).to(dev)
    off = (torch.randn(batch, 306, 96, 72, generator=g) * 3).to(dev)
    msk = torch.randn(batch, 153, 96, 72, generator=g).to(dev)
    wd = (torch.randn(17, 17, 3, 3, generator=g) * 0.2).to(dev)
    bd = torch.zeros(17, device=dev)
    od = torch.empty(batch, 17, 96, 72, device=dev)
    L = hip.lib()

    def dcn():
        hip.check(L.otp_mdcn_forward(hip.ptr(xd), hip.ptr(off), hip.ptr(msk), hip.ptr(wd), hip.ptr(bd), hip.ptr(od),
                                     batch, 17, 96, 72, 17, 3, 3, 1, 6, 6, 1, 17, 0.2, 0.0, 0, hip.stream_of(xd)), "dcn")
    t_dcn = event_time_ms(dcn, 20, st)
    dcn_bytes = DCN_BYTES_PER_CLIP_DIL * batch
    # rocprofv3 FETCH_SIZE / WRITE_SIZE of this launch, CALIBRATED for its access width: tools/micro/fetch_calib.hip reads a known
    # byte count as 27 coalesced dword streams per thread (the operator's pattern) - FETCH_SIZE reports half of it, like the 16-byte
    # streams of the guide, WRITE_SIZE all of it (tools/r05_dcn_traffic.sh, profiles/r05_dcn_traffic.json).  Batch 16 only.
    dcn_traffic, dcn_tsrc = measured_traffic("mdcn_fwd_17x96x72_x16", True) if batch == 16 else (None, None)
    dcn_r = {"kernel": "mdcn_fwd_kernel<17,1,true> 17x96x72 x%d clips, one dilation" % batch, "bound": "hbm",
             "achieved": dcn_bytes / (t_dcn * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
             "frac": dcn_bytes / (t_dcn * 1e-3) / PEAK_HBM,
             "traffic": dcn_traffic,
             "traffic_source": ("committed profile (%s): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this launch, FETCH_SIZE "
                                "x 2.0 as calibrated on a copy kernel with the operator's own access pattern (27 coalesced dword streams per "
                                "thread); not collected in this run" % dcn_tsrc) if dcn_traffic is not None else None,
             "hbm_frac_by_traffic": (dcn_traffic / (t_dcn * 1e-3) / PEAK_HBM) if dcn_traffic is not None else None,
             "ms_per_launch": t_dcn, "algorithmic_bytes_per_launch": dcn_bytes}
    # one TransformerBlock ln2 + MLP launch of a temporal encoder (C = 136, hidden 544, T = 96*72) over the batch
    C, HID, T = 136, 544, 96 * 72
    xm = torch.randn(batch, C, T, generator=g).to(dev)
    w1, w2 = (torch.randn(HID, C, 1, generator=g) / C ** 0.5).to(dev), (torch.randn(C, HID, 1, generator=g) / HID ** 0.5).to(dev)
    b1, one, zero = torch.zeros(HID, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)
    om = torch.empty_like(xm)
    mlp_r = None
    x3 = os.environ.get("OTPOSE_CONV_MATH", "x3") != "f32" and ops.mlp_x3_supported(C, HID, T)
    if x3 or ops.mlp_fused_supported(C, HID, T):
        mlp_flop = 4.0 * C * HID * batch * T
        balanced = T % 432 == 0 and batch * (T // 432) >= 192 and os.environ.get("OTP_MLP_BALANCED", "1") != "0"
        if x3:
            # split-half kernel (csrc/mlpx.hip): 3 f16 MFMA products per fp32 product, K padded 136 -> 160 in the first GEMM
            packed = ops.pack_mlp_x3_weights(w1, b1, w2)
            t_mlp = event_time_ms(lambda: ops.ln_mlp_x3(xm, one, zero, 1e-5, packed, one, zero, out=om), 20, st)
            executed = 3.0 * 2.0 * (160 * HID + HID * 144) * batch * T
            nt1 = os.environ.get("OTP_MLP_NT1", "1") != "0" and os.environ.get("OTP_MLP_BALANCED") != "2"
            peak = PEAK_BF16_MATRIX
            kn = ("mlpx_kernel<136,544,8,true,1> (one token tile per wave, two workgroups per CU)" if nt1 else
                  "mlpx_balanced_kernel<136,544,true>" if balanced else "mlpx_kernel<136,544,8,true,2>")
        else:
            packed = ops.pack_mlp_weights(w1, b1, w2)
            t_mlp = event_time_ms(lambda: ops.ln_mlp_fused(xm, one, zero, 1e-5, packed, one, zero, out=om), 20, st)
            executed = mlp_flop
            peak = PEAK_F32_MATRIX
            kn = "mlp_fused_balanced_kernel<136,544,true>" if balanced else "mlp_fused_kernel<136,544,4,true>"
        mlp_r = {"kernel": "%s ln2 + 136->544->gelu->136 + residual, T=6912 x%d clips" % (kn, batch),
                 "bound": "mfma", "achieved": mlp_flop / (t_mlp * 1e-3) / 1e12, "peak": peak / 1e12,
                 "unit": "TFLOP/s", "frac": mlp_flop / (t_mlp * 1e-3) / peak,
                 "mfma_pipe_frac": executed / (t_mlp * 1e-3) / peak,
                 "traffic": measured_traffic("ln_mlp_x3_136_544_T6912_x16" if x3 else "ln_mlp_fused_136_544_T6912_x16"),
                 "traffic_source": "committed profile (profiles/r0*_traffic.json), not collected in this run",
                 "ms_per_launch": t_mlp, "algorithmic_flop_per_launch": mlp_flop, "executed_mfma_flop_per_launch": executed}
    # channel attention of one temporal-encoder block (blocks.py:427-447): S = (q*scale) k^T (68 x 68 per head, contraction
    # over T), softmax, O = P v in the transposed-contiguous image - the QK^T / PV kernels the north star asks the MFMA
    # utilisation of.  4*hs^2*T FLOP per (clip, head); the op is HBM-bound (q, k, v read, out written: 16 B per element).
    nh = 2
    hs = C // nh
    q, k_, v = (torch.randn(batch, C, T, generator=g).to(dev) for _ in range(3))
    t_att = event_time_ms(lambda: ops.chan_attn(q, k_, v, nh, hs ** -0.5), 20, st)
    att_x3 = os.environ.get("OTPOSE_CONV_MATH", "x3") != "f32"
    att_flop = 4.0 * hs * hs * T * nh * batch
    att_bytes = 4.0 * 4 * batch * C * T
    attn_r = {"kernel": "attn_scores_kernel + attn_softmax_kernel + attn_pv_kernel, C=136 nh=2 T=6912 x%d clips" % batch,
              "bound": "hbm", "achieved": att_bytes / (t_att * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
              "frac": att_bytes / (t_att * 1e-3) / PEAK_HBM, "traffic": None, "ms_per_call": t_att,
              "algorithmic_bytes_per_call": att_bytes, "mfma_flop_per_call": att_flop,
              "mfma_achieved_TFLOPs": att_flop / (t_att * 1e-3) / 1e12,
              # the pipe the kernels use: split products = 3 bf16 MFMAs per fp32 product on the 2.5 PFLOP/s bf16 pipe; the f32
              # kernels (OTPOSE_CONV_MATH=f32) run on the 157.3 TFLOP/s f32 MFMA
              "mfma_pipe": "bf16 (3 products per fp32 product)" if att_x3 else "f32",
              "mfma_frac": (3.0 * att_flop / (t_att * 1e-3) / PEAK_BF16_MATRIX) if att_x3
              else att_flop / (t_att * 1e-3) / PEAK_F32_MATRIX}
    # the warping head (ten offset / mask convs + five DCN gathers + weighted sum) as one launch
    head_r = None
    if os.environ.get("OTPOSE_CONV_MATH", "x3") != "f32" and ops.dcn_fused_supported(32, 17, 96, 72, 5):
        dils = (3, 6, 9, 12, 15)
        tr = torch.randn(batch, 32, 96, 72, generator=g).to(dev)
        w_off = [(torch.randn(306, 32, 3, 3, generator=g) / 17.0).to(dev) for _ in dils]
        w_msk = [(torch.randn(153, 32, 3, 3, generator=g) / 17.0).to(dev) for _ in dils]
        w_dcn = [(torch.randn(17, 17, 3, 3, generator=g) * 0.2).to(dev) for _ in dils]
        packed = ops.pack_dcn_fused(w_off, w_msk, w_dcn, [None] * 5)
        ws = torch.empty(hip.lib().otp_dcn_fused_workspace(batch, 96, 72) // 4, dtype=torch.int32, device=dev)
        t_head = event_time_ms(lambda: ops.dcn_fused(tr, xd, packed, dils, 0.2, out=od, workspace=ws), 10, st)
        conv_fl = 2.0 * 459 * 32 * 9 * 96 * 72 * batch * 5                     # the ten 32 -> 306 / 153 convs
        unfused_bytes = (DCN_BYTES_PER_CLIP_DIL + 2 * 459 * 6912 * 4.0) * batch * 5   # DCN streams + conv outputs written
        head_r = {"kernel": "dcn_fused_kernel<17> + dcnf_split_kernel: 5 dilations x (32->306, 32->153 convs + DCN 17->17), "
                            "96x72 x%d clips, one launch" % batch,
                  "bound": "lds", "ms_per_launch": t_head, "algorithmic_conv_flop_per_launch": conv_fl,
                  "achieved": conv_fl / (t_head * 1e-3) / 1e12, "unit": "TFLOP/s (conv part, algorithmic)",
                  "peak": PEAK_BF16_MATRIX / 1e12, "frac": conv_fl / (t_head * 1e-3) / PEAK_BF16_MATRIX,
                  "mfma_pipe_frac": conv_fl * 3.0 * 32.0 / 27.0 / (t_head * 1e-3) / PEAK_BF16_MATRIX,
                  "hbm_bytes_per_launch": (32 * 3 + 17 * 2) * 6912 * 4.0 * batch,
                  "hbm_bytes_of_the_unfused_launches": unfused_bytes,
                  "note": "offsets / masks never leave the chip; bound by LDS fragment reads (DESIGN.md section 3.2b)"}
    return conv, dcn_r, mlp_r, attn_r, head_r


def forward_roofline(alg_flops_per_s, math):
    """Whole-forward MFMA roofline against the pipe the kernels run on: `frac` = algorithmic FLOP/s / peak.  Split mode: every
    fp32 product is three bf16 MFMA products, so the pipe executes >= 3x the algorithmic FLOPs (tap-slot / channel padding on
    top, not counted here) on the 2.5 PFLOP/s dense bf16 pipe (MI355X_MICROARCH.md, Matrix cores) - `mfma_pipe_frac`;
    exact mode: the 157.3 TFLOP/s f32 MFMA."""
    if math == "f32":
        return {"bound": "mfma", "pipe": "f32 MFMA", "achieved": alg_flops_per_s / 1e12, "peak": PEAK_F32_MATRIX / 1e12,
                "unit": "TFLOP/s", "frac": alg_flops_per_s / PEAK_F32_MATRIX, "traffic": None}
    return {"bound": "mfma", "pipe": "bf16 MFMA, 3 products per fp32 product", "achieved": alg_flops_per_s / 1e12,
            "executed": 3.0 * alg_flops_per_s / 1e12, "peak": PEAK_BF16_MATRIX / 1e12, "unit": "TFLOP/s",
            "frac": alg_flops_per_s / PEAK_BF16_MATRIX, "mfma_pipe_frac": 3.0 * alg_flops_per_s / PEAK_BF16_MATRIX,
            "traffic": None}


def golden_parity(model, cfg, dev):
    """Max-abs difference of the 7 forward outputs against the committed reference-generated golden of this very
    configuration (tests/golden/e2e_cfg2_b1.npz: one 384x288 W48 clip run through the reference model in the build
    container, same seeded weights / inputs).  None when the fixture did not travel."""
    import numpy as np
    path = os.path.join(ROOT, "tests", "golden", "e2e_cfg2_b1.npz")
    if not os.path.isfile(path):
        return None
    z = np.load(path)
    x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE)
    with torch.no_grad():
        outs = [o.clone() for o in model(x.to(dev), margin=margin.to(dev))]
    names = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
    deltas = {n: float((o.cpu() - torch.from_numpy(z[n])).abs().max()) for n, o in zip(names, outs) if n in z.files}
    ranges = {n: float(np.abs(z[n]).max()) for n in deltas}
    return {"max_abs_delta": max(deltas.values()), "max_abs_delta_output_heatmaps": deltas.get("output"),
            "max_rel_delta": max(deltas[n] / max(ranges[n], 1e-30) for n in deltas),
            "per_output": {n: {"max_abs_delta": deltas[n], "max_abs_ref": ranges[n]} for n in deltas},
            "tolerance": 1e-3, "vs": "tests/golden/e2e_cfg2_b1.npz (reference model, 1 clip of this config)"}


def train_step_probe(cfg, dev, batch, dtype="bf16", steps=3, dist=None):
    """BASELINE configs[2] (1 GPU) / configs[3] (data parallel): one optimisation step as the reference runs it
    (script/Common.py:118-144) - forward under model.train() (BatchNorm batch statistics, dropout / drop-path), the two
    ST_OHKW terms with the per-joint flags MAX-reduced over ranks, backward through the HIP kernels, RCCL all-reduce of the
    flat gradient buffers (world > 1), fused global-norm clip + AdamW on fp32 master weights.  dtype "bf16": backbone,
    MLP interiors and offset / mask convs on bf16 activations + bf16 matrix cores (fp32 accumulation and statistics).
    Every rank runs it; the time is the max over ranks.  Reported next to the headline metric, never part of `value`."""
    from otpose_amd import parallel as PAR
    from otpose_amd.optim import FusedAdamW
    world = dist.get_world_size() if dist is not None else 1
    model = OTPose(cfg)
    S.fill_synthetic_(model)                      # identical seeded replicas
    model = model.to(dev).train()
    model.train_dtype = dtype
    x, margin = S.synthetic_clip(batch, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.to(dev), margin.to(dev)
    J = cfg.MODEL.NUM_JOINTS
    w, h = cfg.MODEL.HEATMAP_SIZE
    gen = torch.Generator().manual_seed(11)
    g = (torch.rand(batch, J, h, w, generator=gen) * 0.2).to(dev)
    g[:, ::2, 3, 4] = 1.0
    wt = (torch.rand(batch, J, 1, generator=gen) > 0.15).float().to(dev)
    opt = FusedAdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)
    times = []
    loss = None
    torch.cuda.reset_peak_memory_stats(dev)
    for it in range(steps + 1):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        loss = PAR.train_step_dp(model, opt, x, margin, g, wt)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        if it:
            times.append(time.perf_counter() - t0)
    t = torch.tensor([sorted(times)[len(times) // 2]], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    t = float(t.item())
    # the same steps back to back (no synchronisation between them: the host enqueues step i + 1 while the device runs step i).  The
    # synchronised figure above is what a loop sees that reads its results every iteration, as the reference's does (accuracy of the
    # outputs on the CPU, script/Common.py:147) - there the forward starts with an idle device and is bound by the host's ~35 ms of launches
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(4):
        loss = PAR.train_step_dp(model, opt, x, margin, g, wt)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    tf = torch.tensor([(time.perf_counter() - t0) / 4], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tf, op=dist.ReduceOp.MAX)
    tf = float(tf.item())
    flop = 3.0 * FLOP_PER_CLIP * batch * world               # forward + input gradients + weight gradients
    res = {"ms_per_step": 1e3 * t, "frames_per_s": 5 * batch * world / t, "ms_per_step_free_running": 1e3 * tf,
           "clips_per_gpu": batch, "n_gpus": world,
           "dtype": dtype, "collective": ("RCCL all-reduce of the flat fp32 gradient buffers, world %d" % world) if world > 1
           else "none (1 GPU)",
           "what": "forward (train mode) + 2x ST_OHKW loss + backward + grad all-reduce + clip + AdamW, median of %d after "
                   "1 warm-up, every step synchronised, max over ranks; ms_per_step_free_running: 4 steps back to back" % steps,
           "loss_finite": bool(torch.isfinite(loss.detach()).all()),
           "peak_mem_GB": torch.cuda.max_memory_allocated(dev) / 2 ** 30,
           "roofline": {"bound": "mfma", "achieved": flop / t / 1e12 / world,
                        "peak": (PEAK_BF16_MATRIX if dtype == "bf16" else PEAK_F32_MATRIX) / 1e12, "unit": "TFLOP/s per GPU",
                        "frac": flop / t / world / (PEAK_BF16_MATRIX if dtype == "bf16" else PEAK_F32_MATRIX),
                        "traffic": None,
                        "note": "algorithmic 3 x 409 GFLOP per clip; the step is GPU-bound (host side 85-100 ms, hidden; a one-stream "
                                "hipGraph replays no faster than the one-stream eager step: profiles/r04_train_graph_probe.txt) - "
                                "~162 ms of kernel time per step (round 5; 189 before), of which the HRNet's convolutions 22, its BatchNorm "
                                "passes 17 (at their HBM roofline), its weight gradients 21; ~35 ms of the step are temporal-encoder "
                                "launches running one at a time (profiles/r05_train_timeline.txt, DESIGN.md section 3.6)"}}
    # the exchange's own wall time (an extra, untimed step with the collective bracketed by device synchronisations): what a scaling
    # run loses between the last backward kernel and the optimizer - so that a SCALE line can attribute its lost efficiency
    stats = {}
    PAR.train_step_dp(model, opt, x, margin, g, wt, stats=stats)
    res["comm_ms"], res["comm_bytes"], res["comm_overlap"] = stats.get("comm_ms"), stats.get("comm_bytes"), stats.get("overlap")
    del model, opt, loss
    torch.cuda.empty_cache()
    if world == 1 and dtype == "bf16" and os.environ.get("OTPOSE_BENCH_TRAIN_CHECKS", "1") != "0":
        # cfg3 at FULL size is more than `loss_finite`: the step from seeded weights twice - same loss and same
        # updated weights to the last bit (every order-dependent sum of the step is fixed-order or integer: DESIGN.md section 4) -
        # and once in fp32 on the same clips and dropout masks: the bf16 loss against the fp32 loss of the same HIP graph
        def one_step(dt):
            m = OTPose(cfg)
            S.fill_synthetic_(m)
            m = m.to(dev).train()
            m.train_dtype = dt
            o = FusedAdamW([p for p in m.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01, max_grad_norm=1.0)
            torch.manual_seed(4242)
            torch.cuda.manual_seed(4242)                        # the dropout / drop-path masks of the step
            l = float(PAR.train_step_dp(m, o, x, margin, g, wt))
            torch.cuda.synchronize(dev)
            wsum = float(sum(p.detach().double().sum() for p in m.parameters()))
            wabs = float(sum(p.detach().double().abs().sum() for p in m.parameters()))
            del m, o
            torch.cuda.empty_cache()
            return l, wsum, wabs
        a1, b1, c1 = one_step("bf16")
        a2, b2, c2 = one_step("bf16")
        a3, _, _ = one_step("f32")
        res["full_size_checks"] = {"what": "batch %d x 5 x 384x288, one step from the seeded weights, dropout seeds fixed" % batch,
                                   "loss_bf16": a1, "loss_bf16_again": a2, "bit_reproducible": a1 == a2 and b1 == b2 and c1 == c2,
                                   "loss_fp32_same_graph": a3, "loss_rel_diff_bf16_vs_fp32": abs(a1 - a3) / max(abs(a3), 1e-30)}
    return res


def cpu_baseline(batch=16):
    """The oracle (CPU restatement of the reference graph) timed on the host cores on the metric's own workload: after a 1-clip
    warm-up, ONE forward over the whole batch (16 clips x 5 frames, 384x288, W48: ~20-30 s on 16 cores) - baseline and metric share a
    workload.  A box on which the warm-up predicts more than ~60 s for the batch gets a proportionally smaller batch (stated)."""
    from oracle import otpose_oracle as O
    cfg = cfg2()
    m = OTPose(cfg)
    S.fill_synthetic_(m)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    cores = host_threads()
    torch.set_num_threads(cores)
    with torch.no_grad():
        x, margin = S.synthetic_clip(1, cfg.MODEL.IMAGE_SIZE)
        t0 = time.perf_counter()
        O.otpose_forward(sd, cfg, x, margin)
        t1 = time.perf_counter() - t0
        log("cpu_baseline: 1-clip warm-up %.2f s (%d threads)" % (t1, cores))
        # (a batch forward costs less per clip than the warm-up clip did: ~1.3 s per clip on 16 cores)
        clips = batch if t1 * batch * 0.6 <= 60.0 else max(1, int(60.0 / (t1 * 0.6)))
        x, margin = S.synthetic_clip(clips, cfg.MODEL.IMAGE_SIZE)
        t0 = time.perf_counter()
        O.otpose_forward(sd, cfg, x, margin)
        t = time.perf_counter() - t0
        log("cpu_baseline: oracle forward over %d clip(s) took %.2f s" % (clips, t))
    return {"value": 5.0 * clips / t, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%s: %d clip(s) (%d frames) 384x288 HRNet-W48 in one forward of the oracle (torch-CPU restatement) after a 1-clip "
                      "warm-up, %.2f s" % ("the whole batch of the metric" if clips == batch else "a bounded part of the batch", clips,
                                           5 * clips, t)}


def config5_probe(dev, batch, steps):
    """BASELINE configs[4] (extension: the reference has no 7-frame model, no RSN backbone, no fp16 forward): batch 16 x 7-frame
    window x 384x288 in the dtype BASELINE.json states - fp16 (cfg.MODEL.DTYPE = "fp16": the backbone on half activations, csrc/h16.hip,
    the encoders' matrix kernels on half operands) - with the fp32 engine on the same clips beside it: its time, and the max
    difference of the 7 outputs (the self-consistency SURVEY section 7 prescribes for a configuration without an oracle).  The
    dominant launch of the fp16 backbone gets its own roofline point.  1 GPU only, never part of `value`."""
    from otpose_amd.config import cfg5
    names = ("output", "rough", "intersection", "prev_b", "context", "squeezed", "total_b")
    x5, g5 = S.synthetic_clip(batch, cfg5().MODEL.IMAGE_SIZE, frames=7)
    x5, g5 = x5.to(dev), g5.to(dev)
    res = {}
    outs = {}
    for dt in ("fp32", "fp16"):
        m5 = OTPose(cfg5(dt))
        S.fill_synthetic_(m5)
        m5 = m5.to(dev).eval()
        m5.alias_outputs = True
        with torch.no_grad():
            for _ in range(3):
                o5 = m5(x5, margin=g5)
            torch.cuda.synchronize(dev)
            t5 = time.perf_counter()
            for _ in range(steps):
                o5 = m5(x5, margin=g5)
            torch.cuda.synchronize(dev)
            d5 = time.perf_counter() - t5
        res[dt] = {"frames_per_s": 7 * batch * steps / d5, "ms_per_step": 1e3 * d5 / steps,
                   "outputs_finite": all(bool(torch.isfinite(o).all()) for o in o5)}
        outs[dt] = [o.clone() for o in o5]
        m5.check_range()
        del m5, o5
        torch.cuda.empty_cache()
    delta = {n: {"max_abs_delta": float((a - b).abs().max()), "max_abs_fp32": float(b.abs().max())}
             for n, a, b in zip(names, outs["fp16"], outs["fp32"])}
    # dominant launch of the fp16 backbone: 48 -> 48 3x3 @96x72 over the 112 frames of the batch (BasicBlock conv2: + residual + ReLU)
    n = 7 * batch
    g = torch.Generator().manual_seed(9)
    xi = ops.h8_pack(torch.randn(n, 48, 96, 72, generator=g).to(dev))
    ri = ops.h8_pack(torch.randn(n, 48, 96, 72, generator=g).to(dev))
    wt = (torch.randn(48, 48, 3, 3, generator=g) * 0.05).to(dev)
    k = ops.h16_weight_exponent(wt)
    wp = ops.pack_h16_conv_weight(wt, None, k)
    sh = torch.zeros(48, device=dev)
    yo = ops.h8_empty(n, 48, 96, 72, dev)
    d1 = ops.h16_conv_desc(xi, 48, 1, ops.ACT_RELU, yo, None, k)
    d2 = ops.h16_conv_desc(xi, 48, 1, ops.ACT_RELU, yo, ri, k)
    st = torch.cuda.current_stream(dev)
    t1 = event_time_ms(lambda: ops.h16_conv3x3(xi, wp, sh, 48, 1, ops.ACT_RELU, None, out=yo, k=k, desc=d1), 20, st)
    t2 = event_time_ms(lambda: ops.h16_conv3x3(xi, wp, sh, 48, 1, ops.ACT_RELU, ri, out=yo, k=k, desc=d2), 20, st)
    flop = 2.0 * 48 * 48 * 9 * 96 * 72 * n
    b1, b2 = 2.0 * 2 * 48 * 96 * 72 * n, 3.0 * 2 * 48 * 96 * 72 * n
    roof = {"kernel": "h16_conv3x3_kernel<3, 4, 1> (half operands, one f16 MFMA per product, H8 records, window of 48 channels per HBM round "
                      "trip by LDS-DMA, weights streamed through registers, 3 workgroups / CU) 48->48 3x3 @96x72 x%d frames, H8 -> H8" % n,
            "bound": "mfma", "achieved": flop / (t1 * 1e-3) / 1e12, "peak": PEAK_BF16_MATRIX / 1e12, "unit": "TFLOP/s",
            "frac": flop / (t1 * 1e-3) / PEAK_BF16_MATRIX,
            "mfma_pipe_frac": flop * 10.0 / 9.0 / (t1 * 1e-3) / PEAK_BF16_MATRIX,
            "hbm_frac": b1 / (t1 * 1e-3) / PEAK_HBM, "hbm_frac_basis": "algorithmic bytes (half in, half out)", "traffic": None,
            "ms_per_launch": t1, "algorithmic_flop_per_launch": flop, "algorithmic_bytes_per_launch": b1,
            "conv2_form": {"what": "+ H8 residual + ReLU (a BasicBlock's conv2)", "ms_per_launch": t2, "achieved": flop / (t2 * 1e-3) / 1e12,
                           "frac": flop / (t2 * 1e-3) / PEAK_BF16_MATRIX, "hbm_frac": b2 / (t2 * 1e-3) / PEAK_HBM,
                           "algorithmic_bytes_per_launch": b2},
            "note": "one product per multiply leaves 2.9 k MFMA cycles per 256-pixel tile against ~2 k cycles of vector set-up / epilogue "
                    "work and an HBM floor of the same order: the launch is bound by instruction issue, not by one pipe "
                    "(profiles/r05_h16_conv_phase_stamps.txt)"}
    return {"workload": "BASELINE configs[4] as far as the reference defines it: batch %d x 7-frame window x 384x288, HRNet-W48 (no RSN "
                        "backbone / occlusion mask exists in the reference: model/OTPose.py:200, model/blocks.py:399-453), fp16" % batch,
            "dtype": "fp16", "frames_per_s": res["fp16"]["frames_per_s"], "ms_per_step": res["fp16"]["ms_per_step"],
            "outputs_finite": res["fp16"]["outputs_finite"],
            "arithmetic": "backbone activations stored as IEEE half (H8 records), weights rounded to half once (times a per-layer power of "
                          "two), one v_mfma_f32_16x16x32_f16 per product, fp32 accumulation / shift / residual; encoder MLP / projections / "
                          "q-k-v front end on half operands with fp32 LayerNorm, softmax and accumulation; fp32 tensors behind the backbone",
            "fp32_engine_same_clips": dict(res["fp32"], dtype="f32 storage+accumulate / f16x3 split products"),
            "speedup_vs_fp32_engine": res["fp32"]["ms_per_step"] / res["fp16"]["ms_per_step"],
            "self_consistency_fp16_vs_fp32_engine": delta,
            "tolerance": "8e-3 of max(1, range) per output (tests/test_gpu_h16_engine.py; measured 0.8e-3 .. 2.7e-3)",
            "roofline": roof}


def main():
    a = parse_args()

    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a version banner through C stdio) get stderr
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        # the launcher's world and --gpus must agree: a line that says n_gpus = N was measured on N ranks
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d (launch N ranks with torch.distributed.run, or run "
                         "`python bench.py --gpus N` and let it start them)\n" % (a.gpus, world))
        sys.exit(2)
    dist = None
    if world > 1 or os.environ.get("OTPOSE_BENCH_DIST") == "1":     # env: exercise the RCCL path on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    torch.set_num_threads(host_threads())
    cfg = cfg2()
    log("building OTPose (HRNet-W48) + seeded synthetic weights")
    model = OTPose(cfg)
    S.fill_synthetic_(model)                      # identical seeded weights on every rank
    model = model.to(dev).eval()
    log("model on %s" % dev)
    x, margin = S.synthetic_clip(a.batch, cfg.MODEL.IMAGE_SIZE)
    x, margin = x.to(dev), margin.to(dev)         # inputs resident in HBM before the timed region ...
    xb, mb = model.input_buffers(a.batch, dev)    # ... in the tensors the engine reads: a loader that fills them in place
    xb.copy_(x)                                   # (OTPose.input_buffers) saves the 106 MB device-to-device copy per forward
    mb.copy_(margin.to(mb.dtype))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for i in range(max(a.warmup, 1)):
            outs = model(xb, margin=mb)
            if i == 0:
                torch.cuda.synchronize(dev)
                log("engine built, first forward done")
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            outs = model(xb, margin=mb)
        barrier()
        dt = time.perf_counter() - t0
        # the same loop fed from a tensor of the CALLER's (what script/Common.py:116-118 does every iteration): the engine copies the
        # 106 MB clip tensor into its input buffer first - a device-to-device copy that `ms_per_step` (inputs resident in the
        # engine's own buffers, OTPose.input_buffers) does not contain; reported beside it, never part of `value`
        t1 = time.perf_counter()
        for _ in range(a.steps):
            outs = model(x, margin=margin)
        torch.cuda.synchronize(dev)
        dt_copy = time.perf_counter() - t1
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    finite = bool(torch.isfinite(outs[0]).all())
    # the same forward on the exact-fp32 MFMA kernels of round 1 (OTPOSE_CONV_MATH=f32), reported beside the headline so that
    # both arithmetic choices are on one line; 1 GPU only, never part of `value`
    exact = None
    math = os.environ.get("OTPOSE_CONV_MATH", "x3")
    if world == 1 and math != "f32" and not a.no_exact_fp32:
        os.environ["OTPOSE_CONV_MATH"] = "f32"
        hip.lib().otp_chan_attn_set_split(0)                    # the library-wide switch is read once at load time
        try:
            model.invalidate_engine()
            with torch.no_grad():
                for _ in range(2):
                    model(x, margin=margin)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    model(x, margin=margin)
                torch.cuda.synchronize(dev)
                de = time.perf_counter() - t1
            exact = {"frames_per_s": 5 * a.batch * a.steps / de, "ms_per_step": 1e3 * de / a.steps,
                     "kernels": "f32 MFMA: Winograd F(2x2,3x3) / direct convs, csrc/mlp.hip, csrc/dense.hip, f32 attention products, "
                                "unfused warping head",
                     "parity": golden_parity(model, cfg, dev)}
        finally:
            os.environ["OTPOSE_CONV_MATH"] = math
            hip.lib().otp_chan_attn_set_split(1)
            model.invalidate_engine()
    # the eager PyTorch-ROCm forward of the same graph, timed here (1 GPU, rank 0): the denominator of the 5x target
    eager = None
    if world == 1 and not a.no_eager_baseline:
        try:
            eager = eager_baseline(model, cfg, x, margin, dev)
            log("eager PyTorch-ROCm baseline: %.1f ms per forward (warm-up incl. MIOpen search %.1f s)"
                % (eager["eager_forward_ms"], eager["warmup_s"]))
        except Exception as exc:                                   # noqa: BLE001 - the baseline leg must not take the line down
            log("eager baseline failed (%r): falling back to the committed figure" % (exc,))
            eager = eager_ratio_committed(a.batch)
    # BASELINE configs[4] (extension: the reference has no 7-frame model): the same forward with a 7-frame window, batch 16 x 7 x
    # 384 x 288 - 12 x 17 = 204 stacked maps per temporal encoder; 1 GPU only, never part of `value`
    config5 = None
    if world == 1 and not a.no_config5 and a.batch == 16:
        config5 = config5_probe(dev, a.batch, a.steps)
    train = None
    if not a.no_train_step:
        del outs
        model._engine = None                                    # free the inference engine's buffers before the training probe
        torch.cuda.empty_cache()
        train = train_step_probe(cfg, dev, a.batch, "bf16", dist=dist)      # every rank takes part (RCCL collectives)
        if rank == 0:
            log("training-step probe (bf16, %d GPU) done: %.1f ms" % (world, train["ms_per_step"]))

    if rank == 0:
        frames = 5 * a.batch * world * a.steps
        fwd_per_s = a.steps / dt
        line = {
            "metric": "frames/sec at 384x288, 5-frame window, batch 16; heatmap max-abs delta vs ref",
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "ms_per_step_incl_input_copy": 1e3 * dt_copy / a.steps,
            # the arithmetic that was timed: every tensor, accumulator, normalisation and activation is fp32; the products of the
            # convolutions / projections / MLPs / attention are three f16 MFMA products of two-piece fp32 operands (IEEE-half pieces, 22 significand
            # bits per operand; the reference's fp32 is 24, its default TF32 convolutions on NVIDIA hardware 11).  Measured on the
            # headline batch (profiles/r04_headline_parity_probe.txt): heat-maps within 4.5e-5 of the fp32 oracle on all 16 clips,
            # the exact-fp32 kernels within 1.5e-5, the fp32 oracle itself 9e-6 from its fp64 run.  The same forward
            # on exact-fp32 MFMA kernels is `exact_fp32_kernels` of this line.
            "dtype": "f32" if math == "f32" else "f32 storage+accumulate / f16x3 split products", "data": "synthetic",
            "rccl_world": dist.get_world_size() if dist is not None else 1,
            "arithmetic": ("fp32 accumulation everywhere; conv / MLP / projection products as three f16 MFMA products of two-piece "
                           "operands (a = hi + lo in IEEE half, |a - hi - lo| <= max(2^-23 |a|, 2^-25); HRNet / warping-head weights stored times "
                           "a per-layer power of two so both pieces are normal: DESIGN.md section 3.1c); tensors are fp32 except, "
                           "inside an HRNet branch, the tensors between the convolutions of its BasicBlocks, which exist only as that hi | lo "
                           "pair - conv inputs and, since round 4, the residual too (22 significand bits: DESIGN.md section 3.1d)"
                           if math != "f32" else "fp32 throughout (f32 MFMA)"),
            "config": {"workload": "BASELINE configs[1]: batch %d x 5-frame x 384x288, HRNet-W48 + DCN warp + "
                                   "ConvVideoTransformer, fp32 forward (eval), seeded synthetic weights" % a.batch,
                       "clips_per_gpu": a.batch, "frames_per_step_per_gpu": 5 * a.batch, "parallelism": "dp%d" % world,
                       "outputs_finite": finite},
            "roofline_forward": forward_roofline(FLOP_PER_CLIP * a.batch * fwd_per_s, math),
        }
        log("timed region done: %.2f ms/step" % (1e3 * dt / a.steps))
        conv, dcn, mlp, attn, head = kernel_rooflines(dev, a.batch)
        log("kernel rooflines done")
        line["roofline"] = conv
        line["roofline_dcn"] = dcn
        if mlp is not None:
            line["roofline_mlp"] = mlp
        line["roofline_attn"] = attn
        if head is not None:
            line["roofline_warp_head"] = head
        if exact is not None:
            line["exact_fp32_kernels"] = exact
        if config5 is not None:
            line["config5"] = config5
        if eager is None:
            eager = eager_ratio_committed(a.batch)
        if eager is not None:
            ms = 1e3 * dt / a.steps
            eager = dict(eager)
            eager["target"] = 5.0
            eager["speedup"] = eager["eager_forward_ms"] / ms                  # the headline arithmetic (`dtype` of this line)
            eager["speedup_arithmetic"] = line["dtype"]
            if exact is not None:
                eager["speedup_exact_fp32_kernels"] = eager["eager_forward_ms"] / exact["ms_per_step"]
            line["vs_eager_rocm"] = eager
        line["parity"] = golden_parity(model, cfg, dev)     # the "heatmap max-abs delta vs ref" half of the metric
        log("golden parity done")
        if train is not None:
            line["train_step"] = train
        if world == 1 and a.train_f32:
            line["train_step_f32"] = train_step_probe(cfg, dev, a.batch, "f32")
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a.batch)
        print(json.dumps(line), file=json_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
