#!/usr/bin/env python3
"""Per-layer timing of the matrix-core kernels (GPU): forward, data gradient and weight gradient
of every 3x3 conv of Our_UNet at bs=8, 512x512, with HIP events on the launch stream.
Usage: python tools/bench_conv.py [fwd|dgrad|wgrad|all] [reps] [fp32|bf16|bf16x3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import unet_implementations_amd as ua

ops = ua.ops
which = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
N = 8
# name, C0, C1, Cout, H(in), stride
LAYERS = [
    ("enc0.3", 32, 0, 32, 512, 1), ("enc1.0", 32, 0, 64, 512, 2), ("enc1.3", 64, 0, 64, 256, 1),
    ("enc2.0", 64, 0, 128, 256, 2), ("enc2.4", 128, 0, 128, 128, 1), ("enc3.0", 128, 0, 256, 128, 2),
    ("enc3.4", 256, 0, 256, 64, 1), ("enc4.0", 256, 0, 512, 64, 2), ("enc4.4", 512, 0, 512, 32, 1),
    ("enc5.0", 512, 0, 512, 32, 2), ("enc5.4", 512, 0, 512, 16, 1),
    ("dec0.0", 512, 512, 512, 32, 1), ("dec1.0", 512, 256, 256, 64, 1),
    ("dec2.0", 256, 128, 128, 128, 1), ("dec3.0", 128, 64, 64, 256, 1), ("dec4.0", 64, 32, 32, 512, 1),
]


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
print(f"{'layer':8s} {'GFLOP':>8s} | {'fwd us':>8s} {'TF/s':>6s} | {'dgrad us':>8s} {'TF/s':>6s} | {'wgrad us':>8s} {'TF/s':>6s}")
for name, C0, C1, Cout, H, s in LAYERS:
    Cin = C0 + C1
    Ho = H // s
    x0 = torch.randn(N, H, H, C0, device="cuda")
    x1 = torch.randn(N, H, H, C1, device="cuda") if C1 else None
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    b = torch.zeros(Cout, device="cuda")
    wf, wd = ops.pack_conv3x3_weights(w)
    wf3, wd3 = ops.pack_conv3x3_weights_bf16x3(w) if prec == "bf16x3" else (None, None)
    dy = torch.randn(N, Ho, Ho, Cout, device="cuda")
    dw = torch.empty_like(w)
    flops = 2.0 * N * Ho * Ho * 9 * Cin * Cout
    line = f"{name:8s} {flops * 1e-9:8.1f} |"
    if which in ("fwd", "all"):
        t = timeit(lambda: ops.conv3x3_fwd(x0, x1, wf, b, s, bf16=prec, wf3=wf3))
        tot["fwd"][0] += flops; tot["fwd"][1] += t
        line += f" {t * 1e6:8.1f} {flops / t * 1e-12:6.1f} |"
    else:
        line += f" {'-':>8s} {'-':>6s} |"
    if which in ("dgrad", "all"):
        def dg():
            ops.conv3x3_bwd_data(dy, wd, 0, C0, H, H, s, bf16=prec, wd3=wd3)
            if C1:
                ops.conv3x3_bwd_data(dy, wd, C0, C1, H, H, s, bf16=prec, wd3=wd3)
        t = timeit(dg)
        tot["dgrad"][0] += flops; tot["dgrad"][1] += t
        line += f" {t * 1e6:8.1f} {flops / t * 1e-12:6.1f} |"
    else:
        line += f" {'-':>8s} {'-':>6s} |"
    if which in ("wgrad", "all"):
        def wg():
            ops.conv3x3_bwd_weight(x0, dy, dw, 0, s, bf16=prec)
            if C1:
                ops.conv3x3_bwd_weight(x1, dy, dw, C0, s, bf16=prec)
        t = timeit(wg)
        tot["wgrad"][0] += flops; tot["wgrad"][1] += t
        line += f" {t * 1e6:8.1f} {flops / t * 1e-12:6.1f}"
    print(line, flush=True)
    del x0, x1, w, wf, wd, dy, dw
for k, (f, t) in tot.items():
    if t > 0:
        print(f"total {k}: {t * 1e3:.2f} ms  {f / t * 1e-12:.1f} TF/s")
