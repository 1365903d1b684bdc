#!/usr/bin/env python3
"""Weight-gradient kernel with and without activation-on-load, per layer (GPU, HIP events).
Usage: python tools/bench_wgrad_act.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 8
LAYERS = [("enc0.3", 32, 32, 512, 1), ("enc1.3", 64, 64, 256, 1), ("enc2.4", 128, 128, 128, 1),
          ("enc3.4", 256, 256, 64, 1), ("enc4.4", 512, 512, 32, 1), ("enc2.0", 64, 128, 256, 2),
          ("dec1.0s", 256, 256, 64, 1)]

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

print(f"{'layer':8s} {'GFLOP':>7s} | {'plain us':>9s} {'TF/s':>6s} | {'act us':>9s} {'TF/s':>6s} | act/plain")
for name, Cx, Cout, H, s in LAYERS:
    Ho = H // s
    x = torch.randn(N, H, H, Cx, device="cuda")
    al = torch.rand(N, Cx, device="cuda") + 0.5
    be = torch.randn(N, Cx, device="cuda")
    dy = torch.randn(N, Ho, Ho, Cout, device="cuda")
    dw = torch.empty(Cout, Cx, 3, 3, device="cuda")
    fl = 2.0 * N * Ho * Ho * 9 * Cx * Cout
    tp = timeit(lambda: ops.conv_in_bwd_weight(ops.Act(x), 0.01, dy, dw, 0, 3, s))
    ta = timeit(lambda: ops.conv_in_bwd_weight(ops.Act(x, al, be), 0.01, dy, dw, 0, 3, s))
    print(f"{name:8s} {fl * 1e-9:7.1f} | {tp * 1e6:9.1f} {fl / tp * 1e-12:6.1f} | {ta * 1e6:9.1f} {fl / ta * 1e-12:6.1f} | {ta / tp:.3f}", flush=True)
