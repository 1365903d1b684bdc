#!/usr/bin/env python3
"""Winograd vs direct kernels on the net's C -> C stride-1 layers (GPU, HIP events): fused
forward and data gradient with the BSTATS epilogue.  Usage: python tools/bench_wino.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 8
LAYERS = [("enc1.3", 64, 256), ("enc2.4", 128, 128), ("enc3.4", 256, 64), ("enc4.4", 512, 32)]
UP = [("dec0.0", 512, 512, 512, 32), ("dec1.0", 512, 256, 256, 64), ("dec2.0", 256, 128, 128, 128),
      ("dec3.0", 128, 64, 64, 256)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


print(f"{'layer':8s} {'GFLOP':>7s} | fwd direct us  TF/s | fwd wino us  alg TF/s | dgrad direct us | dgrad wino us  alg TF/s")
for name, C, H in LAYERS:
    x = torch.randn(N, H, H, C, device="cuda")
    al = torch.rand(N, C, device="cuda") + 0.5
    be = torch.randn(N, C, device="cuda")
    w = torch.randn(C, C, 3, 3, device="cuda") * (2.0 / (9 * C)) ** 0.5
    b = torch.zeros(C, device="cuda")
    g1 = torch.ones(C, device="cuda"); b1 = torch.zeros(C, device="cuda")
    wf, wd = ops.pack_conv3x3_weights(w)
    uf, ud = ops.pack_wino_weights(w)
    fl = 2.0 * N * H * H * 9 * C * C
    src = ops.Act(x, al, be)
    t_fd = timeit(lambda: ops.conv_in_fwd(src, None, 0.01, wf, b, 3, 1, g1, b1, 1e-5, None))
    t_fw = timeit(lambda: ops.conv_in_fwd(src, None, 0.01, wf, b, 3, 1, g1, b1, 1e-5, None, wu=uf))
    y, st = ops.conv_in_fwd(src, None, 0.01, wf, b, 3, 1, g1, b1, 1e-5, None)
    dy = torch.randn(N, H, H, C, device="cuda")
    def dg(u):
        nn = ops.NextNorm(y, st, g1, b1, None, 0.01)
        return ops.conv3x3_bwd_data(dy, wd, 0, C, H, H, 1, nxt=nn, ud=u)
    t_dd = timeit(lambda: dg(None))
    t_dw = timeit(lambda: dg(ud))
    print(f"{name:8s} {fl * 1e-9:7.1f} | {t_fd * 1e6:9.1f} {fl / t_fd * 1e-12:6.1f} | {t_fw * 1e6:9.1f} "
          f"{fl / t_fw * 1e-12:6.1f} | {t_dd * 1e6:9.1f} | {t_dw * 1e6:9.1f} {fl / t_dw * 1e-12:6.1f}", flush=True)

print("decoder first convolutions (up-sampling loader): direct us / TF/s | Winograd us / alg TF/s")
for name, C0, C1, Cout, H in UP:
    low = torch.randn(N, H // 2, H // 2, C0, device="cuda")
    skip = torch.randn(N, H, H, C1, device="cuda")
    a0, b0 = torch.rand(N, C0, device="cuda") + 0.5, torch.randn(N, C0, device="cuda")
    a1, b1 = torch.rand(N, C1, device="cuda") + 0.5, torch.randn(N, C1, device="cuda")
    w = torch.randn(Cout, C0 + C1, 3, 3, device="cuda") * (2.0 / (9 * (C0 + C1))) ** 0.5
    b = torch.zeros(Cout, device="cuda")
    g1 = torch.ones(Cout, device="cuda"); bb = torch.zeros(Cout, device="cuda")
    wf, _ = ops.pack_conv3x3_weights(w, want_wd=False)
    uf, _ = ops.pack_wino_weights(w, want_d=False)
    sl, ss = ops.Act(low, a0, b0), ops.Act(skip, a1, b1)
    fl = 2.0 * N * H * H * 9 * (C0 + C1) * Cout
    t_d = timeit(lambda: ops.conv_up_in_fwd(sl, ss, 0.01, wf, b, g1, bb, 1e-5, None))
    t_w = timeit(lambda: ops.conv_up_in_fwd(sl, ss, 0.01, wf, b, g1, bb, 1e-5, None, wu=uf))
    print(f"{name:8s} {fl * 1e-9:7.1f} | {t_d * 1e6:9.1f} {fl / t_d * 1e-12:6.1f} | {t_w * 1e6:9.1f} "
          f"{fl / t_w * 1e-12:6.1f}", flush=True)
