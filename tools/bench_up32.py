#!/usr/bin/env python3
"""conv3x3(cat(upsample2x(act(low)), act(skip))) with (64 + 32) -> 32 channels: the direct patch
kernel vs the Winograd form of csrc/conv_c32.hip (conv_wino_up32_kernel).
Usage: python tools/bench_up32.py [reps] [N] [H] [W]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
W = int(sys.argv[4]) if len(sys.argv) > 4 else H
C0, C1, Cout = 64, 32, 32


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


torch.manual_seed(0)
low = torch.randn(N, H // 2, W // 2, C0, device="cuda")
skip = torch.randn(N, H, W, C1, device="cuda")
a0, b0 = torch.rand(N, C0, device="cuda") + 0.5, torch.randn(N, C0, device="cuda")
a1, b1 = torch.rand(N, C1, device="cuda") + 0.5, torch.randn(N, C1, device="cuda")
w = torch.randn(Cout, C0 + C1, 3, 3, device="cuda") * (2.0 / (9 * (C0 + C1))) ** 0.5
b = torch.randn(Cout, device="cuda") * 0.1
g1 = torch.rand(Cout, device="cuda") + 0.5; bb = torch.randn(Cout, device="cuda") * 0.1
wf, _ = ops.pack_conv3x3_weights(w, want_wd=False)
sl, ss = ops.Act(low, a0, b0), ops.Act(skip, a1, b1)
fl = 2.0 * N * H * W * 9 * (C0 + C1) * Cout
res = {}
for form in (False, "always"):
    ops.set_c32_winograd(form)
    y, st = ops.conv_up_in_fwd(sl, ss, 0.01, wf, b, g1, bb, 1e-5, None)
    t = timeit(lambda: ops.conv_up_in_fwd(sl, ss, 0.01, wf, b, g1, bb, 1e-5, None))
    res[form] = (y.clone(), [s.clone() for s in st])
    print(f"{'winograd' if form else 'direct  '}: {t * 1e6:8.1f} us {fl / t * 1e-12:6.1f} TF/s", flush=True)
ops.set_c32_winograd(True)
rel = lambda a, bb: ((a - bb).abs().max() / bb.abs().max()).item()
print("y rel", rel(res["always"][0], res[False][0]), "stats rel", [rel(a, c) for a, c in zip(res["always"][1], res[False][1])])
