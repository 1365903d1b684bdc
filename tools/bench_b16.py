#!/usr/bin/env python3
"""Per-layer times of the mixed-precision (bf16 tensors) kernels at bs 8, 512 x 512: fused
forward (conv_in_fwd b16), data gradient with the BSTATS epilogue, weight gradient.
Usage: [UNET_HIP_LIB=...] python tools/bench_b16.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 8
BF = torch.bfloat16
LAYERS = [("enc0.3", 32, 32, 512, 1), ("enc1.3", 64, 64, 256, 1), ("enc2.4", 128, 128, 128, 1),
          ("enc3.4", 256, 256, 64, 1), ("enc4.4", 512, 512, 32, 1), ("enc5.4", 512, 512, 16, 1),
          ("enc1.0", 32, 64, 512, 2), ("enc2.0", 64, 128, 256, 2), ("enc3.0", 128, 256, 128, 2),
          ("enc4.0", 256, 512, 64, 2), ("enc5.0", 512, 512, 32, 2)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'layer':8s} {'fwd us':>8s} {'dgrad+bs':>9s} {'wgrad us':>9s}")
tot = [0.0, 0.0, 0.0]
for name, Cin, Cout, H, s in LAYERS:
    Ho = H // s
    x = torch.randn(N, H, H, Cin, device="cuda").to(BF)
    al = torch.rand(N, Cin, device="cuda") + 0.5
    be = torch.randn(N, Cin, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * (2.0 / (9 * Cin)) ** 0.5
    b = torch.zeros(Cout, device="cuda")
    g1 = torch.ones(Cout, device="cuda"); b1 = torch.zeros(Cout, device="cuda")
    table = ops.PackTable([w], True, None)
    table.run()
    wf, wd, wf3, wd3 = table.wf[0], table.wd[0], table.wf3[0], table.wd3[0]
    src = ops.Act(x, al, be)
    t_f = timeit(lambda: ops.conv_in_fwd(src, None, 0.01, wf, b, 3, s, g1, b1, 1e-5, None, b16=True, w3=wf3))
    dy = torch.randn(N, Ho, Ho, Cout, device="cuda").to(BF)
    # the layer behind: raw output of the shape of x with its statistics
    st = torch.stack([torch.zeros(N, Cin, device="cuda"), torch.ones(N, Cin, device="cuda"),
                      torch.ones(N, Cin, device="cuda"), torch.zeros(N, Cin, device="cuda")])
    gx = torch.ones(Cin, device="cuda"); bx = torch.zeros(Cin, device="cuda")

    def dg():
        nn = ops.NextNorm(x, st, gx, bx, None, 0.01)
        return ops.conv3x3_bwd_data(dy, wd, 0, Cin, H, H, s, bf16="bf16", wd3=wd3, nxt=nn)
    t_d = timeit(dg)
    dw = torch.zeros(Cout, Cin, 3, 3, device="cuda")
    t_w = timeit(lambda: ops.conv_in_bwd_weight(src, 0.01, dy, dw, 0, 3, s))
    tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
    print(f"{name:8s} {t_f:8.1f} {t_d:9.1f} {t_w:9.1f}", flush=True)
print(f"{'total':8s} {tot[0]:8.1f} {tot[1]:9.1f} {tot[2]:9.1f}")
