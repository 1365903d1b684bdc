#!/usr/bin/env python3
"""Times of the RGB stem (3 -> 32 channels, bs 8, 512 x 512): fused forward and weight gradient, fp32
and bf16 layer tensors.  Usage: [UNET_STEM_WALK=0] python tools/bench_stem.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, H, C = 8, 512, 32


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x = torch.randn(N, H, H, 3, device="cuda")
w = torch.randn(C, 3, 3, 3, device="cuda") * 0.2
b = torch.zeros(C, device="cuda")
g1, b1 = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
wf = ops.pack_conv3x3_weights(w, want_wd=False)[0]
for b16 in (False, True):
    t_f = timeit(lambda: ops.conv_in_fwd(ops.Act(x), None, 0.01, wf, b, 3, 1, g1, b1, 1e-5, None, b16=b16))
    dy = torch.randn(N, H, H, C, device="cuda")
    if b16:
        dy = dy.to(torch.bfloat16)
    dw = torch.zeros(C, 3, 3, 3, device="cuda")
    t_w = timeit(lambda: ops.conv_in_bwd_weight(ops.Act(x), 0.01, dy, dw, 0, 3, 1))
    print(f"{'bf16' if b16 else 'fp32'}: stem forward {t_f:7.1f} us   weight gradient {t_w:7.1f} us", flush=True)
