#!/usr/bin/env python3
"""The 1/32-resolution layers (M = 2048 rows at bs 8): fused forward and data gradient, HIP events.
Environment knobs of the gather-GEMM dispatcher apply (UNET_IGEMM_KG, UNET_IGEMM_NGROUP).
Usage: python tools/bench_deep.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, C = 8, 512

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

out = []
for H, stride in ((16, 1), (32, 2)):
    x = torch.randn(N, H, H, C, device="cuda")
    al, be = torch.rand(N, C, device="cuda") + 0.5, torch.randn(N, C, device="cuda")
    w = torch.randn(C, C, 3, 3, device="cuda") * 0.02
    wf, wd = ops.pack_conv3x3_weights(w)
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    t = timeit(lambda: ops.conv_in_fwd(ops.Act(x, al, be), None, 0.01, wf, b, 3, stride, g, b, 1e-5, None))
    out.append(f"fwd{H}s{stride} {t:6.1f}")
    dy = torch.randn(N, 16, 16, C, device="cuda")
    dx = torch.empty(N, H, H, C, device="cuda")
    t = timeit(lambda: ops.conv3x3_bwd_data(dy, wd, 0, C, H, H, stride, out=dx))
    out.append(f"dgrad{H}s{stride} {t:6.1f}")
print(f"KG={os.environ.get('UNET_IGEMM_KG', '-')} NGROUP={os.environ.get('UNET_IGEMM_NGROUP', '-')}: " + "  ".join(out))
