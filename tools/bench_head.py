#!/usr/bin/env python3
"""Times of the 1x1 head on the activated 32-channel tensor (bs 8, 512 x 512): forward and backward,
fp32 and bf16 layer tensors.  Usage: [UNET_HIP_LIB=...] python tools/bench_head.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, H, C, K = 8, 512, 32, 3


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


w = torch.randn(K, C, device="cuda") * 0.2
b = torch.zeros(K, device="cuda")
al = torch.rand(N, C, device="cuda") + 0.5
be = torch.randn(N, C, device="cuda")
dl = torch.randn(N, K, H, H, device="cuda")
for b16 in (False, True):
    y = torch.randn(N, H, H, C, device="cuda")
    if b16:
        y = y.to(torch.bfloat16)
    s = ops.Act(y, al, be)
    t_f = timeit(lambda: ops.head1x1_in_fwd(s, 0.01, w, b))
    dw, db = torch.empty(K, C, device="cuda"), torch.empty(K, device="cuda")
    t_b = timeit(lambda: ops.head1x1_in_bwd(s, 0.01, dl, w, dw, db))
    print(f"{'bf16' if b16 else 'fp32'}: head forward {t_f:7.1f} us   backward {t_b:7.1f} us", flush=True)
