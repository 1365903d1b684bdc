#!/usr/bin/env python3
"""The 32 -> 32 channel layers at 512 x 512: direct kernel vs the Winograd form of
csrc/conv_c32.hip (fused forward, data gradient with and without the BSTATS epilogue), times by
HIP events and the difference of the results.  Usage: python tools/bench_c32.py [reps] [N] [H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops
from unet_implementations_amd._lib import lib
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
C = 32


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


torch.manual_seed(0)
x = torch.randn(N, H, H, C, device="cuda")
al = torch.rand(N, C, device="cuda") + 0.5
be = torch.randn(N, C, device="cuda")
w = torch.randn(C, C, 3, 3, device="cuda") * (2.0 / (9 * C)) ** 0.5
b = torch.randn(C, device="cuda") * 0.1
g1 = torch.rand(C, device="cuda") + 0.5; b1 = torch.randn(C, device="cuda") * 0.1
wf, wd = ops.pack_conv3x3_weights(w)
src = ops.Act(x, al, be)
dy = torch.randn(N, H, H, C, device="cuda")
fl = 2.0 * N * H * H * 9 * C * C


def fwd():
    return ops.conv_in_fwd(src, None, 0.01, wf, b, 3, 1, g1, b1, 1e-5, None)


res = {}
for form in (0, 2):
    lib().unet_set_c32_winograd(form)
    y, st = fwd()
    if form == 0:
        y_ref, st_ref = y, st       # the BSTATS sums of both forms on the SAME y (lrelu' flips)
    t_f = timeit(fwd)
    def dg(bs):
        nn = ops.NextNorm(y_ref, st_ref, g1, b1, None, 0.01) if bs else None
        dx = ops.conv3x3_bwd_data(dy, wd, 0, C, H, H, 1, nxt=nn)
        return dx, (nn.partial.clone() if bs else None), (nn.tiles if bs else 0)
    dx0, _, _ = dg(False)
    dx1, part, tiles = dg(True)
    t_d = timeit(lambda: dg(False))
    t_b = timeit(lambda: dg(True))
    dw = torch.zeros(C, C, 3, 3, device="cuda")
    def wg():
        return ops.conv_in_bwd_weight(src, 0.01, dy, dw, 0, 3, 1)
    wg()
    t_w = timeit(wg)
    res[form] = (y.clone(), [s.clone() for s in st], dx0.clone(), dx1.clone(), part, tiles, dw.clone())
    print(f"          weight gradient {t_w * 1e6:7.1f} us {fl / t_w * 1e-12:6.1f} TF/s", flush=True)
    print(f"{'winograd' if form else 'direct  '}: fwd {t_f * 1e6:7.1f} us {fl / t_f * 1e-12:6.1f} TF/s | dgrad "
          f"{t_d * 1e6:7.1f} us {fl / t_d * 1e-12:6.1f} | dgrad+bs {t_b * 1e6:7.1f} us {fl / t_b * 1e-12:6.1f}", flush=True)
lib().unet_set_c32_winograd(1)  # back to the default (size rule)


def rel(a, bb):
    return ((a - bb).abs().max() / bb.abs().max()).item()


d, wv = res[0], res[2]
print("y rel", rel(wv[0], d[0]), "stats rel", [rel(a, bb) for a, bb in zip(wv[1], d[1])])
print("dx rel", rel(wv[2], d[2]), "dx(bs) rel", rel(wv[3], d[3]), "tiles", wv[5], d[5],
      "partial rel", rel(wv[4].view(torch.float32)[:wv[5] * N * C * 2], d[4].view(torch.float32)[:d[5] * N * C * 2]))
print("dw rel", rel(wv[6], d[6]))
