#!/usr/bin/env python3
"""Per-layer times of the low-resolution backward of the up-sampled operand on bf16 tensors
(bs 8, 512 x 512): D = upsample2x_bwd_taps(dy), then the data gradient g = sum_tap D_tap . wd[tap]
with the BSTATS epilogue, in the gather form (fp32 weights) and the plain-GEMM form (wd3).
Usage: [UNET_HIP_LIB=...] python tools/bench_lowres_b16.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops
FP32 = "--fp32" in sys.argv          # the same five layers on fp32 tensors (gather form only)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
reps = int(args[0]) if args else 20
N = 8
BF = torch.float32 if FP32 else torch.bfloat16
# (stage, channels of the low-resolution operand, Cout of the convolution, low resolution)
LAYERS = [("up1", 512, 512, 16), ("up2", 512, 256, 32), ("up3", 256, 128, 64),
          ("up4", 128, 64, 128), ("up5", 64, 32, 256)]


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'layer':6s} {'taps us':>8s} {'gather us':>10s} {'GEMM us':>8s} {'GB/s':>7s}")
tot = [0.0, 0.0, 0.0]
for name, C0, Cout, h in LAYERS:
    skip = Cout   # the skip half of the concatenated input (same width as Cout in Our_UNet)
    dy = torch.randn(N, 2 * h, 2 * h, Cout, device="cuda").to(BF)
    w = torch.randn(Cout, C0 + skip, 3, 3, device="cuda") * (2.0 / (9 * Cout)) ** 0.5
    table = ops.PackTable([w], not FP32, None)
    table.run()
    wd, wd3 = table.wd[0], table.wd3[0]
    t_t = timeit(lambda: ops.upsample2x_bwd_taps(dy))
    D = ops.upsample2x_bwd_taps(dy)
    yl = torch.randn(N, h, h, C0, device="cuda").to(BF)
    st = torch.stack([torch.zeros(N, C0, device="cuda"), torch.ones(N, C0, device="cuda"),
                      torch.ones(N, C0, device="cuda"), torch.zeros(N, C0, device="cuda")])
    gx = torch.ones(C0, device="cuda"); bx = torch.zeros(C0, device="cuda")
    out = torch.empty(N, h, h, C0, device="cuda", dtype=BF)

    def run(w3):
        nn = ops.NextNorm(yl, st, gx, bx, None, 0.01)
        return ops.conv3x3_up_bwd_data(D, wd, 0, C0, out=out, nxt=nn, wd3=w3)
    t_g = timeit(lambda: run(None))
    t_d = timeit(lambda: run(None if FP32 else wd3))
    nbytes = D.element_size() * (D.numel() + out.numel() + yl.numel())
    tot[0] += t_t; tot[1] += t_g; tot[2] += t_d
    print(f"{name:6s} {t_t:8.1f} {t_g:10.1f} {t_d:8.1f} {nbytes / t_d / 1e3:7.0f}", flush=True)
print(f"{'total':6s} {tot[0]:8.1f} {tot[1]:10.1f} {tot[2]:8.1f}")
