#!/usr/bin/env python3
"""Accuracy of the two fp32-class operand modes against fp64 (GPU + CPU oracle).

For the whole network (logits and the full gradient vector, oracle in fp64 on the CPU as truth)
and for single convolutions at the bench sizes (fp64 conv on the GPU as truth) this prints the
error of  (a) the oracle's own fp32 CPU run,  (b) the fp32 matrix-core kernels,  (c) the
split-bf16 ("bf16x3") kernels.  Output kept under profiles/ as evidence that bf16x3 is an
fp32-accurate mode.   Usage: python tests/tools/accuracy_vs_fp64.py [hw=64]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

torch.set_num_threads(16)
import unet_implementations_amd as ua
from oracle import unet_ref as O

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 2
sd0 = O.fill_state_dict(2024, trained_like=True)
img, tgt = O.synthetic_batch(1234, n, hw, hw)
masks = O.draw_dropout_masks(77, n)


def oracle(dtype):
    osd = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd0.items()}
    lg = O.unet_forward(osd, img.to(dtype), [m.to(dtype) for m in masks])
    w = O.class_weights(tgt).to(dtype)
    loss = F.cross_entropy(lg, tgt, weight=w, ignore_index=255) + O.dice_loss(lg, tgt)
    loss.backward()
    return lg.detach().double(), torch.cat([v.grad.reshape(-1).double() for v in osd.values()])


def hip(mode):
    model = ua.UNet()
    model.load_state_dict(sd0)
    model = model.to("cuda").train()
    model.matmul_precision = mode
    model.dropout_mask_override = masks
    lg = model(img.cuda())
    ua.SimpleLoss()(lg, tgt.cuda()).backward()
    return lg.detach().double().cpu(), torch.cat([p.grad.reshape(-1).double().cpu()
                                                  for p in model.parameters()])


l64, g64 = oracle(torch.float64)
print(f"whole network, {n} x {hw}x{hw}, trained-like weights, train mode; truth = fp64 oracle (CPU)")
print(f"{'path':34s} {'logits max-rel':>15s} {'gradient |d|/|g|':>17s}")
for name, (lg, g) in [("oracle fp32 (CPU, torch)", oracle(torch.float32)),
                      ("HIP fp32 matrix cores", hip("fp32")),
                      ("HIP bf16x3 (split bf16, 6 products)", hip("bf16x3")),
                      ("HIP bf16 (mixed precision)", hip("bf16"))]:
    el = ((lg - l64).abs().max() / l64.abs().max()).item()
    eg = ((g - g64).norm() / g64.norm()).item()
    print(f"{name:34s} {el:15.3e} {eg:17.3e}")

print()
print("single convolutions, bs 2, truth = fp64 conv (GPU); max-abs error / max|ref|")
print(f"{'layer':22s} {'fwd fp32':>10s} {'fwd x3':>10s} {'dgrad fp32':>11s} {'dgrad x3':>10s} "
      f"{'wgrad fp32':>11s} {'wgrad x3':>10s}")
g = torch.Generator(device="cuda").manual_seed(5)
for name, cin, cout, h in [("32->32 @128", 32, 32, 128), ("64->64 @128", 64, 64, 128),
                           ("128->128 @64", 128, 128, 64), ("256->256 @32", 256, 256, 32),
                           ("512->512 @32", 512, 512, 32), ("768->256 @32", 768, 256, 32)]:
    x = torch.randn(2, cin, h, h, device="cuda", generator=g)
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
    gy = torch.randn(2, cout, h, h, device="cuda", generator=g)
    xd, wd_ = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = F.conv2d(xd, wd_, padding=1)
    gx, gw = torch.autograd.grad(y, (xd, wd_), gy.double())
    xn = x.permute(0, 2, 3, 1).contiguous()
    gyn = gy.permute(0, 2, 3, 1).contiguous()
    wf, wdp = ua.ops.pack_conv3x3_weights(w)
    wf3, wd3 = ua.ops.pack_conv3x3_weights_bf16x3(w)
    b = torch.zeros(cout, device="cuda")

    def err(a, ref):
        return ((a.double() - ref).abs().max() / ref.abs().max()).item()

    row = []
    for mode, kw_f, kw_d in (("fp32", {}, {}), ("bf16x3", {"wf3": wf3}, {"wd3": wd3})):
        yf = ua.ops.conv3x3_fwd(xn, None, wf, b, 1, bf16=mode, **kw_f).permute(0, 3, 1, 2)
        dx = ua.ops.conv3x3_bwd_data(gyn, wdp, 0, cin, h, h, 1, bf16=mode, **kw_d).permute(0, 3, 1, 2)
        dw = torch.zeros_like(w)
        ua.ops.conv3x3_bwd_weight(xn, gyn, dw, 0, 1, bf16=mode)
        row.append((err(yf, y.detach()), err(dx, gx), err(dw, gw)))
    print(f"{name:22s} {row[0][0]:10.2e} {row[1][0]:10.2e} {row[0][1]:11.2e} {row[1][1]:10.2e} "
          f"{row[0][2]:11.2e} {row[1][2]:10.2e}")
