"""How much of the gradient error vs fp64 is LeakyReLU tie-flip lottery?  N trials at 2 x 64x64:
whole-gradient relative error of (a) the oracle in fp32 on the CPU, (b) the HIP fp32 path,
(c) the HIP bf16x3 path, each against the oracle in fp64.  Usage: python tests/tools/flip_lottery.py [trials]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

torch.set_num_threads(16)
import unet_implementations_amd as ua
from oracle import unet_ref as O

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sd0 = O.fill_state_dict(2024)
rows = []
for t in range(trials):
    img, tgt = O.synthetic_batch(1234 + t, 2, 64, 64)
    masks = O.draw_dropout_masks(77 + t, 2)

    def run(dtype):
        osd = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd0.items()}
        lg = O.unet_forward(osd, img.to(dtype), [m.to(dtype) for m in masks])
        w = O.class_weights(tgt).to(dtype)
        loss = torch.nn.functional.cross_entropy(lg, tgt, weight=w, ignore_index=255) + O.dice_loss(lg, tgt)
        loss.backward()
        return lg.detach().double(), torch.cat([v.grad.reshape(-1).double() for v in osd.values()])

    def hip(mode):
        model = ua.UNet(); model.load_state_dict(sd0); model = model.to("cuda").train()
        model.matmul_precision = mode
        model.dropout_mask_override = masks
        lg = model(img.cuda()); ua.SimpleLoss()(lg, tgt.cuda()).backward()
        return lg.detach().double().cpu(), torch.cat([p.grad.reshape(-1).double().cpu() for p in model.parameters()])

    l64, g64 = run(torch.float64)
    row = []
    for lg, g in (run(torch.float32), hip("fp32"), hip("bf16x3")):
        row += [((lg - l64).abs().max() / l64.abs().max()).item(), ((g - g64).norm() / g64.norm()).item()]
    rows.append(row)
    print(f"trial {t}: logits ref32 {row[0]:.2e} hip {row[2]:.2e} x3 {row[4]:.2e} | grad ref32 {row[1]:.2e} hip {row[3]:.2e} x3 {row[5]:.2e}", flush=True)
import math
gm = lambda v: math.exp(sum(math.log(x) for x in v) / len(v))
for name, c in (("ref32", 1), ("hip fp32", 3), ("hip bf16x3", 5)):
    col = [r[c] for r in rows]
    print(f"{name:10s} gradient error: geometric mean {gm(col):.2e}  median {sorted(col)[len(col)//2]:.2e}  max {max(col):.2e}")
for name, c in (("ref32", 0), ("hip fp32", 2), ("hip bf16x3", 4)):
    col = [r[c] for r in rows]
    print(f"{name:10s} logits error:   geometric mean {gm(col):.2e}  max {max(col):.2e}")
