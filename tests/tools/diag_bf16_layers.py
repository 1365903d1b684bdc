"""Diagnostic (GPU): layer by layer, how far the activation gradients of the HIP bf16 pipeline
(ga = dL/da_l entering the InstanceNorm backward, dy = dL/dy_l leaving it) are from the fp32
oracle, beside the same distances of the oracle's bf16 emulation.  Usage: python
tests/tools/diag_bf16_layers.py [hw]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import unet_implementations_amd as ua
from oracle import unet_ref as O

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sd0 = O.fill_state_dict(3)
img, tgt = O.synthetic_batch(1, 2, hw, hw)
masks = O.draw_dropout_masks(4, 2)


def oracle(emulate):
    osd = O.leaf_state_dict(sd0)
    rec = []
    if emulate:
        ol = O.unet_forward(osd, img, masks, bf16_storage=True, record=rec)
    else:
        ol = O.unet_forward(osd, img, masks, record=rec)
    O.simple_loss(ol, tgt).backward()
    grads = {k: v.grad for k, v in osd.items()}
    if emulate:
        return [(y.grad, a.grad) for y, a in rec], grads
    return [(y.grad, None) for y in rec], grads


o32, g32 = oracle(False)
oem, gem = oracle(True)


def hip(mode, fused=True):
    model = ua.UNet()
    model.load_state_dict(sd0)
    model = model.to("cuda").train()
    model.matmul_precision = mode
    model.fused_pipeline = fused
    model.dropout_mask_override = masks
    model._debug_capture = []
    logits = model(img.cuda())
    ua.get_loss_function()(logits, tgt.cuda()).backward()
    cap = {}
    for name, kind, t in model._debug_capture:
        cap[(name, kind)] = t.float().permute(0, 3, 1, 2).cpu()
    return cap, {k: p.grad.detach().cpu() for k, p in model.named_parameters()}


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).pow(2).sum().sqrt() / b.pow(2).sum().sqrt()).item()


hb, gb = hip("bf16")
hs, gs = hip("bf16", fused=False)
rows = O.layer_table()
names = []
for e in range(6):
    names += [f"encoder_stages.{e}.block.0", f"encoder_stages.{e}.block.1"]
for d in range(5):
    names += [f"decoder_stages.{d}.conv_block.block.0", f"decoder_stages.{d}.conv_block.block.1"]
print(f"{'layer':40s} | dy: emu   hip-b16  hip-std | dgamma: emu hip-b16 hip-std | dbeta: emu hip-b16 hip-std")
for li, (name, row) in enumerate(zip(names, rows)):
    prefix, ci, ni = row[0], row[1], row[2]
    ref_dy = o32[li][0]
    e_emu = rel(oem[li][0], ref_dy)
    e_b = rel(hb[(name, "dy")], ref_dy) if (name, "dy") in hb else float("nan")
    e_s = rel(hs[(name, "dy")], ref_dy) if (name, "dy") in hs else float("nan")
    kg, kb = f"{prefix}.{ni}.weight", f"{prefix}.{ni}.bias"
    print(f"{name:40s} | {e_emu:.3e} {e_b:.3e} {e_s:.3e} | {rel(gem[kg], g32[kg]):.3e} "
          f"{rel(gb[kg], g32[kg]):.3e} {rel(gs[kg], g32[kg]):.3e} | {rel(gem[kb], g32[kb]):.3e} "
          f"{rel(gb[kb], g32[kb]):.3e} {rel(gs[kb], g32[kb]):.3e}")
