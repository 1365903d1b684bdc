import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
HW = int(sys.argv[1]) if len(sys.argv) > 1 else 64
import unet_implementations_amd as ua
from oracle import unet_ref as O
for tl in (True, False):
    sd0 = O.fill_state_dict(2024, trained_like=tl)
    img, tgt = O.synthetic_batch(1234, 2, HW, HW)
    masks = O.draw_dropout_masks(77, 2)
    outs = {}
    for mode in ("fp32", "bf16"):
        model = ua.UNet(); model.load_state_dict(sd0); model = model.to("cuda").train()
        model.matmul_precision = mode; model.dropout_mask_override = masks
        logits = model(img.cuda()); loss = ua.SimpleLoss()(logits, tgt.cuda()); loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).double().cpu()
        outs[mode] = (logits.detach().cpu().double(), loss.item(), g)
    d = outs["bf16"][0] - outs["fp32"][0]
    ga, gb = outs["bf16"][2], outs["fp32"][2]
    print("trained_like", tl, "logits rms rel", (d.norm()/outs["fp32"][0].norm()).item(), "max rel", (d.abs().max()/outs["fp32"][0].abs().max()).item(),
          "loss", outs["bf16"][1], outs["fp32"][1], "grad rel", ((ga-gb).norm()/gb.norm()).item(), "cos", (ga@gb/(ga.norm()*gb.norm())).item())
