"""Debug helper (GPU): multi-step trajectory of the HIP path vs the oracle (fp32 and fp64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_num_threads(16)
import unet_implementations_amd as ua
from oracle import unet_ref as O

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 2
sd0 = O.fill_state_dict(2024)
img, tgt = O.synthetic_batch(1234, n, hw, hw)
model = ua.UNet(); model.load_state_dict(sd0); model = model.to("cuda").train()
opt = ua.create_optimizer(model); lossf = ua.get_loss_function()
model3 = ua.UNet(); model3.load_state_dict(sd0); model3 = model3.to("cuda").train()
model3.matmul_precision = "bf16x3"
opt3 = ua.create_optimizer(model3)
modelb = ua.UNet(); modelb.load_state_dict(sd0); modelb = modelb.to("cuda").train()
modelb.matmul_precision = "bf16"      # BASELINE config 4: bf16 layer tensors + bf16 matrix cores
optb = ua.create_optimizer(modelb)
o32 = O.leaf_state_dict(sd0); b32 = [None] * len(o32)
o64 = {k: v.double().clone().requires_grad_(True) for k, v in sd0.items()}; b64 = [None] * len(o64)

def step64(masks):
    for v in o64.values(): v.grad = None
    lg = O.unet_forward(o64, img.double(), [m.double() for m in masks])
    w = O.class_weights(tgt).double()
    loss = torch.nn.functional.cross_entropy(lg, tgt, weight=w, ignore_index=255) + O.dice_loss(lg, tgt)
    loss.backward()
    names = list(o64.keys())
    O.sgd_nesterov_([o64[k] for k in names], [o64[k].grad for k in names], b64)
    return loss.item()

for s in range(steps):
    masks = O.draw_dropout_masks(77 + s, n)
    l32, _, _ = O.train_step(o32, b32, img, tgt, masks)
    l64 = step64(masks)
    model.dropout_mask_override = masks
    lh = ua.train_step(model, opt, lossf, img.cuda(), tgt.cuda()).item()
    model3.dropout_mask_override = masks
    l3 = ua.train_step(model3, opt3, lossf, img.cuda(), tgt.cuda()).item()
    modelb.dropout_mask_override = masks
    lb = ua.train_step(modelb, optb, lossf, img.cuda(), tgt.cuda()).item()
    d3 = max(((p.detach().cpu().double() - o64[k].detach()).abs().max() / (o64[k].detach().abs().max() + 1e-30)).item() for k, p in model3.named_parameters())
    dh = max(((p.detach().cpu().double() - o64[k].detach()).abs().max() / (o64[k].detach().abs().max() + 1e-30)).item() for k, p in model.named_parameters())
    d32 = max(((o32[k].detach().double() - o64[k].detach()).abs().max() / (o64[k].detach().abs().max() + 1e-30)).item() for k in o32)
    print(f"step {s}: loss fp64 {l64:.6f}  ref32 {l32.item():.6f}  hip {lh:.6f}  hip-bf16x3 {l3:.6f}  hip-bf16 {lb:.6f} | "
          f"max rel param diff vs fp64: hip {dh:.3e}  hip-bf16x3 {d3:.3e}  ref32 {d32:.3e}")
