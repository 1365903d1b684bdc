"""Debug helper (GPU): gradient error of the HIP net and of the fp32 oracle, both measured
against an fp64 run of the oracle (is the HIP path as accurate as the reference's own fp32?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_num_threads(16)
import unet_implementations_amd as ua
from oracle import unet_ref as O

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 2
sd0 = O.fill_state_dict(2024)
img, tgt = O.synthetic_batch(1234, n, hw, hw)
masks = O.draw_dropout_masks(77, n)

def run_oracle(dtype):
    osd = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd0.items()}
    lg = O.unet_forward(osd, img.to(dtype), [m.to(dtype) for m in masks])
    # loss in the same dtype
    w = O.class_weights(tgt).to(dtype)
    ce = torch.nn.functional.cross_entropy(lg, tgt, weight=w, ignore_index=255)
    loss = ce + O.dice_loss(lg, tgt)
    loss.backward()
    return lg.detach(), loss.detach(), {k: v.grad for k, v in osd.items()}

l64, loss64, g64 = run_oracle(torch.float64)
l32, loss32, g32 = run_oracle(torch.float32)
model = ua.UNet(); model.load_state_dict(sd0); model = model.to("cuda").train()
model.dropout_mask_override = masks
logits = model(img.cuda())
loss = ua.SimpleLoss()(logits, tgt.cuda()); loss.backward()
den = l64.abs().max().item()
print(f"logits: hip-vs-64 {((logits.detach().cpu().double()-l64).abs().max().item()/den):.3e}   ref32-vs-64 {((l32.double()-l64).abs().max().item()/den):.3e}")
print(f"loss: 64 {loss64.item():.8f}  ref32 {loss32.item():.8f}  hip {loss.item():.8f}")
worst_h = worst_r = 0
for k, p in model.named_parameters():
    r = g64[k]; d = r.abs().max().item() + 1e-30
    eh = (p.grad.cpu().double() - r).abs().max().item() / d
    er = (g32[k].double() - r).abs().max().item() / d
    if d > 1e-4:
        worst_h = max(worst_h, eh); worst_r = max(worst_r, er)
    print(f"{k:55s} max|g64| {d:.3e}  hip {eh:.3e}  ref32 {er:.3e}  ratio {eh/(er+1e-30):.2f}")
print("worst (non-bias) hip", worst_h, "ref32", worst_r)
