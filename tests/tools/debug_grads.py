"""Debug helper (GPU): per-parameter gradient error of the HIP net vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_num_threads(16)
import unet_implementations_amd as ua
from oracle import unet_ref as O

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 2
sd0 = O.fill_state_dict(2024)
model = ua.UNet(); model.load_state_dict(sd0); model = model.to("cuda").train()
img, tgt = O.synthetic_batch(1234, n, hw, hw)
masks = O.draw_dropout_masks(77, n)
osd = O.leaf_state_dict(sd0)
rec = []
ologits = O.unet_forward(osd, img, masks, rec)
oloss = O.simple_loss(ologits, tgt); oloss.backward()
model.dropout_mask_override = masks
model._debug_capture = []
logits = model(img.cuda())
loss = ua.SimpleLoss()(logits, tgt.cuda()); loss.backward()
print("logits rel err", ((logits.detach().cpu()-ologits.detach()).abs().max()/ologits.detach().abs().max()).item())
print("loss", loss.item(), oloss.item())
for k, p in model.named_parameters():
    ref = osd[k].grad
    e = (p.grad.cpu()-ref).abs().max().item()
    print(f"{k:55s} max|ref| {ref.abs().max().item():.3e}  abs err {e:.3e}  rel {e/(ref.abs().max().item()+1e-30):.3e}")

print("---- per-layer dy (grad wrt raw conv output) vs oracle, backward order")
dys = [(n, t) for n, kind, t in model._debug_capture if kind == "dy"]
for (name, t), y in zip(dys, reversed(rec)):
    ref = y.grad
    got = t.permute(0, 3, 1, 2).cpu()
    e = (got - ref).abs().max().item()
    print(f"{name:45s} shape {tuple(ref.shape)} max|ref| {ref.abs().max().item():.3e} abs err {e:.3e} rel {e/ref.abs().max().item():.3e}  mean-offset {((got-ref).mean().item()):.3e}")

print("---- detail for encoder_stages.2.block.0")
caps = {(n, k): t for n, k, t in model._debug_capture}
names = [n for n, k, t in model._debug_capture if k == "dy"]
li = names.index("encoder_stages.2.block.0")
yref = list(reversed(rec))[li]
got = caps[("encoder_stages.2.block.0", "dy")].permute(0, 3, 1, 2).cpu()
err = (got - yref.grad).abs()
thr = 1e-3 * yref.grad.abs().max()
bad = (err > thr)
print("bad elements:", bad.sum().item(), "of", bad.numel())
idx = bad.nonzero()
print("bad n:", idx[:, 0].unique().tolist())
print("bad c (count):", len(idx[:, 1].unique()), idx[:, 1].unique().tolist()[:40])
print("bad h:", idx[:, 2].unique().tolist())
print("bad w:", idx[:, 3].unique().tolist())
# per-channel error
pc = err.amax(dim=(2, 3))
print("per (n,c) max err, top:", torch.topk(pc.flatten(), 10))
m = masks[0]
print("mask values at bad channels:", [(int(n_), int(c_), m[n_, c_].item()) for n_, c_ in zip(*torch.nonzero(pc > thr, as_tuple=True))][:20])
