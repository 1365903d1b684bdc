"""GPU check (not a test): degenerate inputs - all-zero, constant, huge and tiny images - through one
train step against the oracle (finite logits / gradients, zero-variance InstanceNorm planes)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import unet_implementations_amd as ua
from oracle import unet_ref as O
sd0 = O.fill_state_dict(5)
model = ua.UNet(); model.load_state_dict(sd0); model = model.cuda().train()
osd = O.leaf_state_dict(sd0)
for name, img in (("zeros", torch.zeros(2, 3, 64, 64)), ("const", torch.full((2, 3, 64, 64), 2.5)),
                  ("huge", torch.randn(2, 3, 64, 64) * 1e4), ("tiny", torch.randn(2, 3, 64, 64) * 1e-6)):
    _, tgt = O.synthetic_batch(1, 2, 64, 64)
    masks = O.draw_dropout_masks(3, 2)
    model.dropout_mask_override = masks
    model.zero_grad()
    lg = model(img.cuda())
    loss = ua.get_loss_function()(lg, tgt.cuda()); loss.backward()
    ol = O.unet_forward(osd, img, masks)
    oloss = O.simple_loss(ol, tgt)
    fin = bool(torch.isfinite(lg).all()) and all(bool(torch.isfinite(p.grad).all()) for p in model.parameters())
    err = ((lg.detach().cpu() - ol.detach()).abs().max() / (ol.detach().abs().max() + 1e-30)).item()
    print(f"{name}: finite={fin} logits rel err vs oracle {err:.2e} loss {loss.item():.6f} vs {oloss.item():.6f}")
