import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import unet_implementations_amd as ua
from oracle import unet_ref as O
DEV="cuda"
N, hw = 8, 512
sd0 = O.fill_state_dict(77, trained_like=True)
img, _ = O.synthetic_batch(4321, N, hw, hw); img = img.to(DEV)
g = torch.Generator(device="cpu").manual_seed(5)
dlogits = (torch.randn(N, 3, hw, hw, generator=g) * 1e-3).to(DEV)
masks = O.draw_dropout_masks(91, N)
def run(sl, prec="fp32"):
    model = ua.UNet(); model.load_state_dict(sd0); model = model.to(DEV).train()
    model.matmul_precision = prec
    model.dropout_mask_override = [m[sl] for m in masks]
    out = model(img[sl]); out.backward(dlogits[sl])
    return model, out.detach()
mf, lf = run(slice(0, N))
parts = None
for i in range(0, N, 2):
    m, l = run(slice(i, i+2))
    gs = [p.grad.detach().clone() for p in m.parameters()]
    parts = gs if parts is None else [a+b for a,b in zip(parts, gs)]
names=[k for k,_ in mf.named_parameters()]
for k,p,q in zip(names, mf.parameters(), parts):
    e=((p.grad-q).norm()/(q.norm()+1e-30)).item()
    if k.endswith("weight") and p.dim()==4: print(f"{k:50s} {e:.2e}  |g|={q.norm().item():.3e}")
# same comparison bs8 vs bs8 with patch kernels off? (env must be set before first launch) -> second process
