"""Diagnostic: CLIP golden gradients, fused vs stand-alone pipeline, per parameter."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import unet_implementations_amd as ua
from oracle import unet_ref as O
from tests.test_net_gpu import sample_idx

g = np.load("tests/golden/clip64.npz", allow_pickle=False)
n, hw, clip_dim = int(g["n"]), int(g["hw"]), int(g["clip_dim"])
sd0 = O.fill_state_dict(int(g["seed_w"]), clip_dim=clip_dim)
img, tgt = O.synthetic_batch(int(g["seed_x"]), n, hw, hw)
clip = torch.from_numpy(g["clip_features"]).cuda()
res = {}
for fused in (True, False):
    model = ua.CLIPUNet(with_clip_features=True, clip_dim=clip_dim)
    model.load_state_dict(sd0); model = model.cuda(); model.fused_pipeline = fused
    model.train()
    model.dropout_mask_override = O.draw_dropout_masks(int(g["seed_drop"]), n)
    logits = model(img.cuda(), clip)
    loss = ua.SimpleLoss()(logits, tgt.cuda()); loss.backward()
    res[fused] = [p.grad.detach().clone().reshape(-1) for p in model.parameters()]
names = [k for k, _ in model.named_parameters()]
for i, k in enumerate(names):
    ref_norm = float(g[f"gnorm_{i}"])
    if ref_norm < 1e-4: continue
    idx = torch.from_numpy(sample_idx(res[True][i].numel())).cuda()
    ref_s = torch.from_numpy(g[f"gsamp_{i}"])
    tol = 1e-2 * max(ref_s.abs().max().item(), ref_norm / res[True][i].numel() ** 0.5)
    ef = (res[True][i][idx].cpu() - ref_s).abs(); eu = (res[False][i][idx].cpu() - ref_s).abs()
    d = (res[True][i] - res[False][i]).abs().max().item()
    flag = "  <<<" if (ef > tol).sum() > 3 or (eu > tol).sum() > 3 else ""
    print(f"{k:55s} tol {tol:.2e} fused max {ef.max():.2e} n>{int((ef>tol).sum())}  unfused max {eu.max():.2e} n>{int((eu>tol).sum())}  |f-u| {d:.2e}{flag}")
