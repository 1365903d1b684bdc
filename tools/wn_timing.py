import os, sys
sys.path.insert(0, "/root/repo")
import torch
import unet_implementations_amd as ua
ops = ua.ops
N = 8
for name, C, H in [("enc1.3", 64, 256), ("enc2.4", 128, 128), ("enc4.4", 512, 32)]:
    x = torch.randn(N, H, H, C, device="cuda")
    al = torch.rand(N, C, device="cuda") + 0.5
    be = torch.randn(N, C, device="cuda")
    w = torch.randn(C, C, 3, 3, device="cuda") * (2.0 / (9 * C)) ** 0.5
    b = torch.zeros(C, device="cuda")
    g1 = torch.ones(C, device="cuda"); b1 = torch.zeros(C, device="cuda")
    wf, wd = ops.pack_conv3x3_weights(w)
    uf, ud = ops.pack_wino_weights(w)
    src = ops.Act(x, al, be)
    for rep in range(3):
        y, st = ops.conv_in_fwd(src, None, 0.01, wf, b, 3, 1, g1, b1, 1e-5, None, wu=uf)
    torch.cuda.synchronize()
    nwg = N * (H // 8) * (H // 32) * (C // 64)
    t = y.view(-1).view(torch.int64)[: nwg * 4].view(nwg, 4).cpu().double()
    t0 = t[:, 0].min()
    pro = (t[:, 1] - t[:, 0]); loop = (t[:, 2] - t[:, 1]); epi = (t[:, 3] - t[:, 2])
    span = t[:, 3].max() - t0
    print(f"{name}: {nwg} WGs, chunks {C//8}: prologue {pro.mean():.0f} (min {pro.min():.0f} max {pro.max():.0f}), loop {loop.mean():.0f} "
          f"({loop.mean() / (C // 8):.0f}/chunk), epilogue {epi.mean():.0f} ticks; kernel span {span:.0f} ticks; "
          f"sum per WG {(pro + loop + epi).mean():.0f}; WGs per CU {nwg / 256:.0f}")
