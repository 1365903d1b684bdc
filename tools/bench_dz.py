import os, sys
sys.path.insert(0, "/root/repo")
import torch
import unet_implementations_amd as ua
ops = ua.ops
N = 8
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, C, H in [("enc1.3", 64, 256), ("enc2.4", 128, 128), ("enc3.4", 256, 64), ("enc4.4", 512, 32)]:
    y = torch.randn(N, H, H, C, device="cuda")
    g = torch.randn(N, H, H, C, device="cuda")
    gamma = torch.ones(C, device="cuda"); beta = torch.zeros(C, device="cuda")
    st = ops.instnorm_stats(y, gamma, beta, 1e-5)
    w = torch.randn(C, C, 3, 3, device="cuda") * (2.0 / (9 * C)) ** 0.5
    wf, wd = ops.pack_conv3x3_weights(w)
    uf, ud = ops.pack_wino_weights(w)
    coef5 = torch.randn(5, N, C, device="cuda"); sums = torch.randn(N, C, 2, device="cuda")
    dg, db, dbi = (torch.empty(C, device="cuda") for _ in range(3))
    t_plain = timeit(lambda: ops.conv3x3_bwd_data(g, wd, 0, C, H, H, 1, ud=ud))
    t_dz = timeit(lambda: ops.conv3x3_bwd_data_dz(g, y, coef5, sums, gamma, st[1], 0.01, dg, db, dbi, ud, C, 0, C))
    print(f"{name}: plain dgrad {t_plain:.1f} us, dz dgrad {t_dz:.1f} us (+{(t_dz/t_plain-1)*100:.0f} %)")
