#!/usr/bin/env python3
"""Per-call kernel times of one train step, keyed by entry point and operand shape (HIP events
through ops.KernelTimer; the shapes are read from the calling ops function's locals).
Usage: python tools/layer_times.py [steps] [batch] [size]"""
import os, sys, inspect, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
ops = ua.ops

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 512
KEYS = ("N", "H", "W", "Ho", "Wo", "C0", "C1", "Cx", "Cin", "Cout", "stride", "ccols")


class DetailTimer(ops.KernelTimer):
    def end(self, tag, flops, launches, start, executed=None, nbytes=0.0):
        fr = inspect.currentframe().f_back
        loc = fr.f_locals
        shape = " ".join(f"{k}={loc[k]}" for k in KEYS if isinstance(loc.get(k), int))
        nxt = loc.get("nxt")
        key = f"{fr.f_code.co_name}[{shape}{' +bs' if nxt is not None else ''}]"
        super().end(key, flops, launches, start, executed, nbytes)


torch.manual_seed(0)
net = ua.create_model(torch.device("cuda")).train()
opt = ua.create_optimizer(net)
lossf = ua.get_loss_function()
x = torch.randn(N, 3, S, S, device="cuda")
y = torch.randint(0, 3, (N, S, S), device="cuda")


def step():
    return ua.train_step(net, opt, lossf, x, y)


for _ in range(2):
    step()
timer = DetailTimer()
ops.set_timer(timer)
for _ in range(steps):
    step()
ops.set_timer(None)
summ = timer.summary()
tot = sum(d["ms"] for d in summ.values()) / steps
print(f"{'call':95s} {'n':>3s} {'us/call':>8s} {'ms/step':>8s} {'alg TF/s':>8s} {'exe TF/s':>8s} {'GB/s':>7s}")
for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
    ms = d["ms"] / steps
    n = d["calls"] / steps
    s = d["ms"] * 1e-3
    print(f"{k[:95]:95s} {n:3.0f} {ms / n * 1e3:8.1f} {ms:8.3f} {d['flops'] / s * 1e-12:8.1f} "
          f"{d['executed'] / s * 1e-12:8.1f} {d['bytes'] / s * 1e-9:7.0f}")
print(f"{'TOTAL':95s} {'':3s} {'':8s} {tot:8.3f}")
