#!/usr/bin/env python3
"""Host time to ENQUEUE one train step vs its GPU time (is the step launch-bound?).
Usage: python tools/cpu_overhead.py [fp32|bf16|bf16x3]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import unet_implementations_amd as ua
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
dev = torch.device("cuda")
model = ua.create_model(dev).train()
model.matmul_precision = mode
opt = ua.create_optimizer(model)
lossf = ua.get_loss_function()
x = torch.randn(8, 3, 512, 512, device=dev)
y = torch.randint(0, 3, (8, 512, 512), device=dev)
for _ in range(5):
    ua.train_step(model, opt, lossf, x, y)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    ua.train_step(model, opt, lossf, x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{mode}: enqueue {1e3 * (t1 - t0) / n:.2f} ms/step, total {1e3 * (t2 - t0) / n:.2f} ms/step")
