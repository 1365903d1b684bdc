import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torch
import unet_implementations_amd as ua
dev = torch.device("cuda")
model = ua.create_model(dev).train()
opt = ua.create_optimizer(model)
lossf = ua.get_loss_function()
x = torch.randn(8, 3, 512, 512, device=dev)
y = torch.randint(0, 3, (8, 512, 512), device=dev)
for _ in range(5):
    ua.train_step(model, opt, lossf, x, y)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    ua.train_step(model, opt, lossf, x, y)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
