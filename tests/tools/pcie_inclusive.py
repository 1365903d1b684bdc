import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import unet_implementations_amd as ua
from bench import synthetic_batch
dev = torch.device("cuda", 0)
model = ua.create_model(dev).train(); opt = ua.create_optimizer(model); lossf = ua.get_loss_function()
img, tgt = synthetic_batch(1, 8, 512, 512)
img_p, tgt_p = img.pin_memory(), tgt.pin_memory()
u8 = (torch.rand(8, 512, 512, 3) * 255).to(torch.uint8).pin_memory()
m8 = tgt.to(torch.uint8).pin_memory()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
d_img, d_tgt = img.to(dev), tgt.to(dev)
print("resident step ms", t(lambda: ua.train_step(model, opt, lossf, d_img, d_tgt)))
print("copy fp32 img + int64 mask (42 MB) ms", t(lambda: (img_p.to(dev, non_blocking=True), tgt_p.to(dev, non_blocking=True))))
print("step incl. H2D of fp32 batch ms", t(lambda: ua.train_step(model, opt, lossf, img_p.to(dev, non_blocking=True), tgt_p.to(dev, non_blocking=True))))
def u8step():
    x, tg = ua.ops.preprocess_u8(u8.to(dev, non_blocking=True), m8.to(dev, non_blocking=True))
    opt.zero_grad(); loss = lossf(model(x, input_layout="nhwc"), tg); loss.backward(); opt.step()
print("step incl. H2D of uint8 batch (8.4 MB) + preprocess kernel ms", t(u8step))
