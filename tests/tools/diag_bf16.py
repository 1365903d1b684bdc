"""Diagnostic: distance of the bf16 pipelines and of the oracle's bf16 emulation from the fp32 oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import unet_implementations_amd as ua
from oracle import unet_ref as O
from tests.test_bf16_gpu import _hip_run, _rms

sd0 = O.fill_state_dict(3)
img, tgt = O.synthetic_batch(1, 2, 64, 64)
masks = O.draw_dropout_masks(4, 2)
def oracle(emulate):
    osd = O.leaf_state_dict(sd0)
    ol = O.unet_forward(osd, img, masks, bf16_storage=emulate)
    oloss = O.simple_loss(ol, tgt); oloss.backward()
    return ol.detach(), oloss.item(), {k: v.grad for k, v in osd.items()}
l32, loss32, g32 = oracle(False)
lem, lossem, gem = oracle(True)
runs = {"hip b16 fused": _hip_run(ua, sd0, img, tgt, masks, "bf16"),
        "hip bf16 standalone (fp32 storage)": _hip_run(ua, sd0, img, tgt, masks, "bf16", fused=False),
        "hip fp32": _hip_run(ua, sd0, img, tgt, masks, "fp32")}
print("logits rms vs fp32 oracle: emulation %.3e" % _rms(lem, l32), {k: "%.3e" % _rms(v[0], l32) for k, v in runs.items()})
print("loss: fp32 %.5f emu %.5f" % (loss32, lossem), {k: "%.5f" % v[1] for k, v in runs.items()})
flat = lambda d: torch.cat([d[k].double().reshape(-1) for k in g32])
cosf = lambda a, b: F.cosine_similarity(a.reshape(1, -1), b.reshape(1, -1)).item()
print("whole-gradient 1-cos vs fp32: emu %.3e" % (1 - cosf(flat(gem), flat(g32))), {k: "%.3e" % (1 - cosf(flat(v[2]), flat(g32))) for k, v in runs.items()})
for k in g32:
    if g32[k].abs().max() < 1e-4: continue
    ref = g32[k].double()
    print("%-50s n=%8d emu %.2e" % (k, ref.numel(), 1 - cosf(gem[k].double(), ref)), " ".join("%.2e" % (1 - cosf(v[2][k].double(), ref)) for v in runs.values()))
