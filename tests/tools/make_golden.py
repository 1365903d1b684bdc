#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own code on CPU.

Runs only in the build container (needs /root/reference).  Imports the reference's
local files `Our_UNet/models/unet.py` and `Our_UNet/models/losses.py` (torch-only),
loads the deterministic weights of `oracle.unet_ref.fill_state_dict`, and records
small input/output vectors.  The fixtures are data only; nothing from the
reference's source travels.  Usage: python tests/tools/make_golden.py [--skip-512]
"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/Our_UNet")

from models.losses import SimpleLoss as RefLoss  # noqa: E402  (reference)
from models.unet import ConvBlock as RefConvBlock  # noqa: E402
from models.unet import UNet as RefUNet  # noqa: E402
from models.unet import UpBlock as RefUpBlock  # noqa: E402
from utils.metrics import SegmentationMetrics as RefMetrics  # noqa: E402  (numpy + torch only)

from oracle import unet_ref as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED_W, SEED_X, SEED_DROP = 2024, 1234, 77


def sample_idx(numel, k=64, seed=5):
    rng = np.random.Generator(np.random.PCG64(seed + numel))
    return np.sort(rng.choice(numel, size=min(k, numel), replace=False))


def npf(t):
    return t.detach().cpu().numpy()


def ref_model(sd):
    m = RefUNet()
    m.load_state_dict(sd)
    return m


def ops_small():
    """Per-op goldens from the reference's ConvBlock / UpBlock / SimpleLoss / torch SGD."""
    out = {}
    g = torch.Generator().manual_seed(11)
    # ConvBlock 32 -> 64, stride 2, dropout 0.2, train mode (masks replayed from seed)
    blk = RefConvBlock(32, 64, [3, 3], [2, 2], n_convs=2, spatial_dropout_rate=0.2)
    with torch.no_grad():
        for p in blk.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.1 if p.dim() > 1 else 0.3)
                    + (1.0 if p.dim() == 1 else 0.0) * 0.5)
    x = torch.randn(2, 32, 16, 16, generator=g, requires_grad=True)
    blk.train()
    torch.manual_seed(SEED_DROP)
    y = blk(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    torch.manual_seed(SEED_DROP)
    masks = [torch.empty(2, 64, 1, 1).bernoulli_(0.8).div_(0.8).view(2, 64) for _ in range(2)]
    out.update(cb_x=npf(x), cb_y=npf(y), cb_gy=npf(gy), cb_gx=npf(x.grad),
               cb_mask0=npf(masks[0]), cb_mask1=npf(masks[1]))
    for k, v in blk.state_dict().items():
        out["cb_p_" + k] = npf(v)
    for k, p in blk.named_parameters():
        out["cb_g_" + k] = npf(p.grad)
    # restated op agrees with the reference block (same ATen ops)
    sd = blk.state_dict()
    t = O.conv_in_lrelu_drop(x.detach(), sd["block.0.weight"], sd["block.0.bias"],
                             sd["block.1.weight"], sd["block.1.bias"], 2, masks[0])
    t = O.conv_in_lrelu_drop(t, sd["block.4.weight"], sd["block.4.bias"], sd["block.5.weight"],
                             sd["block.5.bias"], 1, masks[1])
    assert torch.equal(t, y.detach()), "oracle ConvBlock restatement != reference"

    # UpBlock 64(up) + 32(skip) -> 32, eval
    ub = RefUpBlock(64, 32, 32, [3, 3], n_convs=2, spatial_dropout_rate=0.0)
    with torch.no_grad():
        for p in ub.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.1 if p.dim() > 1 else 0.3)
                    + (1.0 if p.dim() == 1 else 0.0) * 0.5)
    xl = torch.randn(2, 64, 8, 12, generator=g, requires_grad=True)
    sk = torch.randn(2, 32, 16, 24, generator=g, requires_grad=True)
    yo = ub(xl, sk)
    gyo = torch.randn(yo.shape, generator=g)
    yo.backward(gyo)
    out.update(ub_x=npf(xl), ub_skip=npf(sk), ub_y=npf(yo), ub_gy=npf(gyo), ub_gx=npf(xl.grad),
               ub_gskip=npf(sk.grad))
    for k, v in ub.state_dict().items():
        out["ub_p_" + k] = npf(v)
    for k, p in ub.named_parameters():
        out["ub_g_" + k] = npf(p.grad)

    # SimpleLoss: one image without class 2, 255 border, one image with all classes
    lg = (torch.randn(2, 3, 24, 40, generator=g) * 2.0).requires_grad_(True)
    tg = torch.randint(0, 3, (2, 24, 40), generator=g)
    tg[0][tg[0] == 2] = 1
    tg[:, :2, :] = 255
    tg[:, :, -3:] = 255
    loss = RefLoss()(lg, tg)
    loss.backward()
    out.update(loss_logits=npf(lg), loss_target=tg.numpy(), loss_value=npf(loss),
               loss_dlogits=npf(lg.grad))
    lo = O.simple_loss(lg.detach(), tg)
    assert torch.equal(lo, loss.detach()), "oracle loss restatement != reference"
    # static weights variant
    lg2 = lg.detach().clone().requires_grad_(True)
    cw = torch.tensor([0.5, 1.5, 1.0])
    loss2 = RefLoss(class_weights=cw, dynamic_weights=False)(lg2, tg)
    loss2.backward()
    out.update(loss2_weights=npf(cw), loss2_value=npf(loss2), loss2_dlogits=npf(lg2.grad))

    # torch.optim.SGD with the reference's settings, 3 steps on a fixed gradient sequence
    p = torch.randn(1000, generator=g)
    p0 = p.clone()
    param = torch.nn.Parameter(p)
    opt = torch.optim.SGD([param], lr=0.005, momentum=0.99, nesterov=True, weight_decay=1e-4)
    grads = [torch.randn(1000, generator=g) for _ in range(3)]
    traj = []
    for gr in grads:
        param.grad = gr.clone()
        opt.step()
        traj.append(param.detach().clone())
    out.update(sgd_p0=npf(p0), sgd_grads=np.stack([npf(t) for t in grads]),
               sgd_traj=np.stack([npf(t) for t in traj]))
    np.savez_compressed(os.path.join(OUT, "ops_small.npz"), **out)
    print("ops_small.npz", sum(v.nbytes for v in out.values()) // 1024, "KiB raw")


def net_run(hw, n, steps, tag, full_logits):
    sd0 = O.fill_state_dict(SEED_W)
    img, tgt = O.synthetic_batch(SEED_X, n, hw, hw)
    out = dict(seed_w=SEED_W, seed_x=SEED_X, seed_drop=SEED_DROP, n=n, hw=hw)
    model = ref_model(sd0)
    # --- eval forward
    model.eval()
    with torch.no_grad():
        le = model(img)
        lo = O.unet_forward(sd0, img)
    assert torch.equal(le, lo), "oracle eval forward != reference"
    if full_logits:
        out["eval_logits"] = npf(le)
    else:
        out["eval_logits_s16"] = npf(le[:, :, ::16, ::16])
    am = le.argmax(dim=1).to(torch.uint8).numpy()
    top2 = le.topk(2, dim=1).values
    margin = npf(top2[:, 0] - top2[:, 1])
    out["eval_argmax"] = np.packbits(
        np.stack([(am >> 1) & 1, am & 1], axis=-1).astype(np.uint8).reshape(-1))
    out["eval_argmax_sha256"] = hashlib.sha256(am.tobytes()).hexdigest()
    out["eval_lowmargin"] = np.packbits((margin < 1e-3).reshape(-1))
    # --- train steps with the reference's optimizer and loss
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=0.005, momentum=0.99, nesterov=True,
                          weight_decay=1e-4)
    lossf = RefLoss()
    names = [k for k, _ in model.named_parameters()]
    osd = O.leaf_state_dict(sd0)
    obufs = [None] * len(names)
    for s in range(steps):
        torch.manual_seed(SEED_DROP + s)
        opt.zero_grad()
        logits = model(img)
        loss = lossf(logits, tgt)
        loss.backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        opt.step()
        # oracle replay with replayed masks
        masks = O.draw_dropout_masks(SEED_DROP + s, n)
        oloss, ologits, ograds = O.train_step(osd, obufs, img, tgt, masks)
        assert torch.equal(ologits, logits.detach()), f"oracle train logits != reference (step {s})"
        assert torch.equal(oloss, loss.detach()), f"oracle loss != reference (step {s})"
        for k in names:
            assert torch.allclose(ograds[k], grads[k], rtol=0, atol=0), f"grad {k} differs"
        out[f"loss_{s}"] = npf(loss)
        if s == 0:
            if full_logits:
                out["train_logits"] = npf(logits)
            else:
                out["train_logits_s16"] = npf(logits[:, :, ::16, ::16])
            for i, k in enumerate(names):
                gk = grads[k].reshape(-1)
                out[f"gnorm_{i}"] = np.float64(gk.double().norm().item())
                out[f"gsamp_{i}"] = npf(gk[torch.from_numpy(sample_idx(gk.numel()))])
        for i, (k, p) in enumerate(model.named_parameters()):
            d = (p.detach() - sd0[k]).reshape(-1)
            out[f"dnorm_{s}_{i}"] = np.float64(d.double().norm().item())
            if s == steps - 1:
                out[f"psamp_{i}"] = npf(p.detach().reshape(-1)[
                    torch.from_numpy(sample_idx(d.numel()))])
        print(f"  {tag} step {s}: loss {loss.item():.6f}")
    for k in names:
        assert torch.equal(osd[k].detach(), dict(model.named_parameters())[k].detach()), \
            f"oracle SGD trajectory differs on {k}"
    out["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, f"{tag}.npz"), **out)
    print(f"{tag}.npz written")


def clip_run():
    """CLIP_UNet variant (reference: CLIP_UNet/models/unet.py, torch-only file) with synthetic
    CLIP features: eval logits, one train step's loss and gradient norms/samples."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("clip_ref_unet",
                                                  "/root/reference/CLIP_UNet/models/unet.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n, hw, clip_dim = 2, 64, 512
    sd0 = O.fill_state_dict(SEED_W, clip_dim=clip_dim)
    model = mod.UNet(with_clip_features=True, clip_dim=clip_dim)
    model.load_state_dict(sd0)
    img, tgt = O.synthetic_batch(SEED_X, n, hw, hw)
    gen = torch.Generator().manual_seed(SEED_X + 1)
    clip = torch.randn(n, clip_dim, hw // 32, hw // 32, generator=gen)
    out = dict(seed_w=SEED_W, seed_x=SEED_X, seed_drop=SEED_DROP, n=n, hw=hw, clip_dim=clip_dim,
               clip_features=npf(clip))
    model.eval()
    with torch.no_grad():
        le = model(img, clip)
        lo = O.unet_forward(sd0, img, clip_features=clip)
    assert torch.equal(le, lo), "oracle CLIP eval forward != reference"
    out["eval_logits"] = npf(le)
    model.train()
    torch.manual_seed(SEED_DROP)
    logits = model(img, clip)
    loss = RefLoss()(logits, tgt)
    loss.backward()
    masks = O.draw_dropout_masks(SEED_DROP, n)
    osd = O.leaf_state_dict(sd0)
    ol = O.unet_forward(osd, img, masks, clip_features=clip)
    assert torch.equal(ol.detach(), logits.detach()), "oracle CLIP train forward != reference"
    out["train_logits"] = npf(logits)
    out["loss_0"] = npf(loss)
    names = [k for k, _ in model.named_parameters()]
    for i, (k, p) in enumerate(model.named_parameters()):
        gk = p.grad.reshape(-1)
        out[f"gnorm_{i}"] = np.float64(gk.double().norm().item())
        out[f"gsamp_{i}"] = npf(gk[torch.from_numpy(sample_idx(gk.numel()))])
    out["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "clip64.npz"), **out)
    print("clip64.npz written, loss", loss.item())


def slope1_run():
    """The reference UNet built with nonlin_kwargs = {negative_slope: 1.0} (LeakyReLU becomes the
    identity): a network without activation ties, whose gradients every fp32 implementation
    reproduces element by element.  One train-mode step at 64x64: logits, loss, gradient norms
    and 256 sampled entries per tensor (tests hold them to 1e-4, not to the tie-flip noise)."""
    n, hw = 2, 64
    sd0 = O.fill_state_dict(SEED_W)
    img, tgt = O.synthetic_batch(SEED_X, n, hw, hw)
    model = RefUNet(nonlin_kwargs={"negative_slope": 1.0, "inplace": True})
    model.load_state_dict(sd0)
    model.train()
    torch.manual_seed(SEED_DROP)
    logits = model(img)
    loss = RefLoss()(logits, tgt)
    loss.backward()
    masks = O.draw_dropout_masks(SEED_DROP, n)
    osd = O.leaf_state_dict(sd0)
    oloss, ologits, ograds = O.train_step(osd, [None] * len(osd), img, tgt, masks, slope=1.0)
    assert torch.equal(ologits, logits.detach()), "oracle slope-1 logits != reference"
    assert torch.equal(oloss, loss.detach()), "oracle slope-1 loss != reference"
    out = dict(seed_w=SEED_W, seed_x=SEED_X, seed_drop=SEED_DROP, n=n, hw=hw,
               train_logits=npf(logits), loss_0=npf(loss))
    names = [k for k, _ in model.named_parameters()]
    for i, (k, p) in enumerate(model.named_parameters()):
        assert torch.equal(ograds[k], p.grad), f"oracle slope-1 grad {k} != reference"
        gk = p.grad.reshape(-1)
        out[f"gnorm_{i}"] = np.float64(gk.double().norm().item())
        out[f"gsamp_{i}"] = npf(gk[torch.from_numpy(sample_idx(gk.numel(), k=256))])
    out["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "net64_slope1.npz"), **out)
    print("net64_slope1.npz written, loss", loss.item())


def metrics_run():
    """SegmentationMetrics (Our_UNet/utils/metrics.py:7-235) on argmax predictions of synthetic
    logits: two batches with exact logit ties, a class that never occurs in the targets, a class
    that is never predicted, an all-ignored image and a 255 ring.  Records the logits / targets
    and the reference's accumulators and derived scores, per case and accumulated."""
    out = {}
    g = torch.Generator().manual_seed(31)
    h, w = 24, 40
    cases = []
    # batch 0: random logits with ties; targets {0, 1, 255 ring}: class 2 absent from the labels
    lg0 = torch.randn(3, 3, h, w, generator=g)
    lg0[:, :, : h // 4] = lg0[:, :1, : h // 4]                    # three-way ties -> class 0
    lg0[:, 2, h // 4: h // 2] = lg0[:, 1, h // 4: h // 2]         # 1/2 ties -> first maximum
    t0 = torch.randint(0, 2, (3, h, w), generator=g)
    t0[:, 5:9, 10:30] = 255
    t0[1] = 255                                                   # an all-ignored image
    cases.append((lg0, t0))
    # batch 1: class 1 never predicted (its logit pushed down); all three classes labelled
    lg1 = torch.randn(2, 3, h, w, generator=g)
    lg1[:, 1] -= 100.0
    t1 = torch.randint(0, 3, (2, h, w), generator=g)
    t1[torch.rand(2, h, w, generator=g) < 0.1] = 255
    cases.append((lg1, t1))
    acc = RefMetrics(num_classes=3, ignore_index=255)
    fields = ("intersections", "unions", "true_positives", "false_positives", "false_negatives")

    def record(tag, m):
        for f in fields:
            out[f"{tag}_{f}"] = np.asarray(getattr(m, f), dtype=np.float64)
        out[f"{tag}_total_pixels"] = np.int64(m.total_pixels)
        out[f"{tag}_correct_pixels"] = np.int64(m.correct_pixels)
        out[f"{tag}_pixel_accuracy"] = np.float64(m.compute_pixel_accuracy())
        out[f"{tag}_iou"] = np.array([m.compute_iou(c) for c in range(3)])
        out[f"{tag}_dice"] = np.array([m.compute_dice(c) for c in range(3)])
        out[f"{tag}_precision"] = np.array([m.compute_precision(c) for c in range(3)])
        out[f"{tag}_recall"] = np.array([m.compute_recall(c) for c in range(3)])
        out[f"{tag}_mean_iou"] = np.float64(m.compute_mean_iou())
        out[f"{tag}_mean_dice"] = np.float64(m.compute_mean_dice())

    for k, (lg, t) in enumerate(cases):
        pred = lg.argmax(dim=1)                  # what evaluate / validate feed the class
        one = RefMetrics(num_classes=3, ignore_index=255)
        one.update(pred, t)
        acc.update(pred, t)
        out[f"b{k}_logits"], out[f"b{k}_target"] = npf(lg), t.numpy().astype(np.int64)
        out[f"b{k}_pred"] = pred.numpy().astype(np.uint8)
        record(f"b{k}", one)
        # the restatement agrees with the reference class
        r = O.segmentation_metrics([pred.numpy()], [t.numpy()])
        for f in fields:
            assert np.array_equal(r[f], out[f"b{k}_{f}"]), f
        assert np.array_equal(r["iou"], out[f"b{k}_iou"], equal_nan=True)
        assert np.array_equal(r["dice"], out[f"b{k}_dice"], equal_nan=True)
    record("acc", acc)
    out["n_batches"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
    print("metrics.npz written: accumulated IoU", out["acc_iou"], "Dice", out["acc_dice"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only-slope1", action="store_true")
    ap.add_argument("--skip-512", action="store_true")
    ap.add_argument("--only-clip", action="store_true")
    ap.add_argument("--only-metrics", action="store_true")
    args = ap.parse_args()
    if args.only_metrics:
        os.makedirs(OUT, exist_ok=True)
        metrics_run()
        return
    if args.only_clip:
        os.makedirs(OUT, exist_ok=True)
        clip_run()
        return
    if args.only_slope1:
        slope1_run()
        return
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    ops_small()
    net_run(64, 2, 3, "net64", full_logits=True)
    if not args.skip_512:
        net_run(512, 2, 3, "net512", full_logits=False)
    clip_run()
    slope1_run()
    metrics_run()


if __name__ == "__main__":
    main()
