"""ORACLE — test infrastructure only (never imported by the product path).

CPU restatement of the reference's Our_UNet train step with stock torch fp32 ops:
the same ATen operators the reference reaches through `torch.nn`, written
functionally over a plain `state_dict`.  Each function cites the reference
file:line it follows (paths relative to the reference checkout).

Pinned: `tests/test_oracle_golden.py` checks this file against the fixtures in
`tests/golden/`, which `tests/tools/make_golden.py` generated in the build container by
importing the reference's own `Our_UNet/models/unet.py` and `losses.py`.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg use it.
"""
import numpy as np
import torch
import torch.nn.functional as F

FEATURES = [32, 64, 128, 256, 512, 512]          # Our_UNet/src/train.py:780
STRIDES = [1, 2, 2, 2, 2, 2]                      # Our_UNet/src/train.py:782
ENC_DROPOUT = [0.0, 0.0, 0.1, 0.2, 0.3, 0.3]      # Our_UNet/src/train.py:791
DEC_DROPOUT = [0.3, 0.2, 0.2, 0.1, 0.0]           # Our_UNet/src/train.py:793
NEG_SLOPE = 0.01                                  # nn.LeakyReLU default, models/unet.py:122-123
EPS = 1e-5                                        # models/unet.py:298-299


# --------------------------------------------------------------------------- naming
def block_indices(drop_rate):
    """Indices of (conv, norm) pairs inside ConvBlock.block: a SpatialDropout2d module is
    appended after each activation only when the rate is > 0 (models/unet.py:126-127)."""
    step = 4 if drop_rate > 0 else 3
    return [(0, 1), (step, step + 1)]


def layer_table():
    """Forward-order list of (prefix, conv_idx, norm_idx, cin, cout, stride, drop_rate, kind)."""
    rows = []
    cin = 3
    for e, (f, s, p) in enumerate(zip(FEATURES, STRIDES, ENC_DROPOUT)):
        for k, (ci, ni) in enumerate(block_indices(p)):
            rows.append((f"encoder_stages.{e}.block", ci, ni, cin if k == 0 else f, f,
                         s if k == 0 else 1, p, "enc"))
        cin = f
    for d, p in enumerate(DEC_DROPOUT):
        lvl = len(FEATURES) - 2 - d
        f = FEATURES[lvl]
        for k, (ci, ni) in enumerate(block_indices(p)):
            rows.append((f"decoder_stages.{d}.conv_block.block", ci, ni,
                         FEATURES[lvl + 1] + f if k == 0 else f, f, 1, p,
                         "dec_first" if k == 0 else "dec"))
    return rows


def dropout_layers():
    """(channels, rate) of every SpatialDropout2d in forward order."""
    return [(r[4], r[6]) for r in layer_table() if r[6] > 0]


# --------------------------------------------------------------------------- weights
def fill_state_dict(seed, trained_like=True, clip_dim=None):
    """Deterministic weights shared by the reference, the oracle and the HIP model.

    numpy PCG64 stream -> Kaiming-normal(fan_out, gain sqrt(2)) conv weights
    (models/unet.py:386-392).  With trained_like=True biases and the InstanceNorm affine
    parameters are perturbed so that every one of the 90 tensors influences the output.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}

    def conv(name, cout, cin, k):
        std = np.sqrt(2.0 / (cout * k * k))
        sd[name + ".weight"] = torch.from_numpy(
            (rng.standard_normal((cout, cin, k, k)) * std).astype(np.float32))
        b = rng.standard_normal(cout) * 0.1 if trained_like else np.zeros(cout)
        sd[name + ".bias"] = torch.from_numpy(b.astype(np.float32))

    def norm(name, c):
        g = 1.0 + 0.1 * rng.standard_normal(c) if trained_like else np.ones(c)
        b = 0.1 * rng.standard_normal(c) if trained_like else np.zeros(c)
        sd[name + ".weight"] = torch.from_numpy(g.astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy(b.astype(np.float32))

    for prefix, ci, ni, cin, cout, _, _, _ in layer_table():
        conv(f"{prefix}.{ci}", cout, cin, 3)
        norm(f"{prefix}.{ni}", cout)
    conv("segmentation_output", 3, FEATURES[0], 1)
    if clip_dim is None:
        return sd
    # CLIP_UNet variant: clip_fusion_conv sits between encoder and decoder in the state_dict
    # (CLIP_UNet/models/unet.py:356-362).  Its tensors come from a separate stream so the 90
    # base tensors stay identical to the plain network's.
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    base, sd = sd, {}
    f = FEATURES[-1]
    for k, v in base.items():
        if k.startswith("decoder_stages.0.") and "clip_fusion_conv.0.weight" not in sd:
            conv("clip_fusion_conv.0", f, f + clip_dim, 1)
            norm("clip_fusion_conv.1", f)
        sd[k] = v
    return sd


def synthetic_batch(seed, n, h, w):
    """Images ~ N(0,1) (ImageNet-standardised pixels, src/train.py:303-308) and masks with one
    foreground class per image, a 255 ring around the blob and background elsewhere."""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = torch.from_numpy(rng.standard_normal((n, 3, h, w)).astype(np.float32))
    yy, xx = np.mgrid[0:h, 0:w]
    mask = np.zeros((n, h, w), dtype=np.int64)
    for i in range(n):
        cy, cx = h * (0.4 + 0.2 * rng.random()), w * (0.4 + 0.2 * rng.random())
        ry, rx = h * (0.22 + 0.1 * rng.random()), w * (0.25 + 0.1 * rng.random())
        d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
        cls = 1 + (i % 2)
        mask[i][d < 1.0] = cls
        ring = 1.0 + 4.0 * 2.0 / min(ry, rx)
        mask[i][(d >= 1.0) & (d < ring)] = 255
    return img, torch.from_numpy(mask)


def draw_dropout_masks(seed, n):
    """Replays the reference's draws: one `x.new_empty(N,C,1,1).bernoulli_(1-p)` then
    `.div_(1-p)` per SpatialDropout2d in forward order (models/unet.py:30-31), from torch's
    global CPU generator seeded with `seed`."""
    torch.manual_seed(seed)
    out = []
    for c, p in dropout_layers():
        m = torch.empty(n, c, 1, 1).bernoulli_(1 - p)
        out.append(m.div_(1 - p).view(n, c))
    return out


# --------------------------------------------------------------------------- per-op
class _StoreBF16(torch.autograd.Function):
    """A tensor that lives in HBM as bf16: rounded where it is written, forward and backward
    (the mixed-precision pipeline stores activations and activation gradients as bf16)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _operand_bf16(w):
    """A matrix-core operand rounded on chip (weights): forward value rounded, gradient exact."""
    return w + (w.bfloat16().float() - w).detach()


def conv_in_lrelu_drop_bf16(x, w, b, gamma, beta, stride, mask=None, slope=NEG_SLOPE,
                            record=None):
    """The conv -> InstanceNorm -> LeakyReLU -> dropout unit with the rounding points of the
    MI355X mixed-precision pipeline (BASELINE config 4; the reference's own AMP path is fp16
    autocast, Our_UNet/src/train.py:638-652): operands of the convolution rounded to bf16, fp32
    accumulation, statistics from the fp32 result, the result stored as bf16, everything after
    it computed in fp32 from the stored value.  `x` is the previous unit's fp32 activation
    (computed on the fly from its stored bf16 tensor)."""
    y = F.conv2d(_operand_bf16(x) if not x.requires_grad else _StoreBF16.apply(x),
                 _operand_bf16(w), b, stride=stride, padding=w.shape[-1] // 2)
    mu = y.mean(dim=(2, 3), keepdim=True)
    var = y.var(dim=(2, 3), unbiased=False, keepdim=True)
    ys = _StoreBF16.apply(y)
    z = (ys - mu) * torch.rsqrt(var + EPS) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    a = F.leaky_relu(z, slope)
    if mask is not None:
        a = a * mask.view(mask.shape[0], mask.shape[1], 1, 1)
    if record is not None:      # debug: (raw output, activation) with their gradients retained
        if y.requires_grad:
            y.retain_grad()
            a.retain_grad()
        record.append((y, a))
    return a


def conv_in_lrelu_drop(x, w, b, gamma, beta, stride, mask=None, record=None, slope=NEG_SLOPE,
                       branch=None, tie_eps=0.0, tie_diag=None):
    """Conv2d(3x3, pad 1) -> InstanceNorm2d(eps, affine) -> LeakyReLU(0.01) -> channel mask
    (models/unet.py:101-134; SpatialDropout2d.forward :22-35).  `record` (debug): list that
    receives the raw conv output with retain_grad set.
    `branch` (test hook, tie-aware comparison): a bool tensor "z > 0" decided by ANOTHER fp32
    implementation of the same layer.  Where this run's own pre-activation is within `tie_eps`
    of zero - where two fp32 implementations may legitimately land on different sides - the
    LeakyReLU takes that decision instead of its own, so both runs differentiate the same
    piecewise-linear function; elsewhere the two must agree (counted in `tie_diag`)."""
    y = F.conv2d(x, w, b, stride=stride, padding=1)
    if record is not None:
        if y.requires_grad:
            y.retain_grad()
        record.append(y)
    y = F.instance_norm(y, weight=gamma, bias=beta, eps=EPS)
    if branch is not None:
        own = y.detach() > 0
        risky = y.detach().abs() < tie_eps
        if tie_diag is not None:
            tie_diag["risky"] = tie_diag.get("risky", 0) + int(risky.sum())
            tie_diag["taken_from_other"] = tie_diag.get("taken_from_other", 0) + \
                int((risky & (own != branch)).sum())
            tie_diag["disagree_away_from_ties"] = tie_diag.get("disagree_away_from_ties", 0) + \
                int((~risky & (own != branch)).sum())
        pos = torch.where(risky, branch, own)
        y = torch.where(pos, y, y * slope)
    else:
        y = F.leaky_relu(y, slope)   # `slope` = nonlin_kwargs["negative_slope"] of the reference ctor
    if mask is not None:
        y = y * mask.view(mask.shape[0], mask.shape[1], 1, 1)
    return y


def upsample_concat(x, skip):
    """UpBlock.forward up-sampling + concat, up-sampled tensor first (models/unet.py:215-228)."""
    if x.shape[2:] != skip.shape[2:]:
        x = F.interpolate(x, size=skip.shape[2:], mode="bilinear", align_corners=False)
    return torch.cat([x, skip], dim=1)


# --------------------------------------------------------------------------- network
def unet_forward(sd, x, masks=None, record=None, clip_features=None, slope=NEG_SLOPE,
                 bf16_storage=False, branches=None, tie_eps=0.0, tie_diag=None):
    """UNet.forward (models/unet.py:399-432).  `masks`: list from draw_dropout_masks (train
    mode) or None (eval / rates 0).  `record` (debug): collects each conv's raw output.
    `clip_features` [N,clip_dim,h/32,w/32]: the CLIP_UNet bottleneck fusion
    (CLIP_UNet/models/unet.py:441-478): cat -> 1x1 conv -> InstanceNorm -> LeakyReLU."""
    mi = iter(masks) if masks is not None else None
    skips = []
    rows = layer_table()
    cur = x
    n_enc = 2 * len(FEATURES)
    for li, (prefix, ci, ni, _, _, stride, p, kind) in enumerate(rows):
        if li == n_enc and clip_features is not None:
            cur = torch.cat([cur, clip_features], dim=1)
            cur = F.conv2d(cur, sd["clip_fusion_conv.0.weight"], sd["clip_fusion_conv.0.bias"])
            cur = F.instance_norm(cur, weight=sd["clip_fusion_conv.1.weight"],
                                  bias=sd["clip_fusion_conv.1.bias"], eps=EPS)
            cur = F.leaky_relu(cur, slope)
        if kind == "dec_first":
            cur = upsample_concat(cur, skips.pop())
        m = next(mi) if (mi is not None and p > 0) else None
        if bf16_storage:     # emulation of the MI355X mixed-precision pipeline's rounding points
            cur = conv_in_lrelu_drop_bf16(cur, sd[f"{prefix}.{ci}.weight"],
                                          sd[f"{prefix}.{ci}.bias"], sd[f"{prefix}.{ni}.weight"],
                                          sd[f"{prefix}.{ni}.bias"], stride, m, slope, record)
            if kind == "enc" and li % 2 == 1 and li < n_enc - 1:
                skips.append(cur)
            continue
        cur = conv_in_lrelu_drop(cur, sd[f"{prefix}.{ci}.weight"], sd[f"{prefix}.{ci}.bias"],
                                 sd[f"{prefix}.{ni}.weight"], sd[f"{prefix}.{ni}.bias"], stride, m,
                                 record, slope, None if branches is None else branches[li],
                                 tie_eps, tie_diag)
        if kind == "enc" and li % 2 == 1 and li < n_enc - 1:
            skips.append(cur)
    return F.conv2d(cur, sd["segmentation_output.weight"], sd["segmentation_output.bias"])


# --------------------------------------------------------------------------- loss
def class_weights(target, ignore_index=255, num_classes=3):
    """SimpleLoss._compute_class_weights (models/losses.py:24-62)."""
    valid = target != ignore_index
    total = valid.sum().float()
    counts = torch.stack([((target == c) & valid).sum().float() for c in range(num_classes)])
    counts = torch.where(counts == 0, torch.ones_like(counts), counts)
    w = total / counts
    return w * (num_classes / w.sum())


def dice_loss(logits, target, ignore_index=255, smooth=1e-5):
    """SimpleLoss._dice_loss (models/losses.py:84-121)."""
    valid = (target != ignore_index).float()
    prob = F.softmax(logits, dim=1)
    n, k = logits.shape[:2]
    total = 0
    for c in range(k):
        t = ((target == c).float() * valid).reshape(n, -1)
        p = (prob[:, c] * valid).reshape(n, -1)
        inter = (p * t).sum(dim=1)
        union = p.sum(dim=1) + t.sum(dim=1)
        total = total + (1.0 - ((2.0 * inter + smooth) / (union + smooth)).mean())
    return total / k


def simple_loss(logits, target, weight_dice=1.0, weight_ce=1.0, ignore_index=255, smooth=1e-5,
                dynamic_weights=True, fixed_weights=None):
    """SimpleLoss.forward (models/losses.py:64-82)."""
    w = class_weights(target, ignore_index) if dynamic_weights else fixed_weights
    ce = F.cross_entropy(logits, target, weight=w, ignore_index=ignore_index)
    return weight_ce * ce + weight_dice * dice_loss(logits, target, ignore_index, smooth)


# --------------------------------------------------------------------------- optimizer
def sgd_nesterov_(params, grads, bufs, lr=0.005, momentum=0.99, weight_decay=1e-4):
    """optim.SGD(nesterov=True, dampening=0) update (src/train.py:445-451); `bufs` entries are
    None before the first step."""
    with torch.no_grad():
        for i, (p, g) in enumerate(zip(params, grads)):
            g = g.add(p, alpha=weight_decay)
            if bufs[i] is None:
                bufs[i] = g.clone()
            else:
                bufs[i].mul_(momentum).add_(g)
            g = g.add(bufs[i], alpha=momentum)
            p.add_(g, alpha=-lr)


def train_step(sd, bufs, images, target, masks=None, lr=0.005, momentum=0.99, weight_decay=1e-4,
               clip_features=None, slope=NEG_SLOPE):
    """One step in the order of train_one_epoch (src/train.py:634-664).  `sd` values must be
    leaf tensors with requires_grad=True; returns (loss, {name: grad})."""
    names = list(sd.keys())
    for v in sd.values():
        v.grad = None
    logits = unet_forward(sd, images, masks, clip_features=clip_features, slope=slope)
    loss = simple_loss(logits, target)
    loss.backward()
    grads = {k: sd[k].grad.detach().clone() for k in names}
    sgd_nesterov_([sd[k] for k in names], [grads[k] for k in names], bufs, lr, momentum,
                  weight_decay)
    return loss.detach(), logits.detach(), grads


def leaf_state_dict(sd):
    return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


# ---------------------------------------------------------------- validation / input pipeline
def segmentation_metrics(preds, targets, num_classes=3, ignore_index=255):
    """Accumulators of SegmentationMetrics over a list of (pred, target) batches of class maps
    (numpy [B, H, W]) and the derived scores, restating Our_UNet/utils/metrics.py:59-91
    (`_update_single`: per-class intersection, union, TP / FP / FN over the valid pixels),
    :93-151 (pixel accuracy, IoU, Dice, nan for an empty denominator) and :121-169 (means over
    the non-nan classes).  PINNED: tests/golden/metrics.npz was recorded from the reference
    class itself (tests/tools/make_golden.py metrics_run)."""
    inter = np.zeros(num_classes)
    union = np.zeros(num_classes)
    tp, fp, fn = np.zeros(num_classes), np.zeros(num_classes), np.zeros(num_classes)
    total = correct = 0
    for pred_b, target_b in zip(preds, targets):
        for pred, target in zip(pred_b, target_b):
            mask = target != ignore_index
            total += mask.sum()
            correct += ((pred == target) & mask).sum()
            for c in range(num_classes):
                pc, tc = (pred == c) & mask, (target == c) & mask
                i = (pc & tc).sum()
                inter[c] += i
                union[c] += pc.sum() + tc.sum() - i
                tp[c] += i
                fp[c] += pc.sum() - i
                fn[c] += tc.sum() - i
    nan = float("nan")
    iou = [float(inter[c] / union[c]) if union[c] > 0 else nan for c in range(num_classes)]
    dice = [float(2 * tp[c] / (2 * tp[c] + fp[c] + fn[c])) if (2 * tp[c] + fp[c] + fn[c]) > 0
            else nan for c in range(num_classes)]

    def mean_valid(v):
        v = [x for x in v if not np.isnan(x)]
        return float(sum(v) / len(v)) if v else nan

    return dict(intersections=inter, unions=union, true_positives=tp, false_positives=fp,
                false_negatives=fn, total_pixels=int(total), correct_pixels=int(correct),
                pixel_accuracy=float(correct / total) if total > 0 else nan,
                iou=np.array(iou), dice=np.array(dice), mean_iou=mean_valid(iou),
                mean_dice=mean_valid(dice))


# PARITY UNPINNED for the three functions below: Our_UNet/src/train.py imports cv2, which is not
# installed here, so the reference's validate() / dataset code could not be executed to record
# fixtures.  They restate the arithmetic read from the source lines cited; the argmax/count part
# is integer-exact against torch.argmax in tests/test_kernels_gpu.py and - through
# `segmentation_metrics` above - pinned to the reference's SegmentationMetrics.
def batch_dice_scores(logits, masks, ignore_label=255):
    """Per-class Dice of one validation batch, as Our_UNet/src/train.py:556-577 computes it:
    argmax predictions, ignore pixels masked out, 2I/(|P|+|M|+1e-5), or 1.0 for an absent class."""
    preds = torch.argmax(logits, dim=1)
    valid = masks != ignore_label
    out = []
    for cls in range(3):
        pred_cls = ((preds == cls) & valid).float()
        mask_cls = ((masks == cls) & valid).float()
        inter = (pred_cls * mask_cls).sum().item()
        union = (pred_cls.sum() + mask_cls.sum()).item()
        out.append(2.0 * inter / (union + 1e-5) if union > 0 else 1.0)
    return out


def validate_scores(logit_batches, mask_batches, ignore_label=255):
    """Batch-averaged scores dictionary of validate() (src/train.py:579-589)."""
    per = [batch_dice_scores(l, m, ignore_label) for l, m in zip(logit_batches, mask_batches)]
    n = max(len(per), 1)
    d = [sum(p[c] for p in per) / n for c in range(3)]
    return {"background": d[0], "cat": d[1], "dog": d[2], "mean_foreground": (d[1] + d[2]) / 2.0}


def preprocess_sample(image_hwc_u8, mask_u8):
    """PetSegmentationDataset.__getitem__ arithmetic (Our_UNet/src/train.py:300-311) on numpy
    uint8 arrays: returns (CHW float32 image, int64 mask)."""
    import numpy as np
    mask = np.where((mask_u8 > 2) & (mask_u8 != 255), 0, mask_u8)
    image = torch.from_numpy(image_hwc_u8).float().permute(2, 0, 1) / 255.0
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    image = (image - mean) / std
    return image, torch.from_numpy(mask).long()
