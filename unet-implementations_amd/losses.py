"""Drop-in `SimpleLoss` (reference: Our_UNet/models/losses.py:5-121).

Dice + cross-entropy with per-batch inverse-frequency class weights and
ignore_index handling.  `forward(logits[N,3,H,W] fp32, target[N,H,W] int64)`
returns a 0-dim fp32 tensor supporting `.backward()` / `.item()`.  One fused
kernel sequence computes the loss AND dL/dlogits in the forward call; autograd's
backward only scales that stored gradient.
"""
import torch
import torch.nn as nn

from . import ops


class _LossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, mod, class_weights):
        want_grad = ctx.needs_input_grad[0]
        out, dl = ops.dice_wce_loss_fwd_bwd(
            logits, target, mod.smooth, mod.weight_dice, mod.weight_ce, mod.ignore_index,
            mod.dynamic_weights, class_weights=class_weights, grad_scale=mod.grad_scale,
            want_grad=want_grad)
        ctx.dl = dl
        mod.last_terms = out  # [total, ce, dice, w0, w1, w2, -, -] on device (no sync)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        dl = ctx.dl
        ctx.dl = None
        return dl.mul_(g), None, None, None  # g is the 0-dim upstream gradient (1 for loss.backward())


class SimpleLoss(nn.Module):
    def __init__(self, weight_dice=1.0, weight_ce=1.0, ignore_index=255, smooth=1e-5,
                 class_weights=None, dynamic_weights=True):
        super().__init__()
        self.weight_dice = weight_dice
        self.weight_ce = weight_ce
        self.ignore_index = ignore_index
        self.smooth = smooth
        self.class_weights = class_weights
        self.dynamic_weights = dynamic_weights
        self.grad_scale = 1.0  # data-parallel training pre-scales the gradient by 1/world
        self.last_terms = None

    def forward(self, input, target):
        if not input.is_cuda:
            raise RuntimeError("unet-implementations_amd.SimpleLoss runs on MI355X only "
                               "(no CPU fallback exists)")
        if input.shape[-2:] != target.shape[-2:]:
            raise NotImplementedError("logits and target must have the same H, W on the HIP path")
        if target.dtype != torch.int64:
            target = target.long()
        cw = None
        dynamic = bool(self.dynamic_weights) and target.size(0) > 0
        if not dynamic and self.class_weights is not None:
            cw = torch.as_tensor(self.class_weights, dtype=torch.float32,
                                 device=input.device).contiguous()
        mod = self
        if dynamic != bool(self.dynamic_weights):
            raise NotImplementedError("empty batch")
        return _LossFunction.apply(input.contiguous().float(), target.contiguous(), mod, cw)
