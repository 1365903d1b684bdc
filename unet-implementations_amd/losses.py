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
    """Forward: the loss value and, in the workspace, the class weights / Dice coefficients the
    gradient needs.  Backward: ONE launch of the gradient kernel with autograd's upstream scalar
    applied inside it (`unet_dice_wce_loss_grad`) - no stock-torch scaling pass over dlogits."""

    @staticmethod
    def forward(ctx, logits, target, mod, class_weights):
        want_grad = ctx.needs_input_grad[0]
        world = mod._world()
        if world > 1:
            import torch.distributed as dist
            stats, ws = ops.dice_wce_loss_shard_stats(logits, target, mod.smooth, mod.ignore_index)
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=mod.process_group)
            out, _ = ops.dice_wce_loss_shard_apply(
                logits, target, stats, logits.shape[0] * world, ws, mod.smooth, mod.weight_dice,
                mod.weight_ce, mod.ignore_index, mod.dynamic_weights, class_weights=class_weights,
                grad_scale=mod.grad_scale, want_grad=False)
        else:
            ws = ops.dice_wce_loss_workspace(logits)
            out, _ = ops.dice_wce_loss_fwd_bwd(
                logits, target, mod.smooth, mod.weight_dice, mod.weight_ce, mod.ignore_index,
                mod.dynamic_weights, class_weights=class_weights, grad_scale=mod.grad_scale,
                want_grad=False, ws=ws)
        if want_grad:
            ctx.held = (logits, target, ws, mod.ignore_index)
        mod.last_terms = out  # [total, ce, dice, w0, w1, w2, -, -] on device (no sync)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        logits, target, ws, ignore_index = ctx.held
        ctx.held = None
        # g is the 0-dim upstream gradient (1 for loss.backward()), read on the device
        return ops.dice_wce_loss_grad(logits, target, ws, g, ignore_index), None, None, None


class SimpleLoss(nn.Module):
    def __init__(self, weight_dice=1.0, weight_ce=1.0, ignore_index=255, smooth=1e-5,
                 class_weights=None, dynamic_weights=True, batch_sync="local", process_group=None):
        """`batch_sync="global"` (data parallel only; not in the reference, which is
        single-process): every rank evaluates the loss of the CONCATENATED batch - class weights,
        CE denominator and the Dice batch mean taken over all ranks' images (equal per-rank batch
        sizes) - so that N ranks x batch b reproduce one process at batch N*b.  Gradients must
        then be summed, not averaged: `ddp.GradBucketAllReduce(..., average=False)`."""
        super().__init__()
        if batch_sync not in ("local", "global"):
            raise ValueError("batch_sync must be 'local' or 'global'")
        self.batch_sync = batch_sync
        self.process_group = process_group
        self.weight_dice = weight_dice
        self.weight_ce = weight_ce
        self.ignore_index = ignore_index
        self.smooth = smooth
        self.class_weights = class_weights
        self.dynamic_weights = dynamic_weights
        self.grad_scale = 1.0  # data-parallel training pre-scales the gradient by 1/world
        self.last_terms = None

    def _world(self):
        if self.batch_sync != "global":
            return 1
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.process_group)

    def forward(self, input, target):
        if not input.is_cuda:
            raise RuntimeError("unet-implementations_amd.SimpleLoss runs on MI355X only "
                               "(no CPU fallback exists)")
        if input.shape[-2:] != target.shape[-2:]:
            # the reference resizes the logits to the target (Our_UNet/models/losses.py:66-68)
            input = ops.resize_bilinear(input, target.shape[-2:])
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
