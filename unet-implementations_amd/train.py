"""The train step of Our_UNet/src/train.py:592-680 on the HIP path.

`train_step` reproduces the reference's hot-loop order
    optimizer.zero_grad() -> model(images) -> loss_function(outputs, masks)
    -> loss.backward() -> optimizer.step()
(src/train.py:634,:654,:658,:663,:664) without the per-step `loss.item()` sync
(:670): the loss stays a device scalar.  `create_model` / `create_optimizer` /
`create_lr_scheduler` mirror src/train.py:776-798, :431-453 and :456-477.
"""
import torch

from . import ops
from .losses import SimpleLoss
from .optim import FusedSGD
from .unet import UNet


def create_model(device="cuda"):
    """The exact configuration built at Our_UNet/src/train.py:776-795."""
    model = UNet(in_channels=3, num_classes=3, n_stages=6,
                 features_per_stage=[32, 64, 128, 256, 512, 512], kernel_sizes=[[3, 3]] * 6,
                 strides=[[1, 1], [2, 2], [2, 2], [2, 2], [2, 2], [2, 2]],
                 n_conv_per_stage=[2] * 6, n_conv_per_stage_decoder=[2] * 5, conv_bias=True,
                 norm_op=torch.nn.InstanceNorm2d, norm_op_kwargs={"eps": 1e-5, "affine": True},
                 dropout_op=None, nonlin=torch.nn.LeakyReLU, nonlin_kwargs={"inplace": True},
                 encoder_dropout_rates=[0.0, 0.0, 0.1, 0.2, 0.3, 0.3],
                 decoder_dropout_rates=[0.3, 0.2, 0.2, 0.1, 0.0])
    return model.to(device)


def create_optimizer(model, lr=0.005, weight_decay=1e-4, momentum=0.99):
    """SGD + Nesterov as at Our_UNet/src/train.py:445-451 (defaults from :66-99)."""
    return FusedSGD(model.parameters(), lr=lr, weight_decay=weight_decay, momentum=momentum,
                    nesterov=True, model=model)


def create_lr_scheduler(optimizer, max_epochs):
    """Polynomial decay (1 - e/E)^0.9 stepped per epoch (Our_UNet/src/train.py:468-475)."""
    return torch.optim.lr_scheduler.LambdaLR(
        optimizer, lr_lambda=lambda epoch: (1 - epoch / max_epochs) ** 0.9)


def get_loss_function():
    """Default branch of Our_UNet/src/train.py:862-869."""
    return SimpleLoss(weight_dice=1.0, weight_ce=1.0, ignore_index=255, dynamic_weights=True)


def train_step(model, optimizer, loss_function, images, masks, grad_sync=None):
    """One optimisation step; returns the loss as a 0-dim device tensor (no host sync).

    `grad_sync` (optional) is called between backward and the optimizer step; the
    data-parallel wrapper passes its gradient all-reduce finaliser here.
    """
    optimizer.zero_grad()
    outputs = model(images)
    loss = loss_function(outputs, masks)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    optimizer.step()
    return loss.detach()


@torch.no_grad()
def validate(model, val_loader, loss_function, device, ignore_label=255):
    """Counterpart of validate() (Our_UNet/src/train.py:510-589): eval-mode forward, loss, and
    per-batch Dice of the argmax predictions for background / cat / dog, averaged over batches.
    argmax and the nine integer counts come from one kernel and the per-batch Dice arithmetic
    stays on the device, so the loop has no host sync (the reference syncs 4 times per batch)."""
    model.eval()
    val_loss = torch.zeros((), device=device)
    dice_sum = torch.zeros(3, device=device, dtype=torch.float64)
    n = 0
    for batch in val_loader:
        images = batch["image"].to(device, non_blocking=True)
        masks = batch["mask"].to(device, non_blocking=True)
        outputs = model(images)
        val_loss += loss_function(outputs, masks).detach()
        _, counts = ops.argmax_dice_counts(outputs, masks, ignore_label, want_preds=False)
        inter = counts[:, 0].double()
        union = (counts[:, 1] + counts[:, 2]).double()
        dice_sum += torch.where(union > 0, 2.0 * inter / (union + 1e-5), torch.ones_like(inter))
        n += 1
    n = max(n, 1)
    d = (dice_sum / n).tolist()
    scores = {"background": d[0], "cat": d[1], "dog": d[2], "mean_foreground": (d[1] + d[2]) / 2.0}
    return (val_loss / n).item(), scores


@torch.no_grad()
def predict_masks(model, images):
    """Inference as in Our_UNet/src/evaluate.py:185-207: eval-mode forward and per-pixel argmax,
    returned as a uint8 class map on the device (the reference resizes on the CPU afterwards)."""
    was_training = model.training
    model.eval()
    try:
        return ops.argmax_classes(model(images))
    finally:
        model.train(was_training)


def save_checkpoint(model, optimizer, scheduler, epoch, best_dice, output_dir, is_best=False):
    """Same files and dictionary keys as Our_UNet/src/train.py:683-739 (the reference's embedded
    `config` block describes an 8-stage net that is not the model it saves; here it records the
    actual constructor geometry)."""
    import os
    ckpt_dir = os.path.join(str(output_dir), "checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)
    checkpoint = {
        "epoch": epoch,
        "model_state_dict": model.state_dict(),
        "optimizer_state_dict": optimizer.state_dict(),
        "scheduler_state_dict": scheduler.state_dict() if scheduler is not None else None,
        "best_dice": best_dice,
        "config": {"in_channels": model.in_channels, "num_classes": model.num_classes,
                   "n_stages": model.n_stages, "features_per_stage": list(model.features_per_stage),
                   "conv_bias": True, "norm_op_kwargs": {"eps": 1e-5, "affine": True},
                   "nonlin_kwargs": {"inplace": True}},
    }
    path = os.path.join(ckpt_dir, f"checkpoint_epoch_{epoch}.pth")
    torch.save(checkpoint, path)
    if is_best:
        torch.save(checkpoint, os.path.join(str(output_dir), "best_model.pth"))
    return path


def load_checkpoint(path, model, optimizer=None, scheduler=None, device="cuda"):
    """Resume as Our_UNet/src/train.py:888-902 does; returns (start_epoch, best_dice)."""
    checkpoint = torch.load(path, map_location=device, weights_only=True)
    model.load_state_dict(checkpoint["model_state_dict"])
    if optimizer is not None and checkpoint.get("optimizer_state_dict") is not None:
        optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
    if scheduler is not None and checkpoint.get("scheduler_state_dict") is not None:
        scheduler.load_state_dict(checkpoint["scheduler_state_dict"])
    return checkpoint["epoch"] + 1, checkpoint["best_dice"]


def train_one_epoch(model, train_loader, optimizer, loss_function, device, scaler=None):
    """Signature-compatible with the reference's train_one_epoch (src/train.py:592-680)."""
    if scaler is not None:
        raise NotImplementedError("fp16 GradScaler AMP is not part of the fp32 HIP path")
    model.train()
    total = torch.zeros((), device=device)
    n = 0
    for batch in train_loader:
        images = batch["image"].to(device, non_blocking=True)
        masks = batch["mask"].to(device, non_blocking=True)
        total += train_step(model, optimizer, loss_function, images, masks)
        n += 1
    return (total / max(n, 1)).item()
