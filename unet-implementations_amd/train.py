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


class GraphedTrainStep:
    """`train_step` captured once in a HIP graph and replayed: one host call per step instead of
    the ~330 ctypes calls / ~250 kernel launches of the eager walk (5 ms of host time per step;
    the bf16 mode's whole step is 8-11 ms).  The step order is the reference's
    (Our_UNet/src/train.py:634-664); shapes are fixed at capture (the reference trains on fixed
    512x512 crops), dropout masks are drawn inside the graph by torch's graph-safe generator,
    the learning rate is read from device memory (FusedSGD.use_device_hyper) so an LR schedule
    needs no re-capture.

        step = GraphedTrainStep(model, optimizer, loss_function, images, masks)
        loss = step(images, masks)        # device scalar, no host sync

    Data parallel: pass the model's `ddp.GradBucketAllReduce` as `grad_sync`.  Its bucketed
    all-reduces are issued from the backward hooks while the step is being captured, so with
    backend "nccl" (RCCL) they become nodes of the graph on RCCL's stream - forked from the
    backward kernels where a bucket becomes final and joined before the SGD launch, exactly the
    overlap of the eager step - and every rank replays its step with one host call.  The gloo
    backend runs its collectives on the host and cannot be captured: use `train_step` there.

    Capturing runs `warmup` throw-away steps first (kernel attributes, workspaces, the packing
    table, RCCL's communicator); parameters, momentum - including momentum just loaded from a
    checkpoint - and the step counter are restored afterwards.
    """

    def __init__(self, model, optimizer, loss_function, images, masks, warmup=2, grad_sync=None):
        if not images.is_cuda:
            raise RuntimeError("GraphedTrainStep needs ROCm tensors (no CPU fallback exists)")
        if not isinstance(optimizer, FusedSGD):
            raise TypeError("GraphedTrainStep needs the FusedSGD optimizer")
        if grad_sync is None and model.grad_ready_hook is not None:
            raise RuntimeError("the model has a data-parallel gradient hook: pass its "
                               "GradBucketAllReduce as grad_sync so the exchange is captured")
        finish = None
        if grad_sync is not None:
            import torch.distributed as dist
            backend = dist.get_backend(getattr(grad_sync, "group", None))
            if backend != "nccl":
                raise RuntimeError(f"the '{backend}' backend runs its collectives on the host and "
                                   "cannot be captured in a HIP graph: use train_step (eager)")
            finish = grad_sync.finish
        self.model, self.optimizer, self.loss_function = model, optimizer, loss_function
        self.grad_sync = grad_sync
        self.images = images.detach().clone()
        self.masks = masks.detach().clone()
        arena, _ = model.flat_parameters()
        keep_arena = arena.detach().clone()
        # the momentum the optimizer holds NOW (zeros for a fresh one, the loaded buffers after
        # load_state_dict) is what the first replay must start from
        optimizer.adopt_flat_momentum()
        keep_buf = optimizer._flat_buf.detach().clone()
        keep_steps = optimizer._steps
        optimizer.use_device_hyper(True)
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                train_step(model, optimizer, loss_function, self.images, self.masks,
                           grad_sync=finish)
        cur.wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # With a process group alive, its watchdog thread polls the events of earlier collectives
        # (hipEventQuery) while this thread captures: under the default "global" capture mode that
        # query is an error ("operation not permitted when stream is capturing") and takes the
        # process down.  "thread_local" confines the checks to the capturing thread.
        mode = "thread_local" if grad_sync is not None else "global"
        if grad_sync is not None:
            torch.cuda.synchronize()      # the warm-up collectives have completed
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.loss = train_step(model, optimizer, loss_function, self.images, self.masks,
                                   grad_sync=finish)
        # undo the throw-away steps (the captured step itself did not execute)
        with torch.no_grad():
            arena.copy_(keep_arena)
            optimizer._flat_buf.copy_(keep_buf)
        optimizer._steps = keep_steps

    def __call__(self, images, masks):
        if images.shape != self.images.shape or masks.shape != self.masks.shape:
            raise ValueError("GraphedTrainStep was captured for batches of shape "
                             f"{tuple(self.images.shape)} / {tuple(self.masks.shape)}")
        if images.data_ptr() != self.images.data_ptr():
            self.images.copy_(images, non_blocking=True)
        if masks.data_ptr() != self.masks.data_ptr():
            self.masks.copy_(masks, non_blocking=True)
        self.optimizer.sync_device_hyper()
        self.graph.replay()
        self.optimizer._steps += 1
        return self.loss


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
