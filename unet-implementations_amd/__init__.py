"""unet-implementations_amd — the Our_UNet train step of Ulixes-8/UNet-Implementations
as hand-written gfx950 (MI355X) HIP kernels behind a C ABI, with the reference's
`UNet` / `SimpleLoss` / SGD module surface as the drop-in boundary.

The directory name contains a hyphen, so import it through the shim at the repo
root: `import unet_implementations_amd as ua`.
"""
from . import _lib, ops  # noqa: F401
from ._lib import LIB_PATH, UNetHipError, build, lib  # noqa: F401
from .losses import SimpleLoss  # noqa: F401
from .metrics import SegmentationMetrics  # noqa: F401
from .optim import FusedSGD  # noqa: F401
from .train import (create_lr_scheduler, create_model, create_optimizer,  # noqa: F401
                    get_loss_function, load_checkpoint, save_checkpoint, train_one_epoch,
                    train_step, validate, predict_masks, GraphedTrainStep)
from .clip_unet import CLIPUNet  # noqa: F401
from .unet import ConvBlock, SpatialDropout2d, UNet, UpBlock  # noqa: F401

__all__ = ["UNet", "CLIPUNet", "ConvBlock", "UpBlock", "SpatialDropout2d", "SimpleLoss", "SegmentationMetrics", "FusedSGD",
           "create_model", "create_optimizer", "create_lr_scheduler", "get_loss_function",
           "train_step", "GraphedTrainStep", "train_one_epoch", "save_checkpoint", "load_checkpoint", "validate", "predict_masks", "ops", "build", "lib", "UNetHipError", "LIB_PATH"]
