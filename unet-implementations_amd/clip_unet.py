"""Drop-in for the CLIP_UNet variant (reference: CLIP_UNet/models/unet.py:233-483).

Same network as `UNet` plus a bottleneck fusion: the encoder output [N,512,h/32,w/32] is
concatenated with CLIP image features of the same spatial size and passed through
`clip_fusion_conv` = Conv2d(512 + clip_dim, 512, 1) -> InstanceNorm2d -> LeakyReLU
(CLIP_UNet/models/unet.py:356-362, :441-478).  The CLIP model itself stays an external frozen
feature source (CLIP_UNet/src/train.py:440-470); `forward(x, clip_features)` takes its output.
State dict = the 90 UNet tensors + clip_fusion_conv.{0,1}.{weight,bias}, in the reference's
order (the fusion layer is registered between encoder and decoder).
"""
import torch.nn as nn

from . import ops
from .unet import UNet, _Layer


class CLIPUNet(UNet):
    def __init__(self, *args, with_clip_features: bool = True, clip_dim: int = 512, **kwargs):
        # read by _build_bottleneck(), which UNet.__init__ calls between encoder and decoder
        object.__setattr__(self, "_clip_cfg", (bool(with_clip_features), int(clip_dim)))
        super().__init__(*args, **kwargs)
        self.with_clip_features, self.clip_dim = self._clip_cfg

    def _build_bottleneck(self, common):
        with_clip, clip_dim = self._clip_cfg
        if not with_clip:
            return
        f = self.features_per_stage[-1]
        self.clip_fusion_conv = nn.Sequential(
            nn.Conv2d(f + clip_dim, f, kernel_size=1, bias=common["conv_bias"]),
            common["norm_op"](f, **common["norm_op_kwargs"]),
            common["nonlin"](**common["nonlin_kwargs"]))

    def _build_plan(self):
        plan = super()._build_plan()
        if self.with_clip_features:
            conv, norm, act = (self.clip_fusion_conv[i] for i in range(3))
            if not (isinstance(norm, nn.InstanceNorm2d) and norm.affine
                    and isinstance(act, nn.LeakyReLU) and conv.bias is not None
                    and tuple(conv.kernel_size) == (1, 1)):
                raise NotImplementedError("clip_fusion_conv must be Conv2d(1x1, bias) + "
                                          "InstanceNorm2d(affine) + LeakyReLU on the HIP path")
            self._fusion_layer = _Layer(conv, norm, float(act.negative_slope), None, 1, False,
                                        "clip_fusion_conv", ksize=1)
        return plan

    def forward(self, x, clip_features=None):
        return super().forward(x, clip_features)

    def _bottleneck_input(self, x, clip_features):
        if not self.with_clip_features or clip_features is None:
            return None
        if clip_features.dim() != 4 or clip_features.shape[1] != self.clip_dim:
            raise ValueError(f"clip_features must be [N, {self.clip_dim}, h/32, w/32] (the reference "
                             "re-creates the fusion layer for other widths after the optimizer was "
                             "built, CLIP_UNet/models/unet.py:459-474; pass clip_dim instead)")
        if not clip_features.is_cuda:
            raise RuntimeError("clip_features must live on the ROCm device (no CPU fallback exists)")
        f = clip_features.detach().contiguous().float()
        n_down = self.n_stages - 1
        grid = (x.shape[2] >> n_down, x.shape[3] >> n_down)
        if tuple(f.shape[2:]) != grid:
            # the reference resizes the features to the bottleneck grid
            # (CLIP_UNet/models/unet.py:444-450: bilinear, align_corners=False)
            f = ops.resize_bilinear(f, grid)
        return ops.nchw_to_nhwc(f)
