"""Drop-in `UNet` for the reference's Our_UNet (Our_UNet/models/unet.py:233-432).

Same constructor keywords, same module tree (`encoder_stages[i].block[j]`,
`decoder_stages[i].conv_block.block[j]`, `segmentation_output`) and therefore
the same 90 `state_dict()` keys/shapes, same `forward(x) -> logits` contract
(NCHW fp32 in, NCHW fp32 logits out).  The sub-modules only hold parameters:
`forward` runs the whole network as ONE autograd node whose forward/backward
launch the gfx950 kernels of libunet_hip.so through the C ABI.

Internal layout: activations NHWC fp32.  Default (`fused_pipeline`, fp32 operand mode): per
conv layer only the RAW output y and its per-(n,c) InstanceNorm statistics live in HBM; the
statistics come out of the convolution's epilogue and every consumer (next conv, its weight
gradient, the up-sampling, the head) applies a = dropout(leaky_relu(IN(y))) while it stages
the operand (`ops.Act`).  The stand-alone pipeline (bf16 / bf16x3 operand modes, or
`fused_pipeline = False`) also materialises a.
"""
from typing import Dict, List, Optional, Tuple, Type, Union

import torch
import torch.nn as nn

from . import ops


class SpatialDropout2d(nn.Module):
    """Channel dropout (reference: Our_UNet/models/unet.py:13-35).

    Only carries `drop_prob`.  On the fused path `UNet.forward` draws the masks [N, C]
    (already divided by 1-p) of ALL dropout modules with one Bernoulli launch
    (`_draw_masks`): the same distribution as the reference's per-module
    `new_empty(N, C, 1, 1).bernoulli_(1 - p).div_(1 - p)`, but not the same random stream,
    so parity tests inject masks through `UNet.dropout_mask_override`.
    """

    def __init__(self, drop_prob):
        super().__init__()
        self.drop_prob = drop_prob

    def extra_repr(self):
        return f"drop_prob={self.drop_prob}"

    def draw_mask(self, n, channels, device):
        mask = torch.empty(n, channels, 1, 1, device=device).bernoulli_(1 - self.drop_prob)
        return mask.div_(1 - self.drop_prob).view(n, channels)

    def forward(self, x):  # NCHW tensor, stock torch semantics (not on the fused path)
        if not self.training or self.drop_prob == 0:
            return x
        return x * self.draw_mask(x.size(0), x.size(1), x.device).view(x.size(0), x.size(1), 1, 1)


def _same_padding(kernel_size):
    if isinstance(kernel_size, int):
        return kernel_size // 2
    return (kernel_size[0] // 2, kernel_size[1] // 2)


class ConvBlock(nn.Module):
    """[Conv2d -> norm -> nonlin -> SpatialDropout2d?] x n_convs; stride on the first conv.

    Mirrors Our_UNet/models/unet.py:37-141 (module order inside `.block` decides the
    state_dict indices: 0/1 and 3/4, or 0/1 and 4/5 when dropout modules are present).
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride, n_convs=2, padding=None,
                 norm_op=nn.InstanceNorm2d, norm_op_kwargs=None, dropout_op=None,
                 dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs=None, conv_bias=True,
                 spatial_dropout_rate=0.0):
        super().__init__()
        norm_op_kwargs = {"eps": 1e-5, "affine": True} if norm_op_kwargs is None else norm_op_kwargs
        nonlin_kwargs = {"inplace": True} if nonlin_kwargs is None else nonlin_kwargs
        dropout_op_kwargs = {} if dropout_op_kwargs is None else dropout_op_kwargs
        if padding is None:
            padding = _same_padding(kernel_size)
        mods = []
        cin = in_channels
        for k in range(n_convs):
            mods.append(nn.Conv2d(cin, out_channels, kernel_size, stride if k == 0 else 1, padding,
                                  bias=conv_bias))
            if norm_op is not None:
                mods.append(norm_op(out_channels, **norm_op_kwargs))
            if nonlin is not None:
                mods.append(nonlin(**nonlin_kwargs))
            if spatial_dropout_rate > 0:
                mods.append(SpatialDropout2d(spatial_dropout_rate))
            if dropout_op is not None:
                mods.append(dropout_op(**dropout_op_kwargs))
            cin = out_channels
        self.block = nn.Sequential(*mods)

    def forward(self, x):
        """The block on its own (reference: Our_UNet/models/unet.py:136-141 `self.block(x)`):
        NCHW fp32 in, NCHW fp32 out, differentiable.  Inside `UNet.forward` the blocks are never
        called - the network runs as one fused walk; a stand-alone call runs the same HIP entry
        points layer by layer (conv -> statistics -> InstanceNorm + LeakyReLU + dropout)."""
        return _run_block_standalone(self, x, None)


class UpBlock(nn.Module):
    """Bilinear up-sample to the skip size, concat [up, skip], ConvBlock
    (Our_UNet/models/unet.py:143-231)."""

    def __init__(self, in_channels, skip_channels, out_channels, kernel_size, n_convs=2,
                 norm_op=nn.InstanceNorm2d, norm_op_kwargs=None, dropout_op=None,
                 dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs=None, conv_bias=True,
                 spatial_dropout_rate=0.0):
        super().__init__()
        self.conv_block = ConvBlock(in_channels + skip_channels, out_channels, kernel_size,
                                    stride=1, n_convs=n_convs, padding=None, norm_op=norm_op,
                                    norm_op_kwargs=norm_op_kwargs, dropout_op=dropout_op,
                                    dropout_op_kwargs=dropout_op_kwargs, nonlin=nonlin,
                                    nonlin_kwargs=nonlin_kwargs, conv_bias=conv_bias,
                                    spatial_dropout_rate=spatial_dropout_rate)

    def forward(self, x, skip):
        """Bilinear 2x up-sampling of `x` to the skip's size, cat([up, skip]), conv block
        (reference: Our_UNet/models/unet.py:203-231), stand-alone and differentiable; NCHW fp32.
        The HIP up-sampling kernel is the exact-2x stencil the network uses."""
        return _run_block_standalone(self.conv_block, x, skip)


def _run_block_standalone(block, x, skip):
    if not x.is_cuda:
        raise RuntimeError("unet-implementations_amd blocks run on MI355X only: move the module "
                           "and the input to a ROCm device (no CPU fallback exists)")
    if x.dim() != 4 or (skip is not None and skip.dim() != 4):
        raise ValueError("expected NCHW tensors")
    if skip is not None and (skip.shape[2] != 2 * x.shape[2] or skip.shape[3] != 2 * x.shape[3]):
        raise NotImplementedError("UpBlock on the HIP path up-samples by exactly 2x")
    layers = _parse_block(block, False, type(block).__name__)
    params = [q for l in layers for q in (l.conv.weight, l.conv.bias, l.norm.weight, l.norm.bias)]
    return _BlockFunction.apply(block, layers, x, skip, *params)


class _BlockFunction(torch.autograd.Function):
    """One ConvBlock (optionally behind up-sample + concat) through the stand-alone entry points:
    what `UNet(fused_pipeline=False)` runs per layer."""

    @staticmethod
    def forward(ctx, block, layers, x, skip, *params):
        x0 = ops.nchw_to_nhwc(x.detach().contiguous().float())
        x1 = None
        if skip is not None:
            x0 = ops.upsample2x_fwd(x0)
            x1 = ops.nchw_to_nhwc(skip.detach().contiguous().float())
        override = getattr(block, "dropout_mask_override", None)
        masks = iter(override) if override is not None else None
        recs = []
        for l in layers:
            w = l.conv.weight.detach()
            if w.shape[0] % 32:
                raise NotImplementedError("the HIP convolution needs Cout % 32 == 0")
            wf, wd = ops.pack_conv3x3_weights(w.contiguous())
            y = ops.conv3x3_fwd(x0, x1, wf, l.conv.bias.detach(), l.stride)
            st = ops.instnorm_stats(y, l.norm.weight.detach(), l.norm.bias.detach(), l.norm.eps)
            m = None
            if l.drop is not None and l.drop.drop_prob > 0:
                if masks is not None:
                    m = next(masks).to(device=y.device, dtype=torch.float32).contiguous()
                elif block.training:
                    m = l.drop.draw_mask(y.shape[0], y.shape[3], y.device).contiguous()
            a = ops.instnorm_lrelu_drop_fwd(y, st[2], st[3], m, l.slope)
            recs.append((l, x0, x1, y, st, m, wd))
            x0, x1 = a, None
        ctx.recs, ctx.up = recs, skip is not None
        return ops.nhwc_to_nchw(x0)

    @staticmethod
    def backward(ctx, gout):
        g = ops.nchw_to_nhwc(gout.contiguous().float())
        grads = []
        dx1 = None
        for l, x0, x1, y, st, m, wd in reversed(ctx.recs):
            C = y.shape[3]
            dgm, dbt, dbias = (torch.empty(C, device=y.device) for _ in range(3))
            dy = ops.instnorm_lrelu_drop_bwd(g, y, st[0], st[1], l.norm.weight.detach(),
                                             l.norm.bias.detach(), m, l.slope, dgm, dbt, dbias)
            dw = torch.empty_like(l.conv.weight)
            ops.conv3x3_bwd_weight(x0, dy, dw, 0, l.stride)
            N, H, W, C0 = x0.shape
            if x1 is not None:
                ops.conv3x3_bwd_weight(x1, dy, dw, C0, l.stride)
                dx1 = ops.conv3x3_bwd_data(dy, wd, C0, x1.shape[3], H, W, l.stride)
            first = l is ctx.recs[0][0]
            if first and not ctx.needs_input_grad[2]:
                g = None
            elif C0 % 32:
                raise NotImplementedError("the HIP data gradient needs a multiple of 32 input "
                                          "channels (the RGB stem's input gets no gradient)")
            else:
                g = ops.conv3x3_bwd_data(dy, wd, 0, C0, H, W, l.stride)
            grads = [dw, dbias, dgm, dbt] + grads
        ctx.recs = None
        gx = None
        if g is not None:
            gx = ops.nhwc_to_nchw(ops.upsample2x_bwd(g) if ctx.up else g)
        gskip = ops.nhwc_to_nchw(dx1) if dx1 is not None else None
        return (None, None, gx, gskip, *grads)


def _parse_block(block, first_of_decoder, prefix):
    """[Conv2d -> InstanceNorm2d(affine) -> LeakyReLU -> SpatialDropout2d?] x n of a ConvBlock as
    `_Layer`s; raises NotImplementedError for anything the HIP path does not cover."""
    mods = list(block.block)
    layers, i = [], 0
    while i < len(mods):
        conv = mods[i]
        if not isinstance(conv, nn.Conv2d):
            raise NotImplementedError(f"{prefix}: unexpected module {type(conv).__name__}")
        ks, st = _as_int(conv.kernel_size), _as_int(conv.stride)
        if ks != 3 or st not in (1, 2) or _as_int(conv.padding) != 1 or conv.bias is None \
                or conv.groups != 1 or _as_int(conv.dilation) != 1:
            raise NotImplementedError(
                f"{prefix}.block.{i}: the HIP path covers 3x3/pad 1/stride 1|2 convs with bias")
        i += 1
        norm = mods[i] if i < len(mods) else None
        if not (isinstance(norm, nn.InstanceNorm2d) and norm.affine
                and not norm.track_running_stats):
            raise NotImplementedError(
                f"{prefix}: the HIP path needs InstanceNorm2d(affine=True) after each conv")
        i += 1
        act = mods[i] if i < len(mods) else None
        if not isinstance(act, nn.LeakyReLU):
            raise NotImplementedError(f"{prefix}: the HIP path needs LeakyReLU after the norm")
        i += 1
        drop = None
        if i < len(mods) and isinstance(mods[i], SpatialDropout2d):
            drop = mods[i]
            i += 1
        if i < len(mods) and not isinstance(mods[i], nn.Conv2d):
            raise NotImplementedError(
                f"{prefix}: dropout_op={type(mods[i]).__name__} is not on the HIP path")
        layers.append(_Layer(conv, norm, float(act.negative_slope), drop, st,
                             first_of_decoder and not layers, f"{prefix}.block.{len(layers)}"))
    return layers


class _Layer:
    """One conv3x3 + InstanceNorm + LeakyReLU (+ dropout) unit of the fused plan."""

    __slots__ = ("conv", "norm", "slope", "drop", "stride", "first_of_decoder", "name", "ksize")

    def __init__(self, conv, norm, slope, drop, stride, first_of_decoder, name, ksize=3):
        self.conv, self.norm, self.slope, self.drop = conv, norm, slope, drop
        self.stride, self.first_of_decoder, self.name = stride, first_of_decoder, name
        self.ksize = ksize


def _as_int(v):
    if isinstance(v, (tuple, list)):
        if len(set(v)) != 1:
            return None
        return int(v[0])
    return int(v)


class UNet(nn.Module):
    """6-stage encoder/decoder UNet with the reference's constructor surface."""

    def __init__(self, in_channels: int = 3, num_classes: int = 3, n_stages: int = 6,
                 features_per_stage: List[int] = None,
                 kernel_sizes: List[Tuple[int, int]] = None,
                 strides: List[Tuple[int, int]] = None, n_conv_per_stage: List[int] = None,
                 n_conv_per_stage_decoder: List[int] = None, conv_bias: bool = True,
                 norm_op: Type[nn.Module] = nn.InstanceNorm2d, norm_op_kwargs: Dict = None,
                 dropout_op: Optional[Type[nn.Module]] = None, dropout_op_kwargs: Dict = None,
                 nonlin: Type[nn.Module] = nn.LeakyReLU, nonlin_kwargs: Dict = None,
                 encoder_dropout_rates: List[float] = None,
                 decoder_dropout_rates: List[float] = None):
        super().__init__()
        if features_per_stage is None:
            features_per_stage = [32, 64, 128, 256, 512, 512]
        if kernel_sizes is None:
            kernel_sizes = [[3, 3]] * n_stages
        if strides is None:
            strides = [[1, 1]] + [[2, 2]] * (n_stages - 1)
        if n_conv_per_stage is None:
            n_conv_per_stage = [2] * n_stages
        if n_conv_per_stage_decoder is None:
            n_conv_per_stage_decoder = [2] * (n_stages - 1)
        if norm_op_kwargs is None:
            norm_op_kwargs = {"eps": 1e-5, "affine": True}
        if nonlin_kwargs is None:
            nonlin_kwargs = {"inplace": True}
        if encoder_dropout_rates is None:
            encoder_dropout_rates = [0.0, 0.0, 0.1, 0.2, 0.3, 0.3]
        if decoder_dropout_rates is None:
            decoder_dropout_rates = [0.3, 0.2, 0.2, 0.1, 0.0]

        self.in_channels = in_channels
        self.num_classes = num_classes
        self.n_stages = n_stages
        self.features_per_stage = features_per_stage

        common = dict(norm_op=norm_op, norm_op_kwargs=norm_op_kwargs, dropout_op=dropout_op,
                      dropout_op_kwargs=dropout_op_kwargs, nonlin=nonlin,
                      nonlin_kwargs=nonlin_kwargs, conv_bias=conv_bias)
        self.encoder_stages = nn.ModuleList()
        cin = in_channels
        for s in range(n_stages):
            self.encoder_stages.append(
                ConvBlock(cin, features_per_stage[s], kernel_sizes[s], strides[s],
                          n_convs=n_conv_per_stage[s],
                          spatial_dropout_rate=encoder_dropout_rates[s], **common))
            cin = features_per_stage[s]
        self._fusion_layer = None
        self._build_bottleneck(common)      # registration order = state_dict / arena order
        self.decoder_stages = nn.ModuleList()
        for s in range(n_stages - 1):
            lvl = n_stages - 2 - s
            self.decoder_stages.append(
                UpBlock(features_per_stage[lvl + 1], features_per_stage[lvl],
                        features_per_stage[lvl], kernel_sizes[lvl],
                        n_convs=n_conv_per_stage_decoder[lvl],
                        spatial_dropout_rate=decoder_dropout_rates[s], **common))
        self.segmentation_output = nn.Conv2d(features_per_stage[0], num_classes, kernel_size=1,
                                             stride=1, padding=0, bias=True)
        self.initialize_weights()
        self._plan = None
        self._arena = None       # flat fp32 parameter arena (parameters are views into it)
        self._grad_arena = None  # flat fp32 gradient arena (p.grad are views into it)
        self._offsets = None
        self.dropout_mask_override = None  # test hook: list of [N, C] masks in forward order
        # data-parallel hook: called during backward as `hook(lo)` once every gradient at arena
        # offsets >= lo is final (backward completes the arena back to front)
        self.grad_ready_hook = None
        # "fp32" (default, the parity path) or "bf16": the conv operands (forward, data gradient,
        # stride-1 weight gradient) are rounded to bf16 on chip and contracted on the bf16 matrix
        # cores with fp32 accumulation (tensors, InstanceNorm statistics, master weights: fp32)
        self.matmul_precision = "fp32"
        # fp32 operand mode only: keep just the raw conv outputs in HBM (statistics from the
        # conv epilogue, InstanceNorm + LeakyReLU + dropout applied by the consumers on load)
        self.fused_pipeline = True
        # fp32 fused pipeline: run the stride-1 3x3 layers the Winograd F(2x2,3x3) kernel tiles
        # on it (2.25x fewer matrix-core FLOPs, a few extra fp32 roundings: csrc/conv_wino.hip)
        self.winograd = True
        # fp32 fused pipeline, OFF by default: where a layer's gradient dL/dy is consumed by a
        # Winograd data gradient, apply the InstanceNorm + LeakyReLU + dropout backward in that
        # kernel's loader (which also writes dL/dy for the weight gradient) instead of the
        # elementwise pass.  Measured: 12 of the 22 elementwise passes go (-0.38
        # ms) but the second operand in the loader costs the data gradients +0.43 ms.
        self.fold_instnorm_backward = False
        # normalisation constants of forward(..., input_layout="nhwc_u8") (ImageNet, as the
        # reference's dataset: Our_UNet/src/train.py:303-308)
        self.input_mean, self.input_std = ops.IMAGENET_MEAN, ops.IMAGENET_STD

    def _build_bottleneck(self, common):
        """Hook for variants that add modules between encoder and decoder (CLIPUNet)."""

    # -- reference: Our_UNet/models/unet.py:386-397 --------------------------------------
    def initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="leaky_relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.InstanceNorm2d):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    # -- fused-plan construction -------------------------------------------------------------
    def _block_layers(self, block, first_of_decoder, prefix):
        return _parse_block(block, first_of_decoder, prefix)

    def _build_plan(self):
        enc = [self._block_layers(b, False, f"encoder_stages.{i}")
               for i, b in enumerate(self.encoder_stages)]
        dec = [self._block_layers(b.conv_block, True, f"decoder_stages.{i}.conv_block")
               for i, b in enumerate(self.decoder_stages)]
        head = self.segmentation_output
        if _as_int(head.kernel_size) != 1 or head.in_channels != 32 or head.out_channels != 3:
            raise NotImplementedError("the HIP head kernel is the 32 -> 3 1x1 convolution")
        if enc[0][0].stride != 1:
            raise NotImplementedError("first encoder conv must be stride 1")
        self._plan = (enc, dec)
        return self._plan

    def load_pretrained_encoder(self, pretrained, freeze=True):
        """Counterpart of AE_pretrained/transfer_learning/models/unet.py:409-454: load
        `encoder_stages.*` from a checkpoint (path, full checkpoint dict or state_dict) and
        freeze the encoder.  Files are read with `torch.load(..., weights_only=True)`."""
        ckpt = pretrained
        if isinstance(pretrained, (str, bytes)) or hasattr(pretrained, "__fspath__"):
            ckpt = torch.load(pretrained, map_location="cpu", weights_only=True)
        if "model_state_dict" in ckpt:
            ckpt = ckpt["model_state_dict"]
        if "encoder_stages" in ckpt and isinstance(ckpt["encoder_stages"], dict):
            enc_sd = ckpt["encoder_stages"]
        else:
            enc_sd = {k[len("encoder_stages."):]: v for k, v in ckpt.items()
                      if k.startswith("encoder_stages.")}
        own = self.encoder_stages.state_dict()
        matched = {k: v for k, v in enc_sd.items() if k in own and own[k].shape == v.shape}
        self.encoder_stages.load_state_dict(matched, strict=False)
        if freeze:
            for p in self.encoder_stages.parameters():
                p.requires_grad = False
        return sorted(set(own) - set(matched))     # keys that could not be loaded

    def check_supported(self):
        """Raise NotImplementedError unless this configuration is covered by the HIP path
        (3x3 / pad 1 / stride 1|2 convs with bias, InstanceNorm2d(affine), LeakyReLU,
        optional SpatialDropout2d, 32 -> 3 head).  Works without a GPU."""
        self._build_plan()
        return True

    # -- flat parameter / gradient arenas -------------------------------------------------------
    def _ensure_arena(self, params=None):
        # (a walk of the module tree - 182 modules - costs ~70 us of host time: callers that
        # already hold the parameter list pass it in)
        if params is None:
            params = list(self.parameters())
        dev = params[0].device
        ok = self._arena is not None and self._arena.device == dev
        if ok:
            base = self._arena.data_ptr()
            for p, off in zip(params, self._offsets):
                if p.data_ptr() != base + 4 * off:
                    ok = False
                    break
        if ok:
            return
        offsets, total = [], 0
        for p in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("all UNet parameters must be fp32 on one device")
            offsets.append(total)
            total += (p.numel() + 3) // 4 * 4
        arena = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(params, offsets):
                view = arena[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self._arena, self._offsets = arena, offsets
        self._grad_arena = torch.zeros(total, dtype=torch.float32, device=dev)

    def flat_parameters(self):
        """(param_arena, grad_arena): flat fp32 views that alias every parameter / gradient."""
        self._ensure_arena()
        return self._arena, self._grad_arena

    def _grad_view(self, p):
        idx = self._param_index[id(p)]
        off = self._offsets[idx]
        return self._grad_arena[off:off + p.numel()].view(p.shape)

    # -- forward ----------------------------------------------------------------------------------
    def forward(self, x, extra=None, input_layout="nchw"):
        """`input_layout="nhwc"` takes the fp32 [N,H,W,3] tensor of `ops.preprocess_u8` directly;
        `"nhwc_u8"` takes the dataset's uint8 [N,H,W,3] batch itself: its normalisation
        ((v / 255) - input_mean) / input_std (Our_UNet/src/train.py:303-308) then runs inside the
        loaders of the first convolution (fused pipeline, W % 128 == 0; other cases go through
        `ops.preprocess_u8`)."""
        u8 = None
        if input_layout == "nhwc_u8":
            if x.dtype != torch.uint8:
                raise TypeError("input_layout='nhwc_u8' takes a uint8 [N,H,W,3] tensor")
            fusable = self.matmul_precision == "fp32" and self.fused_pipeline and \
                x.is_cuda and x.dim() == 4 and x.shape[2] % 128 == 0
            if fusable:
                u8 = ops.U8Image(x.contiguous(), self.input_mean, self.input_std)
            else:
                x = ops.preprocess_u8(x.contiguous(), None, self.input_mean, self.input_std)[0]
            x = x.permute(0, 3, 1, 2)
        elif input_layout == "nhwc":
            x = x.permute(0, 3, 1, 2)      # a view: the shape checks below see NCHW sizes
        elif input_layout != "nchw":
            raise ValueError("input_layout must be 'nchw', 'nhwc' or 'nhwc_u8'")
        if not x.is_cuda:
            raise RuntimeError("unet-implementations_amd.UNet runs on MI355X only: move the model "
                               "and the input to a ROCm device (no CPU fallback exists)")
        if x.dim() != 4 or x.shape[1] != self.in_channels or self.in_channels != 3:
            raise ValueError("expected an NCHW batch with 3 channels")
        n_down = self.n_stages - 1
        if x.shape[2] % (1 << n_down) or x.shape[3] % (1 << n_down) or \
                min(x.shape[2], x.shape[3]) < (2 << n_down):
            raise ValueError(f"H and W must be multiples of {1 << n_down} and >= {2 << n_down}")
        if self._plan is None:
            self._build_plan()
        self._check_hooks()
        params = list(self.parameters())
        self._ensure_arena(params)
        self._param_index = {id(p): i for i, p in enumerate(params)}
        if u8 is not None:
            x_nhwc = u8
        elif input_layout != "nchw":
            x_nhwc = x.permute(0, 2, 3, 1).contiguous().float()     # already NHWC in memory
        else:
            x_nhwc = ops.nchw_to_nhwc(x.contiguous().float())
        return _UNetFunction.apply(self, x_nhwc, self._bottleneck_input(x, extra), *params)

    def _check_hooks(self):
        """The fused walk fires hooks of the stage-level modules only (encoder_stages[i],
        decoder_stages[i], its conv_block, segmentation_output, the model itself).  A hook on any
        other sub-module (an inner Conv2d / norm / activation, a `.block` Sequential - the
        reference's Grad-CAM helper accepts any target layer, Our_UNet/utils/visualize.py:
        401-402) would never fire: raise instead of letting the caller fail later on a missing
        feature map."""
        inner = self.__dict__.get("_inner_modules")
        if inner is None:
            ok = {id(self), id(self.segmentation_output), id(self.encoder_stages),
                  id(self.decoder_stages)}
            ok.update(id(m) for m in self.encoder_stages)
            for d in self.decoder_stages:
                ok.update((id(d), id(d.conv_block)))
            inner = self.__dict__["_inner_modules"] = [
                (n, m) for n, m in self.named_modules() if id(m) not in ok]
        for name, m in inner:
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or \
                    m._backward_pre_hooks:
                raise NotImplementedError(
                    f"a hook is registered on {name}: the fused HIP walk never calls inner "
                    "modules, so it would not fire.  Hook a stage-level module instead "
                    "(encoder_stages[i], decoder_stages[i], decoder_stages[i].conv_block, "
                    "segmentation_output).")

    def _bottleneck_input(self, x, extra):
        """Second source of the bottleneck fusion layer (NHWC) or None; CLIPUNet overrides."""
        if extra is not None:
            raise TypeError("this UNet takes no extra bottleneck features")
        return None


def _draw_masks(model, layers, n, device):
    """Dropout masks in forward order; mirrors the reference's per-module draws."""
    if model.dropout_mask_override is not None:
        it = iter(model.dropout_mask_override)
        return [next(it).to(device=device, dtype=torch.float32).contiguous()
                if (l.drop is not None and l.drop.drop_prob > 0) else None for l in layers]
    if not model.training:
        return [None] * len(layers)
    # every SpatialDropout2d's [N, C] mask from ONE bernoulli draw over a cached vector of keep
    # probabilities (2 launches instead of 2 per module; same distribution as the per-module
    # `new_empty(N, C, 1, 1).bernoulli_(1 - p).div_(1 - p)` of the reference)
    sizes = [n * l.conv.out_channels if (l.drop is not None and l.drop.drop_prob > 0) else 0
             for l in layers]
    total = sum(sizes)
    if total == 0:
        return [None] * len(layers)
    key = (n, str(device), tuple(sizes), tuple(l.drop.drop_prob if s else 0.0
                                               for l, s in zip(layers, sizes)))
    cache = model.__dict__.get("_keep_cache")
    if cache is None or cache[0] != key:
        keep = torch.cat([torch.full((s,), 1.0 - l.drop.drop_prob) for l, s in zip(layers, sizes)
                          if s]).to(device)
        cache = model.__dict__["_keep_cache"] = (key, keep)
    keep = cache[1]
    flat = torch.bernoulli(keep).div_(keep)
    out, pos = [], 0
    for l, s in zip(layers, sizes):
        out.append(flat[pos:pos + s].view(n, l.conv.out_channels) if s else None)
        pos += s
    return out


def _hooked(mods, attr):
    return [m for m in mods if getattr(m, attr, None)]


def _fire_forward_hooks(mods, make_output):
    """nn.Module forward hooks of stage-level sub-modules (encoder_stages[i], decoder_stages[i],
    its conv_block, segmentation_output), which the fused walk never calls: the stage output is
    materialised (NCHW fp32) only when such a hook exists and handed to it as `output` (the
    reference's Grad-CAM helper, Our_UNet/utils/visualize.py:392-402).  Hooks observe: a hook
    that returns a replacement output is not supported on this path."""
    mods = _hooked(mods, "_forward_hooks")
    if not mods:
        return
    out = make_output()
    for m in mods:
        for hook in list(m._forward_hooks.values()):
            if hook(m, (None,), out) is not None:
                raise NotImplementedError("forward hooks that replace the output of a sub-module "
                                          "are not supported on the HIP path")


def _fire_backward_hooks(mods, make_grad):
    """register_backward_hook / register_full_backward_hook of the same sub-modules: called with
    grad_output = (dL/d output,) when the backward walk reaches the stage (grad_input is not
    materialised: (None,))."""
    mods = _hooked(mods, "_backward_hooks")
    if not mods:
        return
    g = make_grad()
    for m in mods:
        for hook in list(m._backward_hooks.values()):
            hook(m, (None,), (g,))


class _UNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, extra, *params):
        # The 32-channel layers' kernel choice is made HERE, applied to this walk's calls only and
        # saved for the backward walk (which may run on autograd's own thread, after other
        # models ran): the library's switch is per calling thread.
        c32_mode = "always" if ops.c32_winograd_override() == "always" else bool(model.winograd)
        with ops.c32_winograd_scope(c32_mode):
            logits = _UNetFunction._forward(ctx, model, x, extra, params)
        if any(ctx.needs_input_grad):
            ctx.c32_mode = c32_mode
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        # The slab reductions of the weight gradients (2-3 small launches per gradient, 44 per
        # step) are queued and launched together: at the end of the walk, or - with a
        # data-parallel hook - whenever a stage's gradients are handed to the exchange.
        with ops.c32_winograd_scope(ctx.c32_mode), ops.wgrad_deferral() as deferred:       # (the forward's decision)
            return _UNetFunction._backward(ctx, dlogits, deferred)

    @staticmethod
    def _forward(ctx, model, x, extra, params):
        enc, dec = model._plan
        fusion = model._fusion_layer if extra is not None else None
        layers = [l for blk in enc for l in blk] + ([fusion] if fusion is not None else []) + \
            [l for blk in dec for l in blk]
        need_grad = any(ctx.needs_input_grad)  # False under no_grad / frozen parameters
        N = x.shape[0]
        dev = x.x.device if isinstance(x, ops.U8Image) else x.device
        use_masks = model.training or model.dropout_mask_override is not None
        masks = _draw_masks(model, layers, N, dev) if use_masks else [None] * len(layers)
        mask_of = {id(l): m for l, m in zip(layers, masks)}
        saved = []  # per layer: dict(inputs, y, stats, mask, a)
        if model.matmul_precision not in ("fp32", "bf16", "bf16x3"):
            raise ValueError("matmul_precision must be 'fp32', 'bf16' or 'bf16x3'")
        bf16 = model.matmul_precision      # operand mode handed to every conv call

        # one launch packs every 3x3 weight into the kernels' layouts (persistent buffers); in
        # the fp32 mode the stride-1 layers with >= 64 channels also get their Winograd forms
        # U = G g G^T (forward, incl. the up-sampling loader of the decoder's first convolutions;
        # data gradient: the C -> C layers and the skip halves of the decoder's first
        # convolutions) - ops.conv_wino_supported / conv_up_wino_supported decide per call
        convs = [l.conv.weight for l in layers if l.ksize == 3]
        wino = None
        if bf16 == "fp32" and model.fused_pipeline and model.winograd:
            wino = []
            for l in layers:
                if l.ksize != 3:
                    continue
                co, ci = l.conv.weight.shape[0], l.conv.weight.shape[1]
                s1 = l.stride == 1 and min(co, ci) >= 64
                wino.append((s1 and co % 64 == 0 and ci % 8 == 0,
                             s1 and ci % 64 == 0 and co % 8 == 0))
        # (the 32 -> 32 channel layers' Winograd form needs no weight form of its own: a switch)
        table = model.__dict__.get("_pack_table")
        # (the bf16 planes: all three terms in the split mode; in the mixed-precision mode their
        # first plane is the bf16-rounded weight the patch kernels stage without a conversion)
        planes = 3 if bf16 == "bf16x3" else (1 if (bf16 == "bf16" and model.fused_pipeline) else False)
        if table is None or not table.matches(convs, planes, wino):
            table = model.__dict__["_pack_table"] = ops.PackTable(convs, planes, wino)
        table.run()
        packed = {id(w): k for k, w in enumerate(convs)}

        def run_layer(l, x0, x1):
            w = l.conv.weight
            if l.ksize == 1:
                w2d = w.detach().view(w.shape[0], w.shape[1])
                wd = ops.transpose2d(w2d) if need_grad else None
                y = ops.conv1x1_fwd(x0, x1, w2d, l.conv.bias.detach())
            else:
                k = packed[id(w)]
                wf, wd, wf3, wd3 = table.wf[k], table.wd[k], table.wf3[k], table.wd3[k]
                y = ops.conv3x3_fwd(x0, x1, wf, l.conv.bias.detach(), l.stride, bf16=bf16,
                                    wf3=wf3)
            st = ops.instnorm_stats(y, l.norm.weight.detach(), l.norm.bias.detach(), l.norm.eps)
            m = mask_of[id(l)]
            a = ops.instnorm_lrelu_drop_fwd(y, st[2], st[3], m, l.slope)
            if need_grad:
                saved.append(dict(layer=l, x0=x0, x1=x1, y=y, st=st, mask=m, a=a, wd=wd,
                                  wd3=wd3 if l.ksize == 3 else None))
            return a

        # The fused pipeline serves the fp32 mode and the mixed-precision mode ("bf16": there the
        # layer tensors themselves are bf16 in HBM); 0 <= slope <= 1 (lrelu(z) = max(z, slope z)),
        # one slope for the whole net.  Anything else runs the stand-alone passes.
        slope = layers[0].slope
        fused = model.fused_pipeline and \
            len({l.slope for l in layers}) == 1 and 0.0 <= slope <= 1.0 and \
            not (bf16 == "bf16" and fusion is not None)
        b16 = fused and bf16 == "bf16"
        # split-bf16 mode on the fused pipeline: the stride-1 3x3 layers that tile as 4 x 32
        # pixels run the split patch kernel (forward + data gradient), the rest the fp32 kernels
        x3 = fused and bf16 == "bf16x3"

        # test hook: a list that receives (layer name, raw conv output y, statistics [4,N,C]) of
        # every fused layer in forward order (tests/test_net_gpu.py: the LeakyReLU branch pattern)
        dbg_fwd = getattr(model, "_debug_forward", None)

        def run_layer_fused(l, s0, s1):
            """s0 / s1: ops.Act operands; returns the Act of this layer's output."""
            w = l.conv.weight
            if l.ksize == 1:
                wk = w.detach().view(w.shape[0], w.shape[1])
                wd = ops.transpose2d(wk) if need_grad else None
            else:
                k = packed[id(w)]
                wk, wd = table.wf[k], table.wd[k]
            w3 = table.wf3[k] if ((x3 or b16) and l.ksize == 3) else None
            m = mask_of[id(l)]
            wu = ud = None
            if l.ksize == 3 and not b16 and not x3:
                n_, h_, w_, c0_ = s0.shape
                c1_ = 0 if s1 is None else s1.shape[3]
                if table.uf[k] is not None and s0.alpha is not None and \
                        (s1 is None or s1.alpha is not None) and \
                        ops.conv_wino_supported(n_, h_, w_, c0_, c1_, w.shape[0]):
                    wu = table.uf[k]
                if table.ud[k] is not None and c1_ == 0 and \
                        ops.conv_wino_supported(n_, h_, w_, w.shape[0], 0, c0_):
                    ud = table.ud[k]
            y, st = ops.conv_in_fwd(s0, s1, slope, wk, l.conv.bias.detach(), l.ksize, l.stride,
                                    l.norm.weight.detach(), l.norm.bias.detach(), l.norm.eps, m,
                                    b16=b16, w3=w3, wu=wu)
            if need_grad:
                saved.append(dict(layer=l, x0=s0, x1=s1, y=y, st=st, mask=m, wd=wd,
                                  wd3=table.wd3[k] if w3 is not None else None, ud=ud))
            if dbg_fwd is not None:
                dbg_fwd.append((l.name, y, st))
            return ops.Act(y, st[2], st[3])

        def run_up_layer_fused(l, low, skip):
            """First conv of a decoder stage: conv3x3(cat(upsample2x(act(low)), act(skip))).
            The up-sampled tensor only lives for this call: backward works on `low`
            (ops.conv3x3_up_bwd_weight / _data)."""
            w = l.conv.weight
            if not x3 and l.ksize == 3 and ops.conv_up_in_fwd_supported(low, skip, w.shape[0]):
                # the bilinear gather runs inside the conv's patch loader: no up-sampled tensor
                # (fp32 tensors; bf16 tensors since round 4)
                k = packed[id(w)]
                m = mask_of[id(l)]
                n_, h_, w_, c1_ = skip.shape
                wu = table.uf[k] if (not b16 and table.uf[k] is not None and
                                     low.alpha is not None and
                                     skip.alpha is not None and ops.conv_up_wino_supported(
                                         n_, h_, w_, low.shape[3], c1_, w.shape[0])) else None
                y, st = ops.conv_up_in_fwd(low, skip, slope, table.wf[k], l.conv.bias.detach(),
                                           l.norm.weight.detach(), l.norm.bias.detach(),
                                           l.norm.eps, m, wu=wu,
                                           w3=table.wf3[k] if b16 else None)
                if need_grad:
                    # Winograd form of the data gradient into the skip half
                    n_, h_, w_, c1_ = skip.shape
                    ud1 = table.ud[k] if (table.ud[k] is not None and ops.conv_wino_supported(
                        n_, h_, w_, w.shape[0], 0, c1_)) else None
                    saved.append(dict(layer=l, x0=None, x1=skip, y=y, st=st, mask=m,
                                      wd=table.wd[k], wd3=table.wd3[k] if b16 else None,
                                      x0_low=low, ud1=ud1))
                if dbg_fwd is not None:
                    dbg_fwd.append((l.name, y, st))
                return ops.Act(y, st[2], st[3])
            out = run_layer_fused(l, ops.Act(ops.upsample2x_in_fwd(low, slope)), skip)
            if need_grad:
                saved[-1]["x0"] = None
                saved[-1]["x0_low"] = low
            return out

        if isinstance(x, ops.U8Image) and (not fused or b16):
            raise RuntimeError("the uint8 stem needs the fused fp32 pipeline")
        if fused:
            run_layer = run_layer_fused
            cur = x if isinstance(x, ops.U8Image) else ops.Act(x)   # the image: a plain operand
        else:
            cur = x          # NHWC image
        def stage_output(v):
            """activated stage output as an NCHW fp32 tensor (only materialised for hooks)"""
            if isinstance(v, ops.Act):
                if v.x.dtype != torch.float32:
                    raise NotImplementedError("sub-module hooks on the bf16 pipeline")
                a = v.x if v.alpha is None else ops.instnorm_lrelu_drop_fwd(v.x, v.alpha, v.beta,
                                                                            None, slope)
                return ops.nhwc_to_nchw(a)
            return ops.nhwc_to_nchw(v)

        skips = []
        for bi, blk in enumerate(enc):
            for l in blk:
                cur = run_layer(l, cur, None)
            if bi < len(enc) - 1:
                skips.append(cur)
            _fire_forward_hooks([model.encoder_stages[bi]], lambda: stage_output(cur))
        if fusion is not None:
            if extra.shape[:3] != cur.shape[:3]:
                raise NotImplementedError("bottleneck features must match the 1/32-resolution "
                                          f"grid {tuple(cur.shape[1:3])} (got {tuple(extra.shape[1:3])})")
            cur = run_layer(fusion, cur, ops.Act(extra) if fused else extra)
        for di, blk in enumerate(dec):
            skip = skips[len(skips) - 1 - di]
            if cur.shape[1] * 2 != skip.shape[1] or cur.shape[2] * 2 != skip.shape[2]:
                raise NotImplementedError("decoder up-sampling must be exactly 2x")
            if fused:
                for li, l in enumerate(blk):
                    cur = run_up_layer_fused(l, cur, skip) if li == 0 else run_layer(l, cur, None)
            else:
                up = ops.upsample2x_fwd(cur)
                for li, l in enumerate(blk):
                    cur = run_layer(l, up, skip) if li == 0 else run_layer(l, cur, None)
            _fire_forward_hooks([model.decoder_stages[di], model.decoder_stages[di].conv_block],
                                lambda: stage_output(cur))
        head = model.segmentation_output
        hw = head.weight.detach().view(head.out_channels, -1)
        if fused:
            logits = ops.head1x1_in_fwd(cur, slope, hw, head.bias.detach())
        else:
            logits = ops.head1x1_fwd(cur, hw, head.bias.detach())
        _fire_forward_hooks([head], lambda: logits)
        if need_grad:
            ctx.model = model
            ctx.saved = saved
            ctx.fused = fused
            ctx.slope = slope
            ctx.last = cur
            ctx.n_enc_blocks = len(enc)
            ctx.params = params
            ctx.bf16 = bf16
            ctx.fusion = fusion
        return logits

    @staticmethod
    def _backward(ctx, dlogits, deferred):
        model, saved, params = ctx.model, ctx.saved, ctx.params
        enc, dec = model._plan
        gv = model._grad_view
        dlogits = dlogits.contiguous()
        # The kernels WRITE the gradient arena and the returned views normally become p.grad.
        # A gradient that already exists (a second backward before the step, or
        # zero_grad(set_to_none=False)) must be accumulated into instead: arena-aliased ones
        # are saved here and added back below (autograd gets None for them: adding the returned
        # view to itself would double it); foreign tensors are left to autograd's own `+=`.
        gbase = model._grad_arena.data_ptr()
        carried = {}
        for p, off in zip(params, model._offsets):
            if p.grad is not None and p.grad.data_ptr() == gbase + 4 * off:
                carried[id(p)] = p.grad.detach().clone()
        if carried and model.grad_ready_hook is not None:
            raise RuntimeError(
                "gradient accumulation (a parameter already has .grad) cannot be combined with "
                "the bucketed all-reduce hook: the buckets would ship before the old gradient is "
                "added.  Clear model.grad_ready_hook for the accumulation micro-steps, or call "
                "optimizer.zero_grad() (set_to_none=True) before each backward.")
        touched = set()      # ids of parameters whose gradient this backward produced
        head = model.segmentation_output
        fused, slope = ctx.fused, ctx.slope
        b16_bwd = ctx.bf16 == "bf16"
        x3_bwd = ctx.fused and ctx.bf16 == "bf16x3"
        hw = head.weight.detach().view(head.out_channels, -1)
        if fused:
            # g is the final gradient of the last decoder layer's output: the head's backward
            # also leaves the reductions of that layer's InstanceNorm backward (NextNorm)
            pr_ = saved[-1]
            pl_ = pr_["layer"]
            nxt_h = ops.NextNorm(pr_["y"], pr_["st"], pl_.norm.weight.detach(),
                                 pl_.norm.bias.detach(), pr_["mask"], pl_.slope) \
                if isinstance(ctx.last, ops.Act) and ctx.last.alpha is not None else None
            g = ops.head1x1_in_bwd(ctx.last, slope, dlogits, hw,
                                   gv(head.weight).view(head.out_channels, -1), gv(head.bias),
                                   nxt=nxt_h)
            if nxt_h is not None and nxt_h.tiles > 0:
                pr_["nxt"] = nxt_h
        else:
            g = ops.head1x1_bwd(saved[-1]["a"], dlogits, hw,
                                gv(head.weight).view(head.out_channels, -1), gv(head.bias))
        ctx.last = None
        touched.update(id(q) for q in (head.weight, head.bias))

        idx = len(saved) - 1
        skip_grads = {}
        hook = model.grad_ready_hook

        def ready(module):
            if hook is not None:
                deferred.flush()      # the gradients handed over must be final
                first = next(module.parameters())     # (only with a data-parallel hook installed)
                hook(model._offsets[model._param_index[id(first)]])

        ready(head)
        _fire_backward_hooks([head], lambda: dlogits)

        def grad_nchw(t):
            if t.dtype != torch.float32:
                raise NotImplementedError("sub-module hooks on the bf16 pipeline")
            return ops.nhwc_to_nchw(t)

        def trainable(l):
            return any(q.requires_grad for q in (l.conv.weight, l.conv.bias, l.norm.weight,
                                                 l.norm.bias))

        # Frozen layers (AE-transfer freezes encoder_stages): nothing upstream of
        # the first trainable layer needs a backward pass, and frozen layers in between only
        # propagate the data gradient (no weight-gradient kernels).
        stop = min((i for i, r in enumerate(saved) if trainable(r["layer"])), default=len(saved))
        skip_src = {}        # encoder stage e -> index in `saved` of the layer producing skip e
        pos = 0
        for bi, blk in enumerate(enc):
            pos += len(blk)
            skip_src[bi] = pos - 1

        def layer_bwd(i, g_a, dx0_out=None, dx0_acc=False, need_dx=True, need_dx1=True):
            rec = saved[i]
            l = rec["layer"]
            st = rec["st"]
            touched.update(id(q) for q in (l.conv.weight, l.conv.bias, l.norm.weight, l.norm.bias))
            need_dx = need_dx and i > stop       # the first trainable layer needs no dx
            dbg = getattr(model, "_debug_capture", None)
            if dbg is not None:
                dbg.append((l.name, "ga", g_a.clone()))
            nn_ = rec.pop("nxt", None)     # reductions left by the kernel that produced g_a
            # dx0 of this layer is the final gradient of the previous layer's output (the skip
            # halves dx1 are accumulated into later, by the encoder): its producer also emits
            # that layer's InstanceNorm-backward reductions
            nxt = None
            if fused and need_dx and i > 0:
                pr_ = saved[i - 1]
                pl_ = pr_["layer"]
                nxt = ops.NextNorm(pr_["y"], pr_["st"], pl_.norm.weight.detach(),
                                   pl_.norm.bias.detach(), pr_["mask"], pl_.slope)
            x0, x1 = rec["x0"], rec["x1"]
            low = rec.get("x0_low")
            # InstanceNorm backward applied ON LOAD by the layer's Winograd data gradient (which
            # also writes dy for the weight gradient): no elementwise pass over the layer tensor.
            # Needs the reductions from the producer of g_a and a Winograd data gradient of this dy.
            dx_fold = None
            wgrad_done = False
            fold_ud = rec.get("ud1") if low is not None else rec.get("ud")
            if model.fold_instnorm_backward and nn_ is not None and nn_.tiles > 0 and \
                    fold_ud is not None and not dx0_acc and dx0_out is None and \
                    (need_dx1 if low is not None else (need_dx and x1 is None)):
                coef5, sums = ops.instnorm_bwd_coefs(rec["y"], st[0], st[1], l.norm.weight.detach(),
                                                     l.norm.bias.detach(), rec["mask"],
                                                     (nn_.partial, nn_.tiles))
                w_ = l.conv.weight
                c0_ = low.shape[3] if low is not None else 0
                cc_ = x1.shape[3] if low is not None else x0.shape[3]
                dx_fold, dy = ops.conv3x3_bwd_data_dz(
                    g_a, rec["y"], coef5, sums, l.norm.weight.detach(), st[1], l.slope,
                    gv(l.norm.weight), gv(l.norm.bias), gv(l.conv.bias), fold_ud, w_.shape[1], c0_,
                    cc_, nxt=None if low is not None else nxt)
                if nxt is not None and low is None:
                    saved[i - 1]["nxt"] = nxt
            elif model.fold_instnorm_backward and fused and not b16_bwd and not x3_bwd and \
                    model.winograd and low is None and \
                    x1 is None and l.ksize == 3 and l.stride == 1 and l.conv.weight.requires_grad and \
                    nn_ is not None and nn_.tiles > 0 and isinstance(x0, ops.Act) and \
                    x0.alpha is not None and g_a.dtype == torch.float32 and \
                    ops.conv_in_bwd_weight_dz_supported(*x0.shape, l.conv.weight.shape[0]):
                # 32 -> 32 channel layers: the dy side of the Winograd weight gradient reads every
                # pixel once, so IT applies the InstanceNorm backward on load and writes dz (over
                # g) for the data gradient - no elementwise pass over the layer tensor.  (Off by
                # default like the rest of the fold: measured -0.16 ms of elementwise pass against
                # +0.09 ms in the weight gradient, whose traffic goes from 0.5 to 1.1 GB.)
                coef5, sums = ops.instnorm_bwd_coefs(rec["y"], st[0], st[1], l.norm.weight.detach(),
                                                     l.norm.bias.detach(), rec["mask"],
                                                     (nn_.partial, nn_.tiles))
                dy = ops.conv_in_bwd_weight_dz(x0, slope, g_a, rec["y"], coef5, sums,
                                               l.norm.weight.detach(), st[1], l.slope,
                                               gv(l.norm.weight), gv(l.norm.bias), gv(l.conv.bias),
                                               gv(l.conv.weight), 0)
                wgrad_done = True
            else:
                dy = ops.instnorm_lrelu_drop_bwd(g_a, rec["y"], st[0], st[1],
                                                 l.norm.weight.detach(), l.norm.bias.detach(),
                                                 rec["mask"], l.slope, gv(l.norm.weight),
                                                 gv(l.norm.bias), gv(l.conv.bias),
                                                 partials=(nn_.partial, nn_.tiles)
                                                 if nn_ is not None and nn_.tiles > 0 else None)
            if dbg is not None:
                dbg.append((l.name, "dy", dy.clone()))
            dw = gv(l.conv.weight)
            want_dw = l.conv.weight.requires_grad and not wgrad_done
            if low is not None:
                # conv3x3(upsample2x(act(low))): both gradients of the up-sampled operand are
                # GEMMs over the LOW-resolution pixels once dy is reduced to its nine D_tap
                C0 = low.shape[3]
                D = ops.upsample2x_bwd_taps(dy) if (want_dw or need_dx) else None
                if want_dw:
                    ops.conv3x3_up_bwd_weight(low, slope, D, dw, 0)
                    ops.conv_in_bwd_weight(x1, slope, dy, dw, C0, 3, 1, x3=x3_bwd)
                g_low = ops.conv3x3_up_bwd_data(D, rec["wd"], 0, C0, nxt=nxt,
                                                wd3=rec["wd3"] if ops._is_b16(D) else None) \
                    if need_dx else None
                if nxt is not None:
                    saved[i - 1]["nxt"] = nxt
                dx1 = dx_fold
                if need_dx1 and dx1 is None:
                    dx1 = ops.conv3x3_bwd_data(dy, rec["wd"], C0, x1.shape[3], x1.shape[1],
                                               x1.shape[2], 1, wd3=rec["wd3"],
                                               bf16="bf16x3" if rec["wd3"] is not None else False,
                                               ud=rec.get("ud1"))
                return g_low, dx1
            if want_dw and fused:      # the weight gradient activates its operand on load
                ops.conv_in_bwd_weight(x0, slope, dy, dw, 0, l.ksize, l.stride, x3=x3_bwd)
                if x1 is not None:
                    ops.conv_in_bwd_weight(x1, slope, dy, dw, x0.shape[3], l.ksize, l.stride,
                                           x3=x3_bwd)
                want_dw = False
            if l.ksize == 1:
                if want_dw:
                    dw2d = dw.view(dw.shape[0], dw.shape[1])
                    ops.conv1x1_bwd_weight(x0, dy, dw2d, 0)
                    if x1 is not None:
                        ops.conv1x1_bwd_weight(x1, dy, dw2d, x0.shape[3])
                dx0 = ops.conv1x1_bwd_data(dy, rec["wd"], 0, x0.shape[3]) if need_dx else None
                return dx0, None      # the second source (frozen CLIP features) needs no gradient
            if want_dw:
                ops.conv3x3_bwd_weight(x0, dy, dw, 0, l.stride, bf16=ctx.bf16)
                if x1 is not None:
                    ops.conv3x3_bwd_weight(x1, dy, dw, x0.shape[3], l.stride, bf16=ctx.bf16)
            dx0 = dx1 = None
            N, H, W, C0 = x0.shape
            if dx_fold is not None:
                dx0 = dx_fold
            elif need_dx:
                dx0 = ops.conv3x3_bwd_data(dy, rec["wd"], 0, C0, H, W, l.stride, out=dx0_out,
                                           accumulate=dx0_acc, bf16=ctx.bf16, wd3=rec["wd3"],
                                           nxt=nxt, ud=rec.get("ud"))
                if nxt is not None:
                    saved[i - 1]["nxt"] = nxt
            if x1 is not None and need_dx1:
                dx1 = ops.conv3x3_bwd_data(dy, rec["wd"], C0, x1.shape[3], H, W, l.stride,
                                           bf16=ctx.bf16, wd3=rec["wd3"])
            return dx0, dx1

        done = False
        # decoder stages, last to first
        for di in range(len(dec) - 1, -1, -1):
            blk = dec[di]
            if done or idx < stop:
                done = True
                break
            _fire_backward_hooks([model.decoder_stages[di], model.decoder_stages[di].conv_block],
                                 lambda: grad_nchw(g))
            for li in range(len(blk) - 1, 0, -1):
                if idx < stop:
                    done = True
                    break
                g, _ = layer_bwd(idx, g)
                idx -= 1
            if done or idx < stop:
                done = True
                break
            e = len(enc) - 2 - di
            g_up, g_skip = layer_bwd(idx, g, need_dx1=skip_src[e] >= stop)
            idx -= 1
            if g_skip is not None:
                skip_grads[e] = g_skip
            if g_up is not None:   # fused pipeline: already the low-resolution gradient
                g = g_up if "x0_low" in saved[idx + 1] else ops.upsample2x_bwd(g_up)
            ready(model.decoder_stages[di])
        if ctx.fusion is not None and not done and idx >= stop:
            g, _ = layer_bwd(idx, g)
            idx -= 1
            ready(model.clip_fusion_conv)
        # encoder stages, last to first
        for bi in range(len(enc) - 1, -1, -1):
            blk = enc[bi]
            if done or idx < stop:
                break
            _fire_backward_hooks([model.encoder_stages[bi]], lambda: grad_nchw(g))
            for li in range(len(blk) - 1, -1, -1):
                if idx < stop:
                    done = True
                    break
                first_layer_of_net = (bi == 0 and li == 0)
                if first_layer_of_net:
                    layer_bwd(idx, g, need_dx=False)
                    g = None
                elif li == 0 and (bi - 1) in skip_grads:
                    # input of this layer is the skip tensor of stage bi-1: accumulate into the
                    # gradient the decoder already wrote for it
                    g, _ = layer_bwd(idx, g, dx0_out=skip_grads.pop(bi - 1), dx0_acc=True)
                else:
                    g, _ = layer_bwd(idx, g)
                idx -= 1
            if not done:
                ready(model.encoder_stages[bi])
        deferred.flush()
        if hook is not None:
            # everything below the first trainable parameter is a frozen prefix: no gradient,
            # nothing to exchange
            hook(next((off for p, off in zip(params, model._offsets) if p.requires_grad), 0))
        ctx.saved = None
        grads = []
        for p in params:
            # no gradient for frozen parameters and for layers that did not run (e.g. the CLIP
            # fusion layer when no features were passed): like the reference, .grad stays None
            if not p.requires_grad or id(p) not in touched:
                grads.append(None)
            elif id(p) in carried:
                gv(p).add_(carried[id(p)])     # p.grad is this view: accumulated in place
                grads.append(None)
            else:
                grads.append(gv(p))
        return (None, None, None, *grads)
