"""Per-kernel parity (-m gpu): every C-ABI entry point against the oracle's stock-torch
CPU ops on seeded inputs, and against the reference-generated fixtures in
tests/golden/ops_small.npz.  Tolerances are relative to the tensor's max magnitude:
fp32 with a different summation order, never looser than 5e-5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def to_nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(DEV)


def from_nhwc(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def check(a, b, tol, what=""):
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = relerr(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# --------------------------------------------------------------------------- layout / pack
def test_layout_roundtrip(ua):
    x = rnd(2, 3, 10, 14, seed=1)
    y = ua.ops.nchw_to_nhwc(x.to(DEV))
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    z = ua.ops.nhwc_to_nchw(y)
    assert torch.equal(z.cpu(), x)


def test_pack_weights(ua):
    w = rnd(64, 32, 3, 3, seed=2)
    wf, wd = ua.ops.pack_conv3x3_weights(w.to(DEV))
    assert torch.equal(wf.cpu(), w.permute(2, 3, 0, 1).reshape(9, 64, 32))   # [tap][co][ci]
    assert torch.equal(wd.cpu(), w.permute(2, 3, 1, 0).reshape(9, 32, 64))   # [tap][ci][co]


def test_pack_weights_batched_matches_per_layer(ua):
    ws = [rnd(32, 3, 3, 3, seed=1).to(DEV), rnd(64, 32, 3, 3, seed=2).to(DEV),
          rnd(128, 96, 3, 3, seed=3).to(DEV), rnd(32, 64, 3, 3, seed=4).to(DEV)]
    for planes in (False, True):
        tab = ua.ops.PackTable(ws, planes)
        tab.run()
        for i, w in enumerate(ws):
            wf, wd = ua.ops.pack_conv3x3_weights(w)
            assert torch.equal(tab.wf[i], wf) and torch.equal(tab.wd[i], wd)
            if planes and w.shape[1] != 3:
                wf3, wd3 = ua.ops.pack_conv3x3_weights_bf16x3(w)
                assert torch.equal(tab.wf3[i], wf3) and torch.equal(tab.wd3[i], wd3)
            else:
                assert tab.wf3[i] is None
        assert tab.matches(ws, planes) and not tab.matches(ws[:2], planes)


# --------------------------------------------------------------------------- conv forward
CONV_SHAPES = [
    # N, H, W, C0, C1, Cout, stride
    (2, 12, 20, 32, 0, 32, 1),
    (2, 12, 20, 32, 0, 64, 2),
    (1, 16, 16, 64, 32, 128, 1),
    (2, 8, 8, 128, 0, 128, 1),
    (1, 24, 40, 3, 0, 32, 1),
    (3, 10, 6, 3, 0, 64, 1),
    (2, 6, 128, 3, 0, 32, 1),       # RGB stem, raw-row form (W % 128 == 0)
    (1, 3, 256, 3, 0, 64, 1),
    (2, 64, 64, 64, 0, 64, 1),
    (1, 16, 16, 512, 0, 512, 1),
    (2, 2, 2, 512, 0, 512, 1),
    (1, 4, 4, 512, 512, 512, 1),
    (2, 34, 18, 32, 0, 32, 2),
    (1, 128, 128, 64, 32, 32, 1),
    (1, 32, 32, 256, 128, 128, 1),
    (2, 4, 256, 32, 0, 32, 1),      # row-fused kernel (W % 128 == 0, Cout 32)
    (1, 3, 128, 64, 32, 32, 1),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv3x3_fwd(ua, shape):
    N, H, W, C0, C1, Cout, s = shape
    x = rnd(N, C0 + C1, H, W, seed=3)
    w = rnd(Cout, C0 + C1, 3, 3, seed=4, scale=0.1)
    b = rnd(Cout, seed=5)
    ref = F.conv2d(x, w, b, stride=s, padding=1)
    wf, _ = ua.ops.pack_conv3x3_weights(w.to(DEV))
    x0 = to_nhwc(x[:, :C0])
    x1 = to_nhwc(x[:, C0:]) if C1 else None
    y = ua.ops.conv3x3_fwd(x0, x1, wf, b.to(DEV), s)
    check(from_nhwc(y), ref, 2e-5, f"conv fwd {shape}")


BF16_SHAPES = [(2, 12, 20, 32, 0, 32, 1), (2, 12, 20, 32, 0, 64, 2), (1, 16, 16, 64, 32, 128, 1),
               (1, 16, 16, 512, 0, 512, 1), (2, 2, 2, 512, 0, 512, 1), (1, 64, 64, 64, 0, 64, 1),
               (1, 8, 128, 64, 32, 32, 1), (1, 24, 40, 3, 0, 32, 1)]


def _bf(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("shape", BF16_SHAPES)
def test_conv3x3_fwd_bf16(ua, shape):
    """bf16 operands, fp32 accumulation: exact products, so the result matches an fp32 conv of
    the bf16-rounded operands up to summation order (the RGB stem stays fp32)."""
    N, H, W, C0, C1, Cout, s = shape
    x = rnd(N, C0 + C1, H, W, seed=3)
    w = rnd(Cout, C0 + C1, 3, 3, seed=4, scale=0.1)
    b = rnd(Cout, seed=5)
    ref = F.conv2d(x, w, b, stride=s, padding=1) if C0 == 3 else \
        F.conv2d(_bf(x), _bf(w), b, stride=s, padding=1)
    wf, _ = ua.ops.pack_conv3x3_weights(w.to(DEV))
    x0 = to_nhwc(x[:, :C0])
    x1 = to_nhwc(x[:, C0:]) if C1 else None
    y = ua.ops.conv3x3_fwd(x0, x1, wf, b.to(DEV), s, bf16=True)
    check(from_nhwc(y), ref, 2e-5, f"bf16 conv fwd {shape}")
    full = F.conv2d(x, w, b, stride=s, padding=1)     # and it is a bf16-accurate conv
    assert relerr(from_nhwc(y), full) < 2e-2


@pytest.mark.parametrize("shape", [(2, 12, 20, 32, 32, 1, (0, 32)), (1, 16, 16, 96, 64, 1, (32, 64)),
                                   (2, 16, 24, 32, 64, 2, (0, 32)), (2, 4, 4, 512, 512, 2, (0, 512)),
                                   (1, 64, 64, 64, 64, 1, (0, 64))])
def test_conv3x3_bwd_data_bf16(ua, shape):
    N, H, W, Cin, Cout, s, (off, cc) = shape
    x = rnd(N, Cin, H, W, seed=6).requires_grad_(True)
    w = rnd(Cout, Cin, 3, 3, seed=7, scale=0.1)
    y = F.conv2d(x, _bf(w), None, stride=s, padding=1)
    gy = rnd(*y.shape, seed=8)
    (gx,) = torch.autograd.grad(y, x, _bf(gy))
    _, wd = ua.ops.pack_conv3x3_weights(w.to(DEV))
    dx = ua.ops.conv3x3_bwd_data(to_nhwc(gy), wd, off, cc, H, W, s, bf16=True)
    check(from_nhwc(dx), gx[:, off:off + cc], 2e-5, f"bf16 dgrad {shape}")


@pytest.mark.parametrize("shape", [(2, 12, 64, 32, 0, 32, 32, 1), (1, 16, 16, 64, 0, 64, 64, 1),
                                   (2, 8, 8, 128, 0, 128, 128, 1), (1, 32, 32, 64, 32, 96, 64, 1),
                                   (1, 64, 128, 32, 0, 32, 32, 1), (1, 32, 64, 32, 0, 32, 64, 1),
                                   (2, 16, 24, 32, 0, 32, 64, 2), (3, 34, 70, 32, 0, 32, 32, 1)])
def test_conv3x3_bwd_weight_bf16(ua, shape):
    """bf16 operands via ds_read_b64_tr_b16, fp32 sums: equals the fp32 weight gradient of the
    bf16-rounded x and dy (shapes the bf16 kernel does not cover fall back to fp32)."""
    N, H, W, Cx, off, Ct, Cout, s = shape
    x = rnd(N, Ct, H, W, seed=10)
    w = rnd(Cout, Ct, 3, 3, seed=11, scale=0.1).requires_grad_(True)
    y = F.conv2d(_bf(x), w, None, stride=s, padding=1)
    gy = rnd(*y.shape, seed=12)
    (gw,) = torch.autograd.grad(y, w, _bf(gy))
    y32 = F.conv2d(x, w, None, stride=s, padding=1)
    (gw32,) = torch.autograd.grad(y32, w, gy)
    dw = torch.zeros((Cout, Ct, 3, 3), device=DEV)
    ua.ops.conv3x3_bwd_weight(to_nhwc(x[:, off:off + Cx]), to_nhwc(gy), dw, off, s, bf16=True)
    got = dw[:, off:off + Cx].cpu()
    e_bf, e_32 = relerr(got, gw[:, off:off + Cx]), relerr(got, gw32[:, off:off + Cx])
    on_bf16_kernel = s == 1 and not (Cx == 32 and Cout == 32 and W < 64)   # mirrors make_plan()
    if on_bf16_kernel:
        assert e_bf <= 3e-5 and e_32 > 1e-4, \
            f"bf16 wgrad {shape}: vs bf16-rounded {e_bf:.2e}, vs fp32 {e_32:.2e}"
    else:
        assert e_32 <= 3e-5, f"fp32 fallback {shape}: {e_32:.2e}"


# --------------------------------------------------------------------------- conv dgrad
DGRAD_SHAPES = [
    # N, H, W, Cin, Cout, stride, (ci_offset, ccols)
    (2, 12, 20, 32, 32, 1, (0, 32)),
    (2, 12, 20, 64, 32, 1, (32, 32)),
    (1, 16, 16, 96, 64, 1, (0, 64)),
    (2, 16, 24, 32, 64, 2, (0, 32)),
    (1, 8, 8, 128, 256, 2, (0, 128)),
    (2, 4, 4, 512, 512, 2, (0, 512)),
    (1, 32, 32, 384, 128, 1, (256, 128)),
    (1, 64, 64, 64, 64, 1, (0, 64)),
    (2, 32, 32, 64, 128, 2, (0, 64)),
    (2, 16, 16, 128, 128, 1, (0, 128)),
    (2, 64, 64, 32, 64, 2, (0, 32)),
    (2, 5, 128, 32, 32, 1, (0, 32)),     # row-fused kernel
    (1, 4, 256, 96, 32, 1, (64, 32)),
]


@pytest.mark.parametrize("shape", DGRAD_SHAPES)
def test_conv3x3_bwd_data(ua, shape):
    N, H, W, Cin, Cout, s, (off, cc) = shape
    x = rnd(N, Cin, H, W, seed=6).requires_grad_(True)
    w = rnd(Cout, Cin, 3, 3, seed=7, scale=0.1)
    y = F.conv2d(x, w, None, stride=s, padding=1)
    gy = rnd(*y.shape, seed=8)
    (gx,) = torch.autograd.grad(y, x, gy)
    _, wd = ua.ops.pack_conv3x3_weights(w.to(DEV))
    dx = ua.ops.conv3x3_bwd_data(to_nhwc(gy), wd, off, cc, H, W, s)
    check(from_nhwc(dx), gx[:, off:off + cc], 2e-5, f"dgrad {shape}")
    # accumulate flag adds into the existing tensor
    base = rnd(N, cc, H, W, seed=9)
    buf = to_nhwc(base)
    ua.ops.conv3x3_bwd_data(to_nhwc(gy), wd, off, cc, H, W, s, out=buf, accumulate=True)
    check(from_nhwc(buf), gx[:, off:off + cc] + base, 2e-5, f"dgrad acc {shape}")


# --------------------------------------------------------------------------- 1x1 conv (CLIP fusion)
@pytest.mark.parametrize("shape", [(2, 2, 2, 512, 512, 512), (1, 16, 16, 512, 512, 512),
                                   (2, 4, 6, 64, 32, 96), (3, 5, 7, 32, 0, 32)])
def test_conv1x1(ua, shape):
    N, H, W, C0, C1, Cout = shape
    x = rnd(N, C0 + C1, H, W, seed=40).requires_grad_(True)
    w = rnd(Cout, C0 + C1, 1, 1, seed=41, scale=0.1).requires_grad_(True)
    b = rnd(Cout, seed=42)
    y = F.conv2d(x, w, b)
    gy = rnd(*y.shape, seed=43)
    gx, gw = torch.autograd.grad(y, (x, w), gy)
    w2d = w.detach().view(Cout, C0 + C1).to(DEV).contiguous()
    x0 = to_nhwc(x.detach()[:, :C0])
    x1 = to_nhwc(x.detach()[:, C0:]) if C1 else None
    yd = ua.ops.conv1x1_fwd(x0, x1, w2d, b.to(DEV))
    check(from_nhwc(yd), y.detach(), 2e-5, f"conv1x1 fwd {shape}")
    wT = ua.ops.transpose2d(w2d)
    assert torch.equal(wT.cpu(), w2d.cpu().t().contiguous())
    dx0 = ua.ops.conv1x1_bwd_data(to_nhwc(gy), wT, 0, C0)
    check(from_nhwc(dx0), gx[:, :C0], 2e-5, f"conv1x1 dgrad {shape}")
    if C1:
        dx1 = ua.ops.conv1x1_bwd_data(to_nhwc(gy), wT, C0, C1)
        check(from_nhwc(dx1), gx[:, C0:], 2e-5, f"conv1x1 dgrad slice {shape}")
    dw = torch.full((Cout, C0 + C1), 3.0, device=DEV)
    ua.ops.conv1x1_bwd_weight(x0, to_nhwc(gy), dw, 0)
    if C1:
        ua.ops.conv1x1_bwd_weight(x1, to_nhwc(gy), dw, C0)
    check(dw.cpu(), gw.view(Cout, C0 + C1), 3e-5, f"conv1x1 wgrad {shape}")


# --------------------------------------------------------------------------- conv wgrad
WGRAD_SHAPES = [
    # N, H, W, Cx, ci_offset, Cin_total, Cout, stride
    (2, 12, 20, 32, 0, 32, 32, 1),
    (2, 12, 40, 32, 0, 32, 32, 1),
    (2, 16, 24, 32, 0, 32, 64, 2),
    (1, 16, 16, 64, 0, 64, 64, 1),
    (2, 8, 8, 128, 0, 128, 128, 1),
    (2, 8, 8, 64, 0, 64, 128, 2),
    (1, 32, 32, 64, 32, 96, 32, 1),
    (1, 24, 40, 3, 0, 3, 32, 1),
    (2, 6, 128, 3, 0, 3, 32, 1),       # RGB stem, raw-row form (W % 128 == 0)
    (1, 5, 256, 3, 0, 3, 64, 1),
    (2, 2, 2, 512, 0, 512, 512, 1),
    (2, 4, 4, 512, 0, 512, 512, 2),
    (1, 64, 64, 64, 0, 64, 64, 1),
    (3, 34, 70, 32, 0, 32, 32, 1),
    (2, 32, 32, 64, 0, 64, 128, 2),
    (2, 64, 64, 32, 0, 32, 64, 2),
    (2, 16, 16, 128, 0, 128, 256, 2),
    (1, 64, 64, 64, 0, 64, 128, 2),
]


@pytest.mark.parametrize("shape", WGRAD_SHAPES)
def test_conv3x3_bwd_weight(ua, shape):
    N, H, W, Cx, off, Ct, Cout, s = shape
    x = rnd(N, Ct, H, W, seed=10)
    w = rnd(Cout, Ct, 3, 3, seed=11, scale=0.1).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    y = F.conv2d(x, w, b, stride=s, padding=1)
    gy = rnd(*y.shape, seed=12)
    gw, gb = torch.autograd.grad(y, (w, b), gy)
    dw = torch.full((Cout, Ct, 3, 3), 7.0, device=DEV)
    db = torch.empty(Cout, device=DEV)
    ua.ops.conv3x3_bwd_weight(to_nhwc(x[:, off:off + Cx]), to_nhwc(gy), dw, off, s, db=db)
    check(dw[:, off:off + Cx].cpu(), gw[:, off:off + Cx], 3e-5, f"wgrad {shape}")
    check(db.cpu(), gb, 3e-5, f"bias grad {shape}")
    if off > 0:  # untouched columns keep their value
        assert torch.all(dw[:, :off] == 7.0)


# --------------------------------------------------------------------------- instance norm
IN_SHAPES = [(2, 16, 16, 128), (2, 12, 20, 32), (2, 16, 16, 64), (1, 8, 8, 512), (2, 2, 2, 512), (1, 64, 64, 32),
             (2, 33, 7, 128), (1, 128, 128, 32)]


@pytest.mark.parametrize("shape", IN_SHAPES)
@pytest.mark.parametrize("with_mask", [False, True])
def test_instnorm_lrelu_drop(ua, shape, with_mask):
    N, H, W, C = shape
    y = (rnd(N, C, H, W, seed=13) * 2.0 + 3.0).requires_grad_(True)   # large mean: cancellation test
    gamma = (1 + 0.1 * rnd(C, seed=14)).requires_grad_(True)
    beta = (0.1 * rnd(C, seed=15)).requires_grad_(True)
    mask = None
    if with_mask:
        g = torch.Generator().manual_seed(16)
        mask = torch.empty(N, C).bernoulli_(0.7, generator=g).div_(0.7)
    z = F.leaky_relu(F.instance_norm(y, weight=gamma, bias=beta, eps=1e-5), 0.01)
    a_ref = z * mask.view(N, C, 1, 1) if with_mask else z
    ga = rnd(N, C, H, W, seed=17)
    gy_ref, gg_ref, gb_ref = torch.autograd.grad(a_ref, (y, gamma, beta), ga)

    yd = to_nhwc(y.detach())
    st = ua.ops.instnorm_stats(yd, gamma.detach().to(DEV), beta.detach().to(DEV), 1e-5)
    mean_ref = y.detach().mean(dim=(2, 3))
    var_ref = y.detach().var(dim=(2, 3), unbiased=False)
    check(st[0].cpu(), mean_ref, 1e-6, "mean")
    check(st[1].cpu(), 1.0 / torch.sqrt(var_ref + 1e-5), 2e-6, "rstd")
    md = mask.to(DEV) if with_mask else None
    a = ua.ops.instnorm_lrelu_drop_fwd(yd, st[2], st[3], md, 0.01)
    check(from_nhwc(a), a_ref.detach(), 1e-5, "IN apply")
    dg = torch.empty(C, device=DEV)
    dbt = torch.empty(C, device=DEV)
    dbias = torch.empty(C, device=DEV)
    dy = ua.ops.instnorm_lrelu_drop_bwd(to_nhwc(ga), yd, st[0], st[1], gamma.detach().to(DEV),
                                        beta.detach().to(DEV), md, 0.01, dg, dbt, dbias)
    check(from_nhwc(dy), gy_ref, 5e-5, "IN bwd dy")
    check(dg.cpu(), gg_ref, 5e-5, "dgamma")
    check(dbt.cpu(), gb_ref, 5e-5, "dbeta")
    # conv-bias gradient = sum of dy: mathematically 0, numerically tiny
    assert dbias.abs().max().item() <= 1e-3 * max(1.0, gy_ref.abs().sum().item() / C)


# --------------------------------------------------------------------------- upsample
@pytest.mark.parametrize("shape", [(2, 8, 12, 64), (1, 1, 1, 32), (2, 2, 2, 512), (1, 16, 16, 32),
                                   (1, 5, 3, 128)])
def test_upsample2x(ua, shape):
    N, h, w, C = shape
    x = rnd(N, C, h, w, seed=18).requires_grad_(True)
    y = F.interpolate(x, size=(2 * h, 2 * w), mode="bilinear", align_corners=False)
    gy = rnd(*y.shape, seed=19)
    (gx,) = torch.autograd.grad(y, x, gy)
    yd = ua.ops.upsample2x_fwd(to_nhwc(x.detach()))
    check(from_nhwc(yd), y.detach(), 1e-6, "upsample fwd")
    gxd = ua.ops.upsample2x_bwd(to_nhwc(gy))
    check(from_nhwc(gxd), gx, 1e-6, "upsample bwd")
    base = rnd(N, C, h, w, seed=20)
    buf = to_nhwc(base)
    ua.ops.upsample2x_bwd(to_nhwc(gy), out=buf, accumulate=True)
    check(from_nhwc(buf), gx + base, 1e-6, "upsample bwd acc")


# --------------------------------------------------------------------------- head
@pytest.mark.parametrize("shape", [(2, 12, 20), (1, 64, 64), (3, 17, 9), (2, 128, 128)])
def test_head1x1(ua, shape):
    N, H, W = shape
    a = rnd(N, 32, H, W, seed=21).requires_grad_(True)
    w = rnd(3, 32, 1, 1, seed=22, scale=0.3).requires_grad_(True)
    b = rnd(3, seed=23).requires_grad_(True)
    y = F.conv2d(a, w, b)
    gy = rnd(*y.shape, seed=24)
    ga, gw, gb = torch.autograd.grad(y, (a, w, b), gy)
    ad = to_nhwc(a.detach())
    wd = w.detach().view(3, 32).to(DEV)
    logits = ua.ops.head1x1_fwd(ad, wd, b.detach().to(DEV))
    check(logits.cpu(), y.detach(), 1e-5, "head fwd")
    dw = torch.empty(3, 32, device=DEV)
    db = torch.empty(3, device=DEV)
    da = ua.ops.head1x1_bwd(ad, gy.to(DEV), wd, dw, db)
    check(from_nhwc(da), ga, 1e-5, "head da")
    check(dw.cpu(), gw.view(3, 32), 2e-5, "head dw")
    check(db.cpu(), gb, 2e-5, "head db")


# --------------------------------------------------------------------------- loss
def _loss_case(seed, N, H, W, drop_class=None):
    g = torch.Generator().manual_seed(seed)
    lg = torch.randn(N, 3, H, W, generator=g) * 2.0
    tg = torch.randint(0, 3, (N, H, W), generator=g)
    if drop_class is not None:
        tg[tg == drop_class] = 0
    tg[:, :2, :] = 255
    tg[:, :, -1:] = 255
    return lg, tg


@pytest.mark.parametrize("case", [(1, 2, 24, 40, None), (2, 3, 16, 16, 2), (3, 1, 64, 64, 1),
                                  (4, 2, 128, 128, None)])
def test_loss_vs_oracle(ua, case):
    seed, N, H, W, drop = case
    lg, tg = _loss_case(seed, N, H, W, drop)
    lgr = lg.clone().requires_grad_(True)
    ref = O.simple_loss(lgr, tg)
    ref.backward()
    out, dl = ua.ops.dice_wce_loss_fwd_bwd(lg.to(DEV), tg.to(DEV), 1e-5, 1.0, 1.0, 255, True)
    assert abs(out[0].item() - ref.item()) <= 2e-6 * abs(ref.item())
    check(out[3:6].cpu(), O.class_weights(tg), 1e-6, "class weights")
    check(dl.cpu(), lgr.grad, 2e-5, "dlogits")


def test_loss_all_ignored_image(ua):
    """An image made only of 255 pixels contributes zero counts and dice = 1 (smooth/smooth)."""
    lg, tg = _loss_case(7, 2, 16, 16, None)
    tg[1] = 255
    lgr = lg.clone().requires_grad_(True)
    ref = O.simple_loss(lgr, tg)
    ref.backward()
    out, dl = ua.ops.dice_wce_loss_fwd_bwd(lg.to(DEV), tg.to(DEV), 1e-5, 1.0, 1.0, 255, True)
    assert abs(out[0].item() - ref.item()) <= 2e-6 * abs(ref.item())
    check(dl.cpu(), lgr.grad, 2e-5, "dlogits")
    assert torch.all(dl[1] == 0)


def test_loss_golden(ua, golden):
    g = golden("ops_small")
    lg, tg = torch.from_numpy(g["loss_logits"]), torch.from_numpy(g["loss_target"])
    out, dl = ua.ops.dice_wce_loss_fwd_bwd(lg.to(DEV), tg.to(DEV), 1e-5, 1.0, 1.0, 255, True)
    assert abs(out[0].item() - float(g["loss_value"])) <= 2e-6 * abs(float(g["loss_value"]))
    check(dl.cpu(), torch.from_numpy(g["loss_dlogits"]), 2e-5, "golden dlogits")
    cw = torch.from_numpy(g["loss2_weights"]).to(DEV)
    out2, dl2 = ua.ops.dice_wce_loss_fwd_bwd(lg.to(DEV), tg.to(DEV), 1e-5, 1.0, 1.0, 255, False,
                                             class_weights=cw)
    assert abs(out2[0].item() - float(g["loss2_value"])) <= 2e-6 * abs(float(g["loss2_value"]))
    check(dl2.cpu(), torch.from_numpy(g["loss2_dlogits"]), 2e-5, "golden dlogits (static w)")


def test_loss_gradient_pass_applies_the_upstream_scalar(ua):
    """unet_dice_wce_loss_grad (what SimpleLoss's autograd backward launches): with no upstream
    scalar it writes the very bits the one-call form writes; with dL/dloss = s on the device it
    writes round(dlogits * s), i.e. what the separate `dlogits.mul_(s)` pass of round 3 produced -
    and (s * loss).backward() through the module agrees with the oracle's autograd."""
    lg, tg = _loss_case(5, 2, 64, 96, None)
    lgd, tgd = lg.to(DEV), tg.to(DEV)
    out, dl = ua.ops.dice_wce_loss_fwd_bwd(lgd, tgd, 1e-5, 1.0, 1.0, 255, True)
    ws = ua.ops.dice_wce_loss_workspace(lgd)
    out2, none = ua.ops.dice_wce_loss_fwd_bwd(lgd, tgd, 1e-5, 1.0, 1.0, 255, True, want_grad=False,
                                              ws=ws)
    assert none is None and torch.equal(out[:6], out2[:6])      # (entries 6, 7 are unused)
    assert torch.equal(ua.ops.dice_wce_loss_grad(lgd, tgd, ws, None, 255), dl)
    s = torch.tensor(-2.75, device=DEV)
    assert torch.equal(ua.ops.dice_wce_loss_grad(lgd, tgd, ws, s, 255), dl * s)
    lgr = lg.clone().requires_grad_(True)
    (O.simple_loss(lgr, tg) * 0.3).backward()
    x = lgd.clone().requires_grad_(True)
    (ua.SimpleLoss()(x, tgd) * 0.3).backward()
    check(x.grad.cpu(), lgr.grad, 2e-5, "dlogits of 0.3 * loss")


def test_simple_loss_module(ua, golden):
    g = golden("ops_small")
    lg = torch.from_numpy(g["loss_logits"]).to(DEV).requires_grad_(True)
    tg = torch.from_numpy(g["loss_target"]).to(DEV)
    loss = ua.SimpleLoss()(lg, tg)
    assert loss.dim() == 0 and loss.dtype == torch.float32
    loss.backward()
    assert abs(loss.item() - float(g["loss_value"])) <= 2e-6 * abs(float(g["loss_value"]))
    check(lg.grad.cpu(), torch.from_numpy(g["loss_dlogits"]), 2e-5, "module dlogits")


# --------------------------------------------------------------------------- SGD
def test_sgd_golden(ua, golden):
    g = golden("ops_small")
    p = torch.from_numpy(g["sgd_p0"]).to(DEV)
    buf = torch.zeros_like(p)
    for s in range(3):
        gr = torch.from_numpy(g["sgd_grads"][s]).to(DEV)
        ua.ops.sgd_nesterov_step(p, gr, buf, 0.005, 0.99, 1e-4, s == 0)
        check(p.cpu(), torch.from_numpy(g["sgd_traj"][s]), 1e-6, f"sgd step {s}")


def test_sgd_odd_length(ua):
    n = 1003
    p0, g0 = rnd(n, seed=30), rnd(n, seed=31)
    ps, bufs = [p0.clone()], [None]
    O.sgd_nesterov_(ps, [g0], bufs)
    O.sgd_nesterov_(ps, [g0 * 0.5], bufs)
    pd = torch.zeros(1004, device=DEV)[:n]
    pd.copy_(p0)
    bd = torch.zeros(1004, device=DEV)[:n]
    gd = torch.zeros(1004, device=DEV)[:n]
    gd.copy_(g0)
    ua.ops.sgd_nesterov_step(pd, gd, bd, 0.005, 0.99, 1e-4, True)
    gd.mul_(0.5)
    ua.ops.sgd_nesterov_step(pd, gd, bd, 0.005, 0.99, 1e-4, False)
    check(pd.cpu(), ps[0], 1e-6, "sgd odd")


# --------------------------------------------------------------------------- composed blocks (golden)
def _run_block(ua, x0, x1, p, idx, stride, masks, grads_out=None):
    """conv -> IN stats -> apply for the two convs of a reference ConvBlock state_dict `p`."""
    recs = []
    cur0, cur1 = x0, x1
    for k, (ci, ni) in enumerate(idx):
        w = p[f"block.{ci}.weight"].to(DEV)
        wf, wd = ua.ops.pack_conv3x3_weights(w)
        y = ua.ops.conv3x3_fwd(cur0, cur1, wf, p[f"block.{ci}.bias"].to(DEV), stride if k == 0 else 1)
        gm, bt = p[f"block.{ni}.weight"].to(DEV), p[f"block.{ni}.bias"].to(DEV)
        st = ua.ops.instnorm_stats(y, gm, bt, 1e-5)
        m = masks[k].to(DEV) if masks is not None else None
        a = ua.ops.instnorm_lrelu_drop_fwd(y, st[2], st[3], m, 0.01)
        recs.append(dict(x0=cur0, x1=cur1, y=y, st=st, m=m, a=a, wd=wd, gm=gm, bt=bt, ci=ci, ni=ni,
                         stride=stride if k == 0 else 1, w=w))
        cur0, cur1 = a, None
    return recs


def _block_backward(ua, recs, g):
    grads = {}
    dx1 = None
    for r in reversed(recs):
        C = r["y"].shape[3]
        dg, dbt, dbias = (torch.empty(C, device=DEV) for _ in range(3))
        dy = ua.ops.instnorm_lrelu_drop_bwd(g, r["y"], r["st"][0], r["st"][1], r["gm"], r["bt"],
                                            r["m"], 0.01, dg, dbt, dbias)
        dw = torch.empty_like(r["w"])
        ua.ops.conv3x3_bwd_weight(r["x0"], dy, dw, 0, r["stride"])
        N, H, W, C0 = r["x0"].shape
        if r["x1"] is not None:
            ua.ops.conv3x3_bwd_weight(r["x1"], dy, dw, C0, r["stride"])
            dx1 = ua.ops.conv3x3_bwd_data(dy, r["wd"], C0, r["x1"].shape[3], H, W, r["stride"])
        g = ua.ops.conv3x3_bwd_data(dy, r["wd"], 0, C0, H, W, r["stride"])
        grads[f"block.{r['ci']}.weight"] = dw
        grads[f"block.{r['ci']}.bias"] = dbias
        grads[f"block.{r['ni']}.weight"] = dg
        grads[f"block.{r['ni']}.bias"] = dbt
    return g, dx1, grads


def test_convblock_golden(ua, golden):
    """Reference ConvBlock(32->64, stride 2, dropout 0.2) in train mode, forward + backward."""
    g = golden("ops_small")
    p = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("cb_p_")}
    masks = [torch.from_numpy(g["cb_mask0"]), torch.from_numpy(g["cb_mask1"])]
    recs = _run_block(ua, to_nhwc(torch.from_numpy(g["cb_x"])), None, p, [(0, 1), (4, 5)], 2, masks)
    check(from_nhwc(recs[-1]["a"]), torch.from_numpy(g["cb_y"]), 2e-5, "ConvBlock fwd")
    gx, _, grads = _block_backward(ua, recs, to_nhwc(torch.from_numpy(g["cb_gy"])))
    check(from_nhwc(gx), torch.from_numpy(g["cb_gx"]), 5e-5, "ConvBlock gx")
    for k, v in grads.items():
        ref = torch.from_numpy(g["cb_g_" + k])
        if k.endswith("bias") and ref.abs().max() < 1e-4:   # conv bias under IN: ~0 +- rounding
            assert (v.cpu() - ref).abs().max() < 1e-4
        else:
            check(v.cpu(), ref, 5e-5, f"ConvBlock grad {k}")


def test_upblock_golden(ua, golden):
    """Reference UpBlock(64 up + 32 skip -> 32), eval mode, forward + backward."""
    g = golden("ops_small")
    p = {k[16:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ub_p_conv_block.")}
    up = ua.ops.upsample2x_fwd(to_nhwc(torch.from_numpy(g["ub_x"])))
    recs = _run_block(ua, up, to_nhwc(torch.from_numpy(g["ub_skip"])), p, [(0, 1), (3, 4)], 1, None)
    check(from_nhwc(recs[-1]["a"]), torch.from_numpy(g["ub_y"]), 2e-5, "UpBlock fwd")
    g_up, g_skip, grads = _block_backward(ua, recs, to_nhwc(torch.from_numpy(g["ub_gy"])))
    gx = ua.ops.upsample2x_bwd(g_up)
    check(from_nhwc(gx), torch.from_numpy(g["ub_gx"]), 5e-5, "UpBlock gx")
    check(from_nhwc(g_skip), torch.from_numpy(g["ub_gskip"]), 5e-5, "UpBlock gskip")
    for k, v in grads.items():
        ref = torch.from_numpy(g["ub_g_conv_block." + k])
        if k.endswith("bias") and ref.abs().max() < 1e-4:
            assert (v.cpu() - ref).abs().max() < 1e-4
        else:
            check(v.cpu(), ref, 5e-5, f"UpBlock grad {k}")


def test_convblock_and_upblock_modules_run_stand_alone(ua, golden):
    """`ConvBlock.forward(x)` / `UpBlock.forward(x, skip)` called directly, as the reference's
    are (Our_UNet/models/unet.py:136-141, :203-231): NCHW in / out, differentiable through the
    HIP entry points - against the reference fixtures of test_convblock_golden /
    test_upblock_golden (train mode with the recorded dropout draws; eval mode)."""
    g = golden("ops_small")
    blk = ua.ConvBlock(32, 64, [3, 3], [2, 2], n_convs=2, spatial_dropout_rate=0.2)
    blk.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("cb_p_")})
    blk = blk.to(DEV).train()
    blk.dropout_mask_override = [torch.from_numpy(g["cb_mask0"]), torch.from_numpy(g["cb_mask1"])]
    x = torch.from_numpy(g["cb_x"]).to(DEV).requires_grad_(True)
    y = blk(x)
    check(y.detach().cpu(), torch.from_numpy(g["cb_y"]), 2e-5, "ConvBlock module fwd")
    y.backward(torch.from_numpy(g["cb_gy"]).to(DEV))
    check(x.grad.cpu(), torch.from_numpy(g["cb_gx"]), 5e-5, "ConvBlock module gx")
    for k, p in blk.named_parameters():
        ref = torch.from_numpy(g["cb_g_" + k])
        if k.endswith("bias") and ref.abs().max() < 1e-4:
            assert (p.grad.cpu() - ref).abs().max() < 1e-4
        else:
            check(p.grad.cpu(), ref, 5e-5, f"ConvBlock module grad {k}")
    # without an override, train mode draws its own masks (values 0 or 1/(1-p) per (n, c))
    blk.dropout_mask_override = None
    y2 = blk(x.detach())
    assert y2.shape == y.shape and not torch.equal(y2, y.detach())

    up = ua.UpBlock(64, 32, 32, [3, 3], n_convs=2)
    up.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ub_p_")})
    up = up.to(DEV).eval()
    xl = torch.from_numpy(g["ub_x"]).to(DEV).requires_grad_(True)
    sk = torch.from_numpy(g["ub_skip"]).to(DEV).requires_grad_(True)
    yu = up(xl, sk)
    check(yu.detach().cpu(), torch.from_numpy(g["ub_y"]), 2e-5, "UpBlock module fwd")
    yu.backward(torch.from_numpy(g["ub_gy"]).to(DEV))
    check(xl.grad.cpu(), torch.from_numpy(g["ub_gx"]), 5e-5, "UpBlock module gx")
    check(sk.grad.cpu(), torch.from_numpy(g["ub_gskip"]), 5e-5, "UpBlock module gskip")
    for k, p in up.named_parameters():
        ref = torch.from_numpy(g["ub_g_" + k])
        if k.endswith("bias") and ref.abs().max() < 1e-4:
            assert (p.grad.cpu() - ref).abs().max() < 1e-4
        else:
            check(p.grad.cpu(), ref, 5e-5, f"UpBlock module grad {k}")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        blk(torch.zeros(1, 32, 8, 8))


# ---------------------------------------------------------------- split-bf16 ("bf16x3") mode
# fp32 operands split into 3 bf16 terms, 6 products, fp32 accumulation: must be as accurate as
# the fp32 matrix-core kernels.  Both are compared with an fp64 convolution of the same fp32
# inputs; the split path may not be worse than 2x the fp32 path's error, with a floor of 2.5e-6
# (the fp32 kernels of the deep layers sum K in four groups, which lowers THEIR error below what
# a serial fp32 accumulation - the split path's - gives at K = 3456).
def _x3_ok(e_x3, e_32, what):
    assert e_x3 <= max(2.0 * e_32, 2.5e-6), f"{what}: bf16x3 {e_x3:.2e} vs fp32-MFMA {e_32:.2e}"
    assert e_x3 <= 5e-6, f"{what}: bf16x3 error {e_x3:.2e} is not fp32-class"


X3_FWD_SHAPES = BF16_SHAPES + [(1, 32, 32, 256, 128, 128, 1), (2, 34, 18, 32, 0, 32, 2),
                               (2, 8, 64, 32, 0, 32, 1), (1, 4, 32, 64, 32, 64, 1),
                               (3, 12, 96, 128, 0, 256, 1), (2, 64, 64, 32, 32, 32, 1),
                               (8, 64, 64, 64, 0, 64, 1), (8, 64, 64, 32, 0, 32, 1)]


@pytest.mark.parametrize("shape", X3_FWD_SHAPES)
def test_conv3x3_fwd_bf16x3_is_fp32_accurate(ua, shape):
    N, H, W, C0, C1, Cout, s = shape
    x = rnd(N, C0 + C1, H, W, seed=3)
    w = rnd(Cout, C0 + C1, 3, 3, seed=4, scale=0.1)
    b = rnd(Cout, seed=5)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=s, padding=1)
    wf, _ = ua.ops.pack_conv3x3_weights(w.to(DEV))
    wf3, _ = ua.ops.pack_conv3x3_weights_bf16x3(w.to(DEV), want_wd=False)
    planes = wf3.float().sum(0).cpu()          # h + m + l reproduces the fp32 weight
    assert relerr(planes, wf.cpu()) <= 2.0 ** -24
    x0 = to_nhwc(x[:, :C0])
    x1 = to_nhwc(x[:, C0:]) if C1 else None
    y3 = ua.ops.conv3x3_fwd(x0, x1, wf, b.to(DEV), s, bf16="bf16x3", wf3=wf3)
    y32 = ua.ops.conv3x3_fwd(x0, x1, wf, b.to(DEV), s)
    _x3_ok(relerr(from_nhwc(y3), ref), relerr(from_nhwc(y32), ref), f"fwd {shape}")


@pytest.mark.parametrize("shape", [(2, 12, 20, 32, 32, 1, (0, 32)), (1, 16, 16, 96, 64, 1, (32, 64)),
                                   (2, 16, 24, 32, 64, 2, (0, 32)), (2, 4, 4, 512, 512, 2, (0, 512)),
                                   (1, 64, 64, 64, 64, 1, (0, 64)), (1, 32, 32, 256, 256, 1, (128, 128)),
                                   (2, 8, 64, 96, 32, 1, (32, 64)), (1, 4, 32, 32, 128, 1, (0, 32)),
                                   (8, 64, 64, 64, 32, 1, (0, 64)), (8, 64, 64, 32, 64, 1, (0, 32))])
def test_conv3x3_bwd_data_bf16x3_is_fp32_accurate(ua, shape):
    N, H, W, Cin, Cout, s, (off, cc) = shape
    x = rnd(N, Cin, H, W, seed=6).double().requires_grad_(True)
    w = rnd(Cout, Cin, 3, 3, seed=7, scale=0.1)
    y = F.conv2d(x, w.double(), None, stride=s, padding=1)
    gy = rnd(*y.shape, seed=8)
    (gx,) = torch.autograd.grad(y, x, gy.double())
    _, wd = ua.ops.pack_conv3x3_weights(w.to(DEV))
    _, wd3 = ua.ops.pack_conv3x3_weights_bf16x3(w.to(DEV), want_wf=False)
    d3 = ua.ops.conv3x3_bwd_data(to_nhwc(gy), wd, off, cc, H, W, s, bf16="bf16x3", wd3=wd3)
    d32 = ua.ops.conv3x3_bwd_data(to_nhwc(gy), wd, off, cc, H, W, s)
    ref = gx[:, off:off + cc]
    _x3_ok(relerr(from_nhwc(d3), ref), relerr(from_nhwc(d32), ref), f"dgrad {shape}")


@pytest.mark.parametrize("shape", [(2, 12, 64, 32, 0, 32, 32, 1), (1, 16, 16, 64, 0, 64, 64, 1),
                                   (2, 8, 8, 128, 0, 128, 128, 1), (1, 32, 32, 64, 32, 96, 64, 1),
                                   (1, 64, 128, 32, 0, 32, 32, 1), (1, 32, 64, 32, 0, 32, 64, 1),
                                   (2, 16, 24, 32, 0, 32, 64, 2), (3, 34, 70, 32, 0, 32, 32, 1),
                                   (2, 64, 64, 128, 0, 128, 64, 1)])
def test_conv3x3_bwd_weight_bf16x3_is_fp32_accurate(ua, shape):
    N, H, W, Cx, off, Ct, Cout, s = shape
    x = rnd(N, Ct, H, W, seed=10)
    w = rnd(Cout, Ct, 3, 3, seed=11, scale=0.1).double().requires_grad_(True)
    y = F.conv2d(x.double(), w, None, stride=s, padding=1)
    gy = rnd(*y.shape, seed=12)
    (gw,) = torch.autograd.grad(y, w, gy.double())
    ref = gw[:, off:off + Cx]
    xs, gys = to_nhwc(x[:, off:off + Cx]), to_nhwc(gy)
    dw3 = torch.zeros((Cout, Ct, 3, 3), device=DEV)
    dw32 = torch.zeros((Cout, Ct, 3, 3), device=DEV)
    db3 = torch.zeros(Cout, device=DEV)
    ua.ops.conv3x3_bwd_weight(xs, gys, dw3, off, s, db=db3, bf16="bf16x3")
    ua.ops.conv3x3_bwd_weight(xs, gys, dw32, off, s)
    _x3_ok(relerr(dw3[:, off:off + Cx], ref), relerr(dw32[:, off:off + Cx], ref), f"wgrad {shape}")
    check(db3, gy.sum(dim=(0, 2, 3)), 2e-5, "bias gradient")


# ---------------------------------------------------------------- validation metrics / input
@pytest.mark.parametrize("n,h,w", [(1, 64, 64), (3, 96, 160), (8, 512, 512)])
def test_argmax_dice_counts_exact(ua, n, h, w):
    g = torch.Generator().manual_seed(n * h + w)
    logits = torch.randn(n, 3, h, w, generator=g)
    logits[:, :, : h // 4] = logits[:, :1, : h // 4]          # three-way ties -> class 0
    logits[:, 2, h // 4: h // 2] = logits[:, 1, h // 4: h // 2]  # 1/2 ties -> class 1 when largest
    target = torch.randint(0, 3, (n, h, w), generator=g)
    target[torch.rand(n, h, w, generator=g) < 0.1] = 255
    preds, counts = ua.ops.argmax_dice_counts(logits.cuda(), target.cuda())
    ref_p = logits.argmax(dim=1)
    assert torch.equal(preds.cpu().long(), ref_p)
    valid = target != 255
    for c in range(3):
        pc, mc = (ref_p == c) & valid, (target == c) & valid
        assert counts[c].tolist() == [int((pc & mc).sum()), int(pc.sum()), int(mc.sum())]


def test_segmentation_metrics_golden(ua, golden):
    """f2 PINNED: tests/golden/metrics.npz holds the accumulators the REFERENCE's
    SegmentationMetrics (Our_UNet/utils/metrics.py:59-91) produced for argmax predictions of the
    stored logits - exact ties, a class absent from the labels, a class never predicted, an
    all-ignored image, a 255 ring.  `unet_argmax_dice_counts` must reproduce the predictions and
    intersections / unions / TP / FP / FN bit-exactly (integers), per batch and accumulated, and
    the drop-in `SegmentationMetrics` the derived IoU / Dice / precision / recall / means as the
    identical float64 quotients (nan where the reference returns nan)."""
    g = golden("metrics")
    fields = ("intersections", "unions", "true_positives", "false_positives", "false_negatives")
    acc = ua.SegmentationMetrics(num_classes=3, ignore_index=255)
    acc_maps = ua.SegmentationMetrics(num_classes=3, ignore_index=255)

    def same(m, tag):
        for f in fields:
            assert np.array_equal(getattr(m, f), g[f"{tag}_{f}"]), (tag, f)
        assert m.total_pixels == int(g[f"{tag}_total_pixels"])
        assert m.correct_pixels == int(g[f"{tag}_correct_pixels"])
        for f, fn in (("iou", m.compute_iou), ("dice", m.compute_dice),
                      ("precision", m.compute_precision), ("recall", m.compute_recall)):
            got = np.array([fn(c) for c in range(3)])
            assert np.array_equal(got, g[f"{tag}_{f}"], equal_nan=True), (tag, f, got)
        for f, v in (("pixel_accuracy", m.compute_pixel_accuracy()),
                     ("mean_iou", m.compute_mean_iou()), ("mean_dice", m.compute_mean_dice())):
            ref = float(g[f"{tag}_{f}"])
            assert v == ref or (np.isnan(v) and np.isnan(ref)), (tag, f, v, ref)

    for k in range(int(g["n_batches"])):
        lg = torch.from_numpy(g[f"b{k}_logits"]).cuda()
        t = torch.from_numpy(g[f"b{k}_target"]).cuda()
        preds, counts = ua.ops.argmax_dice_counts(lg, t)
        assert np.array_equal(preds.cpu().numpy(), g[f"b{k}_pred"])
        c = counts.cpu().numpy().astype(np.float64)
        assert np.array_equal(c[:, 0], g[f"b{k}_intersections"])
        assert np.array_equal(c[:, 1] + c[:, 2] - c[:, 0], g[f"b{k}_unions"])
        assert np.array_equal(c[:, 1] - c[:, 0], g[f"b{k}_false_positives"])
        assert np.array_equal(c[:, 2] - c[:, 0], g[f"b{k}_false_negatives"])
        one = ua.SegmentationMetrics()
        one.update_from_logits(lg, t)
        same(one, f"b{k}")
        acc.update_from_logits(lg, t)
        acc_maps.update(preds, t)               # the reference's update(pred, target) form
    same(acc, "acc")
    same(acc_maps, "acc")
    d = acc.get_all_metrics()
    assert set(d) == {"pixel_accuracy", "mean_iou", "mean_dice", "class_metrics"}
    assert set(d["class_metrics"]["class_1"]) == {"iou", "dice", "precision", "recall", "f1_score"}
    acc.reset()
    assert acc.total_pixels == 0 and np.isnan(acc.compute_mean_iou())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        acc.update_from_logits(torch.zeros(1, 3, 8, 8), torch.zeros(1, 8, 8, dtype=torch.int64))


def test_predict_masks_is_argmax_of_eval_forward(ua):
    from oracle import unet_ref as O
    model = ua.create_model().train()
    model.load_state_dict(O.fill_state_dict(9, trained_like=True))
    img, _ = O.synthetic_batch(3, 2, 64, 64)
    preds = ua.predict_masks(model, img.cuda())
    assert model.training and preds.dtype == torch.uint8 and preds.shape == (2, 64, 64)
    model.eval()
    with torch.no_grad():
        ref = model(img.cuda()).argmax(dim=1)
    assert torch.equal(preds.long(), ref)


def test_argmax_dice_counts_absent_class_and_all_ignored(ua):
    logits = torch.zeros(2, 3, 64, 64)
    logits[:, 1] = 1.0
    target = torch.full((2, 64, 64), 255, dtype=torch.int64)
    _, counts = ua.ops.argmax_dice_counts(logits.cuda(), target.cuda(), want_preds=False)
    assert counts.sum().item() == 0
    target[0] = 1
    _, counts = ua.ops.argmax_dice_counts(logits.cuda(), target.cuda(), want_preds=False)
    assert counts.tolist() == [[0, 0, 0], [4096, 4096, 4096], [0, 0, 0]]


def test_validate_matches_reference_arithmetic(ua):
    from oracle import unet_ref as O
    model = ua.create_model()
    model.load_state_dict(O.fill_state_dict(3, trained_like=True))
    loss_fn = ua.get_loss_function()
    batches = []
    for s in range(3):
        img, tgt = O.synthetic_batch(20 + s, 2, 64, 64)
        if s == 1:
            tgt[tgt == 2] = 0          # a batch without dogs: the 1.0 branch
        batches.append({"image": img, "mask": tgt})
    val_loss, scores = ua.validate(model, batches, loss_fn, "cuda")
    model.eval()
    with torch.no_grad():
        outs = [model(b["image"].cuda()).cpu() for b in batches]
        ref_loss = sum(O.simple_loss(o, b["mask"]).item() for o, b in zip(outs, batches)) / 3
    ref = O.validate_scores(outs, [b["mask"] for b in batches])
    assert abs(val_loss - ref_loss) <= 1e-5 * abs(ref_loss)
    for k in ref:
        assert abs(scores[k] - ref[k]) <= 1e-9, k        # integer counts -> same quotient


@pytest.mark.parametrize("n,h,w", [(1, 64, 64), (2, 75, 131), (8, 512, 512)])
def test_preprocess_u8_bit_exact(ua, n, h, w):
    from oracle import unet_ref as O
    rng = np.random.default_rng(h)
    img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    img[0, 0, :256 if w >= 256 else w, 0] = np.arange(min(w, 256), dtype=np.uint8)
    mask = rng.choice(np.array([0, 1, 2, 3, 7, 254, 255], dtype=np.uint8), (n, h, w))
    out, tgt = ua.ops.preprocess_u8(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda())
    for i in range(n):
        ri, rm = O.preprocess_sample(img[i], mask[i])
        assert torch.equal(out[i].permute(2, 0, 1).cpu(), ri)
        assert torch.equal(tgt[i].cpu(), rm)
    out2, none = ua.ops.preprocess_u8(torch.from_numpy(img).cuda())
    assert none is None and torch.equal(out2, out)


def test_forward_accepts_preprocessed_nhwc(ua):
    from oracle import unet_ref as O
    model = ua.create_model().eval()
    model.load_state_dict(O.fill_state_dict(5, trained_like=True))
    rng = np.random.default_rng(0)
    img = torch.from_numpy(rng.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)).cuda()
    x_nhwc, _ = ua.ops.preprocess_u8(img)
    with torch.no_grad():
        a = model(x_nhwc, input_layout="nhwc")
        b = model(x_nhwc.permute(0, 3, 1, 2).contiguous())
    assert torch.equal(a, b)


@pytest.mark.parametrize("dynamic", [True, False])
def test_sharded_loss_equals_loss_of_concatenated_batch(ua, dynamic):
    """shard_stats -> sum -> shard_apply on 3 shards == SimpleLoss of the whole batch."""
    g = torch.Generator().manual_seed(11)
    shards = []
    for s, n in enumerate([2, 2, 2]):
        lg = torch.randn(n, 3, 64, 96, generator=g) * 2
        tg = torch.randint(0, 3, (n, 64, 96), generator=g)
        tg[torch.rand(n, 64, 96, generator=g) < 0.15] = 255
        if s == 1:
            tg[tg == 2] = 0                     # a shard with no class-2 pixel
        shards.append((lg, tg))
    cw = None if dynamic else torch.tensor([0.5, 1.25, 1.25])
    all_lg = torch.cat([s[0] for s in shards]).requires_grad_(True)
    all_tg = torch.cat([s[1] for s in shards])
    ref = O.simple_loss(all_lg, all_tg, dynamic_weights=dynamic, fixed_weights=cw)
    ref.backward()
    dev = [(lg.cuda(), tg.cuda()) for lg, tg in shards]
    phase1 = [ua.ops.dice_wce_loss_shard_stats(lg, tg, 1e-5, 255) for lg, tg in dev]
    gstats = sum(p[0] for p in phase1)
    n0 = 0
    for (lg, tg), (_, ws) in zip(dev, phase1):
        out, dl = ua.ops.dice_wce_loss_shard_apply(lg, tg, gstats, 6, ws, 1e-5, 1.0, 1.0, 255,
                                                   dynamic, class_weights=None if cw is None else cw.cuda())
        assert abs(out[0].item() - ref.item()) <= 2e-6 * abs(ref.item())
        rg = all_lg.grad[n0:n0 + lg.shape[0]]
        assert ((dl.cpu() - rg).abs().max() / rg.abs().max()).item() <= 2e-5
        n0 += lg.shape[0]


# ---------------------------------------------------------------- bench-size tiles
# The widest tile instantiations (128 output columns) are only selected when a launch has
# >= 512 tiles, i.e. at the bench's batch: check them on one real layer (enc2.4 at bs 8:
# 128 -> 128 channels, 128 x 128) against a CPU fp32 convolution, forward and data gradient,
# in the fp32 and the split-bf16 operand modes.
@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_conv3x3_bench_size_layer(ua, mode):
    N, C, H = 8, 128, 128
    x = rnd(N, C, H, H, seed=21).requires_grad_(True)
    w = rnd(C, C, 3, 3, seed=22, scale=0.03)
    b = rnd(C, seed=23)
    y = F.conv2d(x, w, b, padding=1)
    gy = rnd(N, C, H, H, seed=24)
    (gx,) = torch.autograd.grad(y, x, gy)
    wf, wd = ua.ops.pack_conv3x3_weights(w.to(DEV))
    kw_f, kw_d = {}, {}
    if mode == "bf16x3":
        wf3, wd3 = ua.ops.pack_conv3x3_weights_bf16x3(w.to(DEV))
        kw_f, kw_d = {"wf3": wf3}, {"wd3": wd3}
    got = ua.ops.conv3x3_fwd(to_nhwc(x.detach()), None, wf, b.to(DEV), 1, bf16=mode, **kw_f)
    check(from_nhwc(got), y.detach(), 2e-5, f"{mode} fwd 8x128x128x128")
    dx = ua.ops.conv3x3_bwd_data(to_nhwc(gy), wd, 0, C, H, H, 1, bf16=mode, **kw_d)
    check(from_nhwc(dx), gx, 2e-5, f"{mode} dgrad 8x128x128x128")
    acc = torch.ones(N, H, H, C, device=DEV)          # accumulate flag of the same tile shape
    ua.ops.conv3x3_bwd_data(to_nhwc(gy), wd, 0, C, H, H, 1, out=acc, accumulate=True, bf16=mode,
                            **kw_d)
    check(from_nhwc(acc), gx + 1.0, 2e-5, f"{mode} dgrad accumulate")


@pytest.mark.parametrize("case", [(2, 3, 64, 64, 128, 128), (1, 3, 48, 80, 96, 200),
                                  (2, 5, 20, 12, 16, 16), (1, 4, 33, 17, 16, 40)])
def test_resize_bilinear_matches_torch_interpolate(ua, case):
    """unet_resize_bilinear_fwd / _bwd (the logits resize of SimpleLoss, losses.py:66-68, and the
    CLIP feature resize, CLIP_UNet/models/unet.py:444-450) against F.interpolate(bilinear,
    align_corners=False) and its autograd, up- and down-scaling, non-integer ratios."""
    import torch.nn.functional as F
    N, C, h, w, H, W = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, h, w, generator=g)
    gy = torch.randn(N, C, H, W, generator=g)
    # (the reference is the fp32 CPU op: source indices and weights are computed in fp32 there
    # too; an fp64 evaluation differs from either by the rounding of the weights, ~1e-5)
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(H, W), mode="bilinear", align_corners=False)
    yr.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    y = ua.ops.resize_bilinear(xd, (H, W))
    y.backward(gy.to(DEV))
    assert (y.detach().cpu() - yr.detach()).abs().max() <= 2e-6 * yr.abs().max()
    assert (xd.grad.cpu() - xr.grad).abs().max() <= 5e-6 * xr.grad.abs().max()
