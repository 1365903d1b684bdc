"""Fused layer pipeline (-m gpu): the C-ABI entry points `unet_conv_in_fwd`,
`unet_conv_in_bwd_weight`, `unet_upsample2x_in_fwd`, `unet_head1x1_in_fwd/_bwd` against stock
torch CPU ops on seeded inputs, and against the reference's ConvBlock / UpBlock fixtures
(tests/golden/ops_small.npz).  The shapes are chosen so that every kernel instantiation behind
the entry points (patch-staged 128/64/32 columns, row-fused with and without resident weights,
gather-GEMM tiles, RGB stem rows / gather, 1x1) runs with activation-on-load and its
statistics epilogue (or the stand-alone statistics fallback).

Reference lines: Our_UNet/models/unet.py:101-134 (conv -> InstanceNorm2d -> LeakyReLU ->
SpatialDropout2d), :215-231 (interpolate + cat), :374-381 (head).
Tolerances are relative to the tensor's max magnitude (fp32, different summation order)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
SLOPE = 0.01


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


def coeffs(n, c, seed, dropped=True):
    """Folded coefficients as a producing layer would leave them: alpha > 0 mostly, some
    channels dropped (alpha = beta = 0), some negative scales."""
    al = rnd(n, c, seed=seed) * 0.5 + 1.0
    be = rnd(n, c, seed=seed + 1) * 0.7
    if dropped:
        drop = torch.rand(n, c, generator=torch.Generator().manual_seed(seed + 2)) < 0.15
        al = torch.where(drop, torch.zeros_like(al), al)
        be = torch.where(drop, torch.zeros_like(be), be)
    return al, be


def act_ref(x_nchw, al, be):
    """a = lrelu(x * alpha[n, c] + beta[n, c]) on an NCHW tensor (fp64)."""
    z = x_nchw.double() * al.double()[:, :, None, None] + be.double()[:, :, None, None]
    return F.leaky_relu(z, SLOPE)


def make_src(ua, x_nchw, coef):
    if coef is None:
        return ua.ops.Act(to_nhwc(x_nchw))
    return ua.ops.Act(to_nhwc(x_nchw), coef[0].to(DEV).contiguous(), coef[1].to(DEV).contiguous())


# (N, H, W, C0, C1, Cout, stride, ksize, act0, act1): which kernel it reaches is in the comment
FWD_CASES = [
    (1, 256, 256, 32, 32, 128, 1, 3, True, True),    # patch-staged, 128 columns (two sources)
    (1, 256, 256, 64, 0, 64, 1, 3, True, False),     # patch-staged, 64 columns
    (2, 256, 256, 32, 32, 32, 1, 3, False, True),    # patch-staged, 32 columns x 8 rows; plain src0
    (1, 128, 128, 32, 0, 32, 1, 3, True, False),     # 32 -> 32 channel kernel (weights in registers)
    (2, 256, 256, 32, 0, 32, 1, 3, True, False),     # the same, two tiles per persistent workgroup
    (3, 64, 96, 32, 0, 32, 1, 3, False, False),      # the same, 72 tiles (plain walk), plain source
    (1, 12, 128, 32, 0, 32, 1, 3, True, False),      # row-fused, weights resident (H % 8 != 0)
    (1, 12, 128, 64, 0, 32, 1, 3, True, False),      # row-fused, streamed weights
    (2, 256, 256, 32, 0, 128, 2, 3, True, False),    # stride 2, patch-staged 64 columns
    (2, 264, 512, 32, 0, 128, 2, 3, True, False),    # stride 2, patch-staged 128 columns, H != W
    (2, 256, 256, 32, 32, 256, 2, 3, False, True),   # stride 2, patch-staged, plain + activated source
    (2, 256, 256, 32, 0, 64, 2, 3, True, False),     # gather-GEMM 128x64 tiles, stride 2
    (2, 16, 16, 64, 0, 64, 1, 3, True, False),       # gather-GEMM 64x64, two K groups
    (2, 16, 16, 128, 0, 128, 1, 3, True, False),     # deep layer: four K groups per block
    (2, 16, 16, 64, 64, 64, 1, 3, True, True),       # four K groups, a group boundary between sources
    (2, 32, 32, 128, 0, 128, 2, 3, True, False),     # deep stride-2 layer, four K groups
    (3, 4, 4, 32, 32, 32, 1, 3, True, True),         # tiles span images: per-row coefficients
    (2, 2, 2, 64, 0, 64, 2, 3, True, False),         # 2x2 -> 1x1 grid, stand-alone statistics
    (2, 12, 20, 32, 0, 32, 1, 3, True, False),       # ragged grid (240 positions)
    (2, 16, 16, 64, 32, 64, 1, 1, True, False),      # 1x1 (CLIP fusion layer), plain second source
    (2, 8, 128, 3, 0, 32, 1, 3, False, False),       # RGB stem, row form + statistics epilogue
    (2, 12, 20, 3, 0, 32, 1, 3, False, False),       # RGB stem, gather form
]


@pytest.mark.parametrize("case", FWD_CASES)
@pytest.mark.parametrize("with_mask", [False, True])
def test_conv_in_fwd(ua, case, with_mask):
    N, H, W, C0, C1, Cout, stride, ks, act0, act1 = case
    x0 = rnd(N, C0, H, W, seed=1)
    x1 = rnd(N, C1, H, W, seed=2) if C1 else None
    c0 = coeffs(N, C0, 10) if act0 else None
    c1 = coeffs(N, C1, 20) if (act1 and C1) else None
    w = rnd(Cout, C0 + C1, ks, ks, seed=3, scale=(2.0 / (ks * ks * (C0 + C1))) ** 0.5)
    b = rnd(Cout, seed=4, scale=0.3)
    gamma = rnd(Cout, seed=5) * 0.2 + 1.0
    beta = rnd(Cout, seed=6) * 0.2
    mask = None
    if with_mask:
        keep = torch.rand(N, Cout, generator=torch.Generator().manual_seed(7)) < 0.7
        mask = keep.float() / 0.7
    # reference (fp64)
    a0 = act_ref(x0, *c0) if c0 else x0.double()
    parts = [a0]
    if C1:
        parts.append(act_ref(x1, *c1) if c1 else x1.double())
    y_ref = F.conv2d(torch.cat(parts, 1), w.double(), b.double(), stride=stride, padding=ks // 2)
    mean_ref = y_ref.mean(dim=(2, 3))
    var_ref = y_ref.var(dim=(2, 3), unbiased=False)
    rstd_ref = 1.0 / torch.sqrt(var_ref + 1e-5)
    mk = mask.double() if mask is not None else torch.ones(N, Cout, dtype=torch.double)
    alpha_ref = gamma.double()[None] * rstd_ref * mk
    beta_ref = (beta.double()[None] - mean_ref * gamma.double()[None] * rstd_ref) * mk
    # HIP
    if ks == 3:
        wk, _ = ua.ops.pack_conv3x3_weights(w.to(DEV), want_wd=False)
    else:
        wk = w.view(Cout, C0 + C1).to(DEV).contiguous()
    s0 = make_src(ua, x0, c0)
    s1 = make_src(ua, x1, c1) if C1 else None
    y, st = ua.ops.conv_in_fwd(s0, s1, SLOPE, wk, b.to(DEV), ks, stride, gamma.to(DEV),
                               beta.to(DEV), 1e-5, None if mask is None else mask.to(DEV))
    check(from_nhwc(y), y_ref, 2e-5, "y")
    assert (st[0].cpu().double() - mean_ref).abs().max() <= 2e-5 * (y_ref.abs().max() + 1)
    check(st[1].cpu(), rstd_ref, 5e-5, "rstd")
    check(st[2].cpu(), alpha_ref, 5e-5, "alpha")
    # beta is a difference of O(1) terms: absolute scale of |beta| + |mean * alpha|
    scale = (beta.abs().max() + (mean_ref * gamma[None] * rstd_ref).abs().max()).item() / 0.7
    assert (st[3].cpu().double() - beta_ref).abs().max() <= 5e-5 * scale


WINO_CASES = [  # (N, H, W, C0, C1, Cout, act0, act1): shapes the Winograd kernel tiles
    (4, 128, 128, 64, 0, 64, True, False),     # one column tile, 8 chunks
    (2, 128, 128, 128, 0, 128, True, False),   # two column tiles
    (4, 64, 64, 256, 0, 256, True, False),     # the 256-channel layer shape
    (2, 128, 256, 32, 32, 64, True, True),     # two sources (virtual concat), H != W
    (8, 32, 32, 64, 0, 512, True, False),      # image = one tile column
]


@pytest.mark.parametrize("case", WINO_CASES)
@pytest.mark.parametrize("with_mask", [False, True])
def test_conv_in_fwd_winograd(ua, case, with_mask):
    """The Winograd F(2x2,3x3) form of the fused forward (csrc/conv_wino.hip) against the fp64
    convolution of the activated sources - same tolerances as the direct kernels - and against
    the direct kernel behind the same entry point."""
    N, H, W, C0, C1, Cout, act0, act1 = case
    assert ua.ops.conv_wino_supported(N, H, W, C0, C1, Cout)
    x0 = rnd(N, C0, H, W, seed=1)
    x1 = rnd(N, C1, H, W, seed=2) if C1 else None
    c0 = coeffs(N, C0, 10) if act0 else None
    c1 = coeffs(N, C1, 20) if (act1 and C1) else None
    w = rnd(Cout, C0 + C1, 3, 3, seed=3, scale=(2.0 / (9 * (C0 + C1))) ** 0.5)
    b = rnd(Cout, seed=4, scale=0.3)
    gamma = rnd(Cout, seed=5) * 0.2 + 1.0
    beta = rnd(Cout, seed=6) * 0.2
    mask = None
    if with_mask:
        keep = torch.rand(N, Cout, generator=torch.Generator().manual_seed(7)) < 0.7
        mask = keep.float() / 0.7
    a0 = act_ref(x0, *c0) if c0 else x0.double()
    parts = [a0]
    if C1:
        parts.append(act_ref(x1, *c1) if c1 else x1.double())
    y_ref = F.conv2d(torch.cat(parts, 1), w.double(), b.double(), padding=1)
    mean_ref = y_ref.mean(dim=(2, 3))
    rstd_ref = 1.0 / torch.sqrt(y_ref.var(dim=(2, 3), unbiased=False) + 1e-5)
    wk, _ = ua.ops.pack_conv3x3_weights(w.to(DEV), want_wd=False)
    uf, _ = ua.ops.pack_wino_weights(w.to(DEV), want_d=False)
    s0 = make_src(ua, x0, c0)
    s1 = make_src(ua, x1, c1) if C1 else None
    args = (s0, s1, SLOPE, wk, b.to(DEV), 3, 1, gamma.to(DEV), beta.to(DEV), 1e-5,
            None if mask is None else mask.to(DEV))
    y, st = ua.ops.conv_in_fwd(*args, wu=uf)
    check(from_nhwc(y), y_ref, 2e-5, "y (Winograd)")
    assert (st[0].cpu().double() - mean_ref).abs().max() <= 2e-5 * (y_ref.abs().max() + 1)
    check(st[1].cpu(), rstd_ref, 5e-5, "rstd (Winograd)")
    y32, st32 = ua.ops.conv_in_fwd(*args)
    check(y, y32, 2e-5, "Winograd vs direct")
    check(st[2], st32[2], 5e-5, "alpha: Winograd vs direct")


@pytest.mark.parametrize("case", [(4, 128, 128, 64, 64, 0), (4, 64, 64, 256, 256, 0),
                                  (2, 128, 128, 128, 128, 0), (4, 64, 64, 256, 256, 512)])
def test_data_gradient_winograd(ua, case):
    """Winograd data gradient vs the direct kernel, with the BSTATS epilogue (reductions of the
    next InstanceNorm backward) and with a column slice of a wider weight (the skip half of a
    decoder stage's first convolution: ci_offset > 0, no reductions)."""
    N, H, W, Cout, Ccols, ci_off = case
    cin_total = ci_off + Ccols
    dy = to_nhwc(rnd(N, Cout, H, W, seed=1))
    w = rnd(Cout, cin_total, 3, 3, seed=2, scale=0.1)
    _, wd = ua.ops.pack_conv3x3_weights(w.to(DEV))
    _, ud = ua.ops.pack_wino_weights(w.to(DEV), want_f=False)
    assert ua.ops.conv_wino_supported(N, H, W, Cout, 0, Ccols)
    ref = ua.ops.conv3x3_bwd_data(dy, wd, ci_off, Ccols, H, W, 1)
    if ci_off:
        g = ua.ops.conv3x3_bwd_data(dy, wd, ci_off, Ccols, H, W, 1, ud=ud)
        check(g, ref, 2e-5, "Winograd data gradient (slice) vs direct")
        return
    y, st, gamma, beta, mask = _next_norm(ua, N, Ccols, H, W, 10)
    nn = ua.ops.NextNorm(y, st, gamma, beta, mask, SLOPE)
    g = ua.ops.conv3x3_bwd_data(dy, wd, 0, Ccols, H, W, 1, nxt=nn, ud=ud)
    check(g, ref, 2e-5, "Winograd data gradient vs direct")
    assert nn.tiles == H * W // 256
    _in_bwd_both_ways(ua, g, nn, y, st, gamma, beta, mask)


C32_CASES = [(2, 64, 64, True), (1, 8, 32, True), (3, 40, 96, False), (2, 128, 160, True),
             (2, 256, 256, True)]


@pytest.mark.parametrize("case", C32_CASES)
def test_c32_winograd_forward(ua, case):
    """The 32 -> 32 channel layers' own Winograd form (csrc/conv_c32.hip: U built in the kernel
    from the packed weights, one K chunk, phase per half tile) against the fp64 convolution of
    the activated source and against the direct kernel behind the same entry point - shapes with
    one tile, odd tile counts (a partial persistent walk) and H != W."""
    N, H, W, act = case
    C = 32
    x0 = rnd(N, C, H, W, seed=1)
    c0 = coeffs(N, C, 10) if act else None
    w = rnd(C, C, 3, 3, seed=3, scale=(2.0 / (9 * C)) ** 0.5)
    b = rnd(C, seed=4, scale=0.3)
    gamma = rnd(C, seed=5) * 0.2 + 1.0
    beta = rnd(C, seed=6) * 0.2
    keep = torch.rand(N, C, generator=torch.Generator().manual_seed(7)) < 0.7
    mask = keep.float() / 0.7
    a0 = act_ref(x0, *c0) if c0 else x0.double()
    y_ref = F.conv2d(a0, w.double(), b.double(), padding=1)
    mean_ref = y_ref.mean(dim=(2, 3))
    rstd_ref = 1.0 / torch.sqrt(y_ref.var(dim=(2, 3), unbiased=False) + 1e-5)
    wk, _ = ua.ops.pack_conv3x3_weights(w.to(DEV), want_wd=False)
    args = (make_src(ua, x0, c0), None, SLOPE, wk, b.to(DEV), 3, 1, gamma.to(DEV), beta.to(DEV),
            1e-5, mask.to(DEV))
    # (by default only launches of >= 512 tiles take this form: "always" for the small shapes)
    assert ua.ops.set_c32_winograd("always") is True
    try:
        assert ua.ops._c32_winograd(N, H, W, C, C, 1)
        y, st = ua.ops.conv_in_fwd(*args)
        ua.ops.set_c32_winograd(False)
        assert not ua.ops._c32_winograd(N, H, W, C, C, 1)
        y32, st32 = ua.ops.conv_in_fwd(*args)
    finally:
        ua.ops.set_c32_winograd(True)
    assert ua.ops._c32_winograd(N, H, W, C, C, 1) == (N * (H // 8) * (W // 32) >= 512)
    assert ua.ops._c32_winograd(8, 512, 512, C, C, 1) and not ua.ops._c32_winograd(8, 512, 512, C, 64, 1)
    check(from_nhwc(y), y_ref, 2e-5, "y (Winograd, 32 channels)")
    assert (st[0].cpu().double() - mean_ref).abs().max() <= 2e-5 * (y_ref.abs().max() + 1)
    check(st[1].cpu(), rstd_ref, 5e-5, "rstd (Winograd, 32 channels)")
    check(y, y32, 2e-5, "Winograd vs direct")
    check(st[2], st32[2], 5e-5, "alpha: Winograd vs direct")


@pytest.mark.parametrize("case", [(2, 64, 64, 0), (1, 8, 32, 0), (3, 40, 96, 0), (2, 64, 96, 64)])
def test_c32_winograd_data_gradient(ua, case):
    """Its data-gradient side: with the BSTATS epilogue (reductions of the next InstanceNorm
    backward on the same y), accumulating into an existing gradient, and as the skip half of the
    last decoder stage's first convolution (32 columns at offset 64 of a 96-channel weight)."""
    N, H, W, ci_off = case
    C = 32
    cin_total = ci_off + C
    dy = to_nhwc(rnd(N, C, H, W, seed=1))
    w = rnd(C, cin_total, 3, 3, seed=2, scale=0.1)
    _, wd = ua.ops.pack_conv3x3_weights(w.to(DEV))
    y, st, gamma, beta, mask = _next_norm(ua, N, C, H, W, 10)
    base = to_nhwc(rnd(N, C, H, W, seed=5))

    def run():
        out = {}
        out["plain"] = ua.ops.conv3x3_bwd_data(dy, wd, ci_off, C, H, W, 1)
        out["acc"] = ua.ops.conv3x3_bwd_data(dy, wd, ci_off, C, H, W, 1, out=base.clone(),
                                             accumulate=True)
        if not ci_off:
            nn = ua.ops.NextNorm(y, st, gamma, beta, mask, SLOPE)
            out["bs"] = ua.ops.conv3x3_bwd_data(dy, wd, 0, C, H, W, 1, nxt=nn)
            out["nn"] = nn
        return out

    ua.ops.set_c32_winograd("always")
    try:
        assert ua.ops._c32_winograd(N, H, W, C, C, 1)
        wino = run()
        ua.ops.set_c32_winograd(False)
        direct = run()
    finally:
        ua.ops.set_c32_winograd(True)
    check(wino["plain"], direct["plain"], 2e-5, "Winograd data gradient vs direct")
    check(wino["acc"], direct["acc"], 2e-5, "accumulating Winograd data gradient vs direct")
    if not ci_off:
        check(wino["bs"], direct["bs"], 2e-5, "Winograd data gradient (+BSTATS) vs direct")
        assert wino["nn"].tiles == direct["nn"].tiles == H * W // 256
        _in_bwd_both_ways(ua, wino["bs"], wino["nn"], y, st, gamma, beta, mask)


@pytest.mark.parametrize("case", [(2, 64, 64, True), (1, 8, 32, True), (3, 40, 96, False),
                                  (2, 256, 256, True)])
def test_c32_winograd_weight_gradient(ua, case):
    """The 32 -> 32 channel layers' Winograd F(3x3,2x2) weight gradient (csrc/conv_wgrad.hip:
    conv_wgrad_wino32_kernel - both operands transformed on chip, all 16 xi accumulated over a
    persistent walk, A^T M A once per workgroup) against the fp64 gradient of the activated
    operand and against the direct kernel behind the same entry point; written into a column
    slice of a wider gradient (the skip half of dec4.0: 32 of 96 input channels)."""
    N, H, W, act = case
    C = 32
    x = rnd(N, C, H, W, seed=1)
    c0 = coeffs(N, C, 10) if act else None
    dy = rnd(N, C, H, W, seed=2)
    a = act_ref(x, *c0) if c0 else x.double()
    ref = torch.nn.grad.conv2d_weight(a, (C, C, 3, 3), dy.double(), padding=1)
    src = make_src(ua, x, c0)
    dyd = to_nhwc(dy)

    def run(form):
        ua.ops.set_c32_winograd(form)
        try:
            assert bool(ua._lib.lib().unet_conv3x3_bwd_weight_is_winograd(N, H, W, C, C, 1)) == bool(form)
            dw = torch.zeros(C, C, 3, 3, device=DEV)
            ua.ops.conv_in_bwd_weight(src, SLOPE, dyd, dw, 0, 3, 1)
            wide = torch.full((C, 96, 3, 3), 7.0, device=DEV)
            ua.ops.conv_in_bwd_weight(src, SLOPE, dyd, wide, 64, 3, 1)
        finally:
            ua.ops.set_c32_winograd(True)
        return dw, wide

    dw, wide = run("always")
    dw_d, wide_d = run(False)
    check(dw.cpu(), ref, 2e-5, "Winograd weight gradient (32 channels) vs fp64")
    check(dw, dw_d, 2e-5, "Winograd vs direct weight gradient")
    assert torch.equal(wide[:, 64:], dw) and bool((wide[:, :64] == 7.0).all())
    assert torch.equal(wide_d[:, 64:], dw_d)


@pytest.mark.parametrize("case", [(2, 64, 64), (1, 8, 32), (3, 40, 96), (2, 256, 256)])
def test_c32_weight_gradient_applies_the_instnorm_backward_on_load(ua, case):
    """unet_conv_in_bwd_weight_dz: the dy side of the 32 -> 32 channel Winograd weight gradient
    forms dz = dL/dy from (g, y) on load and writes it for the data gradient - against the
    elementwise pass (unet_instnorm_lrelu_drop_bwd) followed by the plain weight gradient: dz,
    dgamma, dbeta, dbias and dw."""
    N, H, W = case
    C = 32
    x = rnd(N, C, H, W, seed=1)
    c0 = coeffs(N, C, 10)
    src = make_src(ua, x, c0)
    y, st, gamma, beta, mask = _next_norm(ua, N, C, H, W, 10)
    ua.ops.set_c32_winograd("always")
    try:
        assert ua.ops.conv_in_bwd_weight_dz_supported(N, H, W, C, C)
        # g = dL/da of the layer and its per-tile reductions, as the net gets them: from the
        # data gradient of the layer behind it (BSTATS epilogue)
        w2 = rnd(C, C, 3, 3, seed=7, scale=0.1)
        _, wd2 = ua.ops.pack_conv3x3_weights(w2.to(DEV))
        nn = ua.ops.NextNorm(y, st, gamma, beta, mask, SLOPE)
        g = ua.ops.conv3x3_bwd_data(to_nhwc(rnd(N, C, H, W, seed=2)), wd2, 0, C, H, W, 1, nxt=nn)
        assert nn.tiles == H * W // 256
        # reference: elementwise pass, then the (Winograd) weight gradient of its result
        dgam, dbet, dbia = (torch.zeros(C, device=DEV) for _ in range(3))
        dz_ref = ua.ops.instnorm_lrelu_drop_bwd(g.clone(), y, st[0], st[1], gamma, beta, mask, SLOPE,
                                                dgam, dbet, dbia)
        dw_ref = torch.zeros(C, C, 3, 3, device=DEV)
        ua.ops.conv_in_bwd_weight(src, SLOPE, dz_ref, dw_ref, 0, 3, 1)
        coef5, sums = ua.ops.instnorm_bwd_coefs(y, st[0], st[1], gamma, beta, mask,
                                                (nn.partial, nn.tiles))
        dgam2, dbet2, dbia2 = (torch.zeros(C, device=DEV) for _ in range(3))
        dw = torch.zeros(C, C, 3, 3, device=DEV)
        g2 = g.clone()
        dz = ua.ops.conv_in_bwd_weight_dz(src, SLOPE, g2, y, coef5, sums, gamma, st[1], SLOPE,
                                          dgam2, dbet2, dbia2, dw, 0)
    finally:
        ua.ops.set_c32_winograd(True)
    assert dz.data_ptr() == g2.data_ptr()          # written over g
    check(dz, dz_ref, 2e-5, "dz formed on load vs the elementwise pass")
    check(dw, dw_ref, 5e-5, "weight gradient")
    check(dgam2, dgam, 5e-5, "dgamma")
    check(dbet2, dbet, 5e-5, "dbeta")
    assert (dbia2 - dbia).abs().max() <= 1e-3 * (1 + dbet.abs().max())   # both ~0 (closed form)


X3_FUSED_CASES = [  # shapes the split patch kernel takes in the fused pipeline
    (1, 256, 256, 32, 32, 128, 1, 3, True, True),    # 128 columns, two sources
    (2, 256, 256, 64, 0, 64, 1, 3, True, False),     # 64 columns, 8-row tiles
    (1, 128, 256, 64, 0, 64, 1, 3, True, False),     # 64 columns, 4-row tiles
    (2, 256, 256, 32, 32, 32, 1, 3, False, True),    # 32 columns x 8 rows; plain src0
    (2, 16, 16, 64, 0, 64, 1, 3, True, False),       # too few tiles: fp32 kernel behind the same entry
]


@pytest.mark.parametrize("case", X3_FUSED_CASES)
def test_conv_in_fwd_split_bf16(ua, case):
    """The split-bf16 operand mode behind the fused entry point: fp32-class results (same
    tolerance as the fp32 test) with the activation applied before the operand split."""
    N, H, W, C0, C1, Cout, stride, ks, act0, act1 = case
    x0 = rnd(N, C0, H, W, seed=1)
    x1 = rnd(N, C1, H, W, seed=2) if C1 else None
    c0 = coeffs(N, C0, 10) if act0 else None
    c1 = coeffs(N, C1, 20) if (act1 and C1) else None
    w = rnd(Cout, C0 + C1, ks, ks, seed=3, scale=(2.0 / (ks * ks * (C0 + C1))) ** 0.5)
    b = rnd(Cout, seed=4, scale=0.3)
    gamma = rnd(Cout, seed=5) * 0.2 + 1.0
    beta = rnd(Cout, seed=6) * 0.2
    keep = torch.rand(N, Cout, generator=torch.Generator().manual_seed(7)) < 0.7
    mask = keep.float() / 0.7
    a0 = act_ref(x0, *c0) if c0 else x0.double()
    parts = [a0]
    if C1:
        parts.append(act_ref(x1, *c1) if c1 else x1.double())
    y_ref = F.conv2d(torch.cat(parts, 1), w.double(), b.double(), stride=stride, padding=1)
    mean_ref = y_ref.mean(dim=(2, 3))
    rstd_ref = 1.0 / torch.sqrt(y_ref.var(dim=(2, 3), unbiased=False) + 1e-5)
    wk, _ = ua.ops.pack_conv3x3_weights(w.to(DEV), want_wd=False)
    wf3, _ = ua.ops.pack_conv3x3_weights_bf16x3(w.to(DEV))
    s0 = make_src(ua, x0, c0)
    s1 = make_src(ua, x1, c1) if C1 else None
    y, st = ua.ops.conv_in_fwd(s0, s1, SLOPE, wk, b.to(DEV), ks, stride, gamma.to(DEV),
                               beta.to(DEV), 1e-5, mask.to(DEV), w3=wf3)
    check(from_nhwc(y), y_ref, 2e-5, "y (bf16x3)")
    assert (st[0].cpu().double() - mean_ref).abs().max() <= 2e-5 * (y_ref.abs().max() + 1)
    check(st[1].cpu(), rstd_ref, 5e-5, "rstd (bf16x3)")
    # against the fp32 kernel of the same entry point: both within fp32 rounding of fp64
    y32, _ = ua.ops.conv_in_fwd(s0, s1, SLOPE, wk, b.to(DEV), ks, stride, gamma.to(DEV),
                                beta.to(DEV), 1e-5, mask.to(DEV))
    check(y, y32, 2e-5, "bf16x3 vs fp32 MFMA")


@pytest.mark.parametrize("case", [(1, 256, 256, 32, 128, 1, False), (2, 256, 256, 64, 64, 1, True),
                                  (2, 256, 256, 32, 32, 1, False)])
def test_data_gradient_split_bf16_emits_reductions(ua, case):
    N, H, W, Cout, Ccols, stride, acc = case
    dy = to_nhwc(rnd(N, Cout, H, W, seed=1))
    w = rnd(Cout, Ccols, 3, 3, seed=2, scale=0.1)
    _, wd = ua.ops.pack_conv3x3_weights(w.to(DEV))
    _, wd3 = ua.ops.pack_conv3x3_weights_bf16x3(w.to(DEV))
    y, st, gamma, beta, mask = _next_norm(ua, N, Ccols, H, W, 10)
    base = to_nhwc(rnd(N, Ccols, H, W, seed=3)) if acc else None
    ref = ua.ops.conv3x3_bwd_data(dy, wd, 0, Ccols, H, W, stride,
                                  out=base.clone() if acc else None, accumulate=acc)
    nn = ua.ops.NextNorm(y, st, gamma, beta, mask, SLOPE)
    g = ua.ops.conv3x3_bwd_data(dy, wd, 0, Ccols, H, W, stride, out=base.clone() if acc else None,
                                accumulate=acc, nxt=nn, bf16="bf16x3", wd3=wd3)
    check(g, ref, 2e-5, "bf16x3 data gradient vs fp32 MFMA")
    assert nn.tiles > 0
    _in_bwd_both_ways(ua, g, nn, y, st, gamma, beta, mask)


WGRAD_CASES = [  # (N, H, W, Cx, Cout, stride, ksize, act)
    (2, 12, 64, 32, 32, 1, 3, True),     # 32x32 tile
    (1, 16, 32, 64, 64, 1, 3, True),     # 64x64 tile
    (1, 16, 16, 32, 64, 1, 3, True),     # 32x64 tile
    (2, 16, 32, 64, 64, 2, 3, True),     # stride 2
    (2, 32, 64, 32, 32, 2, 3, True),
    (3, 4, 4, 64, 128, 1, 3, True),      # several images per block
    (2, 2, 2, 512, 512, 1, 1, True),     # 1x1 (centre tap)
    (1, 16, 16, 64, 64, 1, 3, False),    # plain operand through the same entry point
    (2, 8, 128, 3, 32, 1, 3, False),     # RGB stem
    (2, 32, 64, 64, 64, 1, 3, True),     # Winograd F(3x3,2x2) form: 64 chunks of 8 tiles
    (1, 64, 32, 128, 64, 1, 3, True),    # Winograd, two ci tiles, image borders on every chunk
    (3, 16, 48, 64, 128, 1, 3, False),   # Winograd, plain operand, 3 chunks per tile row
]


@pytest.mark.parametrize("x3", [False, True], ids=["fp32", "bf16x3"])
@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_in_bwd_weight(ua, case, x3):
    N, H, W, Cx, Cout, stride, ks, act = case
    x = rnd(N, Cx, H, W, seed=1)
    coef = coeffs(N, Cx, 30) if act else None
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = rnd(N, Cout, Ho, Wo, seed=2)
    a = (act_ref(x, *coef) if coef else x.double()).requires_grad_(False)
    wz = torch.zeros(Cout, Cx, ks, ks, dtype=torch.double, requires_grad=True)
    F.conv2d(a, wz, None, stride=stride, padding=ks // 2).backward(dy.double())
    # embed in a wider gradient tensor at a channel offset, like the second source of a concat
    off, total = 32, Cx + 64
    if Cx == 3:
        off, total = 0, 3
    dw = torch.full((Cout, total, ks, ks), 7.0, device=DEV)
    ua.ops.conv_in_bwd_weight(make_src(ua, x, coef), SLOPE, to_nhwc(dy), dw, off, ks, stride, x3=x3)
    check(dw[:, off:off + Cx].cpu(), wz.grad, 3e-5, "dw")
    if total > Cx:
        assert torch.all(dw[:, :off] == 7.0) and torch.all(dw[:, off + Cx:] == 7.0)


@pytest.mark.parametrize("shape", [(2, 8, 12, 64), (1, 1, 1, 32), (2, 2, 2, 512), (1, 16, 16, 32)])
def test_upsample2x_in_fwd(ua, shape):
    N, h, w, C = shape
    x = rnd(N, C, h, w, seed=1)
    coef = coeffs(N, C, 40)
    ref = F.interpolate(act_ref(x, *coef), scale_factor=2, mode="bilinear", align_corners=False)
    up = ua.ops.upsample2x_in_fwd(make_src(ua, x, coef), SLOPE)
    check(from_nhwc(up), ref, 1e-6, "upsample(act(x))")
    # plain operand: the stand-alone kernel's result
    up2 = ua.ops.upsample2x_in_fwd(ua.ops.Act(to_nhwc(x)), SLOPE)
    assert torch.equal(up2, ua.ops.upsample2x_fwd(to_nhwc(x)))


@pytest.mark.parametrize("shape", [(2, 12, 20), (1, 64, 64), (3, 17, 9)])
def test_head1x1_in(ua, shape):
    N, H, W = shape
    x = rnd(N, 32, H, W, seed=1)
    coef = coeffs(N, 32, 50)
    w = rnd(3, 32, seed=2, scale=0.2)
    b = rnd(3, seed=3, scale=0.1)
    a = act_ref(x, *coef).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    logits_ref = F.conv2d(a, wr[:, :, None, None], br)
    dl = rnd(N, 3, H, W, seed=4)
    logits_ref.backward(dl.double())
    src = make_src(ua, x, coef)
    logits = ua.ops.head1x1_in_fwd(src, SLOPE, w.to(DEV), b.to(DEV))
    check(logits.cpu(), logits_ref.detach(), 2e-5, "logits")
    dw, db = torch.empty(3, 32, device=DEV), torch.empty(3, device=DEV)
    da = ua.ops.head1x1_in_bwd(src, SLOPE, dl.to(DEV), w.to(DEV), dw, db)
    check(from_nhwc(da), a.grad, 2e-5, "da")
    check(dw.cpu(), wr.grad, 3e-5, "dw")
    check(db.cpu(), br.grad, 3e-5, "db")


# --------------------------------------------------------------------------- composed blocks (golden)
def _run_block_fused(ua, s0, s1, p, idx, stride, masks):
    """The two conv units of a reference ConvBlock state_dict `p` on the fused pipeline."""
    recs = []
    for k, (ci, ni) in enumerate(idx):
        w = p[f"block.{ci}.weight"].to(DEV)
        wf, wd = ua.ops.pack_conv3x3_weights(w)
        gm, bt = p[f"block.{ni}.weight"].to(DEV), p[f"block.{ni}.bias"].to(DEV)
        m = masks[k].to(DEV) if masks is not None else None
        y, st = ua.ops.conv_in_fwd(s0, s1, SLOPE, wf, p[f"block.{ci}.bias"].to(DEV), 3,
                                   stride if k == 0 else 1, gm, bt, 1e-5, m)
        recs.append(dict(x0=s0, x1=s1, y=y, st=st, m=m, wd=wd, gm=gm, bt=bt, ci=ci, ni=ni,
                         stride=stride if k == 0 else 1, w=w))
        s0, s1 = ua.ops.Act(y, st[2], st[3]), None
    return recs, s0


def _block_backward_fused(ua, recs, g):
    grads = {}
    dx1 = None
    for r in reversed(recs):
        C = r["y"].shape[3]
        dg, dbt, dbias = (torch.empty(C, device=DEV) for _ in range(3))
        dy = ua.ops.instnorm_lrelu_drop_bwd(g, r["y"], r["st"][0], r["st"][1], r["gm"], r["bt"],
                                            r["m"], SLOPE, dg, dbt, dbias)
        dw = torch.empty_like(r["w"])
        ua.ops.conv_in_bwd_weight(r["x0"], SLOPE, dy, dw, 0, 3, r["stride"])
        N, H, W, C0 = r["x0"].shape
        if r["x1"] is not None:
            ua.ops.conv_in_bwd_weight(r["x1"], SLOPE, dy, dw, C0, 3, r["stride"])
            dx1 = ua.ops.conv3x3_bwd_data(dy, r["wd"], C0, r["x1"].shape[3], H, W, r["stride"])
        g = ua.ops.conv3x3_bwd_data(dy, r["wd"], 0, C0, H, W, r["stride"])
        grads[f"block.{r['ci']}.weight"] = dw
        grads[f"block.{r['ci']}.bias"] = dbias
        grads[f"block.{r['ni']}.weight"] = dg
        grads[f"block.{r['ni']}.bias"] = dbt
    return g, dx1, grads


def _materialise(ua, act):
    """a = lrelu(y * alpha + beta) through the stand-alone apply kernel (test helper)."""
    return ua.ops.instnorm_lrelu_drop_fwd(act.x, act.alpha, act.beta, None, SLOPE)


def _check_grads(grads, g, prefix, what):
    for k, v in grads.items():
        ref = torch.from_numpy(g[prefix + k])
        if k.endswith("bias") and ref.abs().max() < 1e-4:   # conv bias under IN: ~0 +- rounding
            assert (v.cpu() - ref).abs().max() < 1e-4
        else:
            check(v.cpu(), ref, 5e-5, f"{what} grad {k}")


def test_convblock_golden_fused(ua, golden):
    """Reference ConvBlock(32->64, stride 2, dropout 0.2) in train mode, forward + backward."""
    g = golden("ops_small")
    p = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("cb_p_")}
    masks = [torch.from_numpy(g["cb_mask0"]), torch.from_numpy(g["cb_mask1"])]
    recs, out = _run_block_fused(ua, ua.ops.Act(to_nhwc(torch.from_numpy(g["cb_x"]))), None, p,
                                 [(0, 1), (4, 5)], 2, masks)
    check(from_nhwc(_materialise(ua, out)), torch.from_numpy(g["cb_y"]), 2e-5, "ConvBlock fwd")
    gx, _, grads = _block_backward_fused(ua, recs, to_nhwc(torch.from_numpy(g["cb_gy"])))
    check(from_nhwc(gx), torch.from_numpy(g["cb_gx"]), 5e-5, "ConvBlock gx")
    _check_grads(grads, g, "cb_g_", "ConvBlock")


def test_upblock_golden_fused(ua, golden):
    """Reference UpBlock(64 up + 32 skip -> 32), eval mode, forward + backward."""
    g = golden("ops_small")
    p = {k[16:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ub_p_conv_block.")}
    up = ua.ops.Act(ua.ops.upsample2x_in_fwd(ua.ops.Act(to_nhwc(torch.from_numpy(g["ub_x"]))), SLOPE))
    recs, out = _run_block_fused(ua, up, ua.ops.Act(to_nhwc(torch.from_numpy(g["ub_skip"]))), p,
                                 [(0, 1), (3, 4)], 1, None)
    check(from_nhwc(_materialise(ua, out)), torch.from_numpy(g["ub_y"]), 2e-5, "UpBlock fwd")
    g_up, g_skip, grads = _block_backward_fused(ua, recs, to_nhwc(torch.from_numpy(g["ub_gy"])))
    gx = ua.ops.upsample2x_bwd(g_up)
    check(from_nhwc(gx), torch.from_numpy(g["ub_gx"]), 5e-5, "UpBlock gx")
    check(from_nhwc(g_skip), torch.from_numpy(g["ub_gskip"]), 5e-5, "UpBlock gskip")
    _check_grads(grads, g, "ub_g_conv_block.", "UpBlock")


# --------------------------------------------------------------------------- low-resolution backward
UP_CASES = [  # (N, h, w, Cx, Cout, act)
    (2, 8, 16, 64, 64, True),      # 64x64 channel tiles
    (1, 16, 16, 32, 64, True),     # 32x64
    (2, 8, 8, 64, 32, True),       # 32x32
    (3, 2, 2, 64, 64, True),       # 2x2 maps: a segment spans images, every border case
    (2, 1, 3, 32, 32, False),      # single row, plain operand
    (1, 16, 32, 128, 64, True),
    (2, 8, 8, 128, 128, True),     # deep layer: gather-GEMM with four K groups
]


@pytest.mark.parametrize("case", UP_CASES)
def test_conv3x3_up_backward_at_low_resolution(ua, case):
    """dW and dL/da of conv3x3(upsample2x(act(x))) through D = upsample2x_bwd_taps(dy) against
    torch autograd of the composed ops (Our_UNet/models/unet.py:219-231)."""
    N, h, w, Cx, Cout, act = case
    x = rnd(N, Cx, h, w, seed=1)
    coef = coeffs(N, Cx, 60) if act else None
    dy = rnd(N, Cout, 2 * h, 2 * w, seed=2)
    wt = rnd(Cout, Cx + 32, 3, 3, seed=3, scale=0.1)     # the up operand is the first Cx channels
    a = (act_ref(x, *coef) if coef else x.double()).requires_grad_(True)
    wr = wt[:, :Cx].double().requires_grad_(True)
    up = F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=False)
    F.conv2d(up, wr, None, padding=1).backward(dy.double())
    D = ua.ops.upsample2x_bwd_taps(to_nhwc(dy))
    assert D.shape == (N, h, w, 9 * Cout)
    dw = torch.full((Cout, Cx + 32, 3, 3), 7.0, device=DEV)
    ua.ops.conv3x3_up_bwd_weight(make_src(ua, x, coef), SLOPE, D, dw, 0)
    check(dw[:, :Cx].cpu(), wr.grad, 3e-5, "dw (up operand)")
    assert torch.all(dw[:, Cx:] == 7.0)
    _, wd = ua.ops.pack_conv3x3_weights(wt.to(DEV))
    g = ua.ops.conv3x3_up_bwd_data(D, wd, 0, Cx)
    check(from_nhwc(g), a.grad, 3e-5, "dL/da (low resolution)")
    g2 = ua.ops.conv3x3_up_bwd_data(D, wd, 0, Cx, out=g.clone(), accumulate=True)
    check(from_nhwc(g2), 2 * a.grad, 3e-5, "accumulate")


# --------------------------------------------------------------------------- 2 GiB batch chunking
@pytest.fixture
def small_chunks(ua):
    """Lower the buffer-range threshold so a 4-image batch is processed in three chunks."""
    def setter(per_image_bytes):
        ua.lib().unet_debug_set_chunk_limit(int(1.5 * per_image_bytes))
    yield setter
    ua.lib().unet_debug_set_chunk_limit(0)


def test_batch_chunking_matches_one_pass(ua, small_chunks):
    """The reference trains at bs 32 (Our_UNet/src/train.py:748), where the 64-channel 512x512
    tensors reach 2 GiB - past the buffer-descriptor range of the conv kernels.  The entry
    points split such batches over N.  Here the threshold is lowered instead of the tensors
    grown: every chunked entry point must reproduce its one-pass result exactly (forward,
    statistics, data gradient) or to summation order (weight gradients: more slabs)."""
    N, H, W, C0, C1, Cout = 4, 16, 32, 64, 32, 64
    x0, x1 = rnd(N, C0, H, W, seed=1), rnd(N, C1, H, W, seed=2)
    c0, c1 = coeffs(N, C0, 10), coeffs(N, C1, 20)
    w = rnd(Cout, C0 + C1, 3, 3, seed=3, scale=0.05)
    b, gamma, beta = rnd(Cout, seed=4), rnd(Cout, seed=5) * 0.2 + 1, rnd(Cout, seed=6) * 0.2
    dy = to_nhwc(rnd(N, Cout, H, W, seed=7))
    wf, wd = ua.ops.pack_conv3x3_weights(w.to(DEV))
    s0, s1 = make_src(ua, x0, c0), make_src(ua, x1, c1)
    xl = rnd(N, C0, H // 2, W // 2, seed=8)
    sl = make_src(ua, xl, coeffs(N, C0, 30))

    def run():
        y, st = ua.ops.conv_in_fwd(s0, s1, SLOPE, wf, b.to(DEV), 3, 1, gamma.to(DEV), beta.to(DEV),
                                   1e-5, None)
        y2 = ua.ops.conv3x3_fwd(s0.x, s1.x, wf, b.to(DEV), 2)
        dx = ua.ops.conv3x3_bwd_data(dy, wd, 0, C0, H, W, 1)
        dw = torch.zeros(Cout, C0 + C1, 3, 3, device=DEV)
        ua.ops.conv_in_bwd_weight(s0, SLOPE, dy, dw, 0, 3, 1)
        ua.ops.conv_in_bwd_weight(s1, SLOPE, dy, dw, C0, 3, 1)
        D = ua.ops.upsample2x_bwd_taps(dy)
        dwu = torch.zeros(Cout, C0, 3, 3, device=DEV)
        ua.ops.conv3x3_up_bwd_weight(sl, SLOPE, D, dwu, 0)
        gl = ua.ops.conv3x3_up_bwd_data(D, wd, 0, C0)
        return [t.clone() for t in (y, st, y2, dx, dw, dwu, gl)]

    ref = run()
    # threshold = 1.5 x the largest per-image operand (D: 9*Cout channels at half resolution):
    # D goes one image at a time, the 64-channel sources three + one
    small_chunks((H // 2) * (W // 2) * 9 * Cout * 4)
    got = run()
    for name, a, r in zip(("y", "stats", "y stride 2", "dx", "dw", "dw up", "g low"), got, ref):
        if name in ("dw", "dw up", "stats"):
            check(a, r, 2e-6, name)
        else:
            assert torch.equal(a, r), name


def test_reference_batch_size_32_layer_runs(ua):
    """dec4.0 of the bs-32 run the reference does: 64 up-sampled + 32 skip channels at 512x512.
    Source 0 is exactly 2^31 bytes.  The call used to return UNET_E_INVALID; now it must run and
    agree with the same layer evaluated per 8-image quarter."""
    N, H, W, C0, C1, Cout = 32, 512, 512, 64, 32, 32
    g = torch.Generator(device=DEV).manual_seed(3)
    x0 = torch.randn(N, H, W, C0, device=DEV, generator=g)
    x1 = torch.randn(N, H, W, C1, device=DEV, generator=g)
    assert x0.numel() * 4 == 1 << 31
    w = rnd(Cout, C0 + C1, 3, 3, seed=3, scale=0.05).to(DEV)
    wf, wd = ua.ops.pack_conv3x3_weights(w)
    b = rnd(Cout, seed=4).to(DEV)
    y = ua.ops.conv3x3_fwd(x0, x1, wf, b, 1)
    for q in range(4):
        sl = slice(8 * q, 8 * q + 8)
        yq = ua.ops.conv3x3_fwd(x0[sl].contiguous(), x1[sl].contiguous(), wf, b, 1)
        assert torch.equal(y[sl], yq)
    dw = torch.zeros(Cout, C0 + C1, 3, 3, device=DEV)
    ua.ops.conv_in_bwd_weight(ua.ops.Act(x0), SLOPE, y, dw, 0, 3, 1)
    dwq = torch.zeros_like(dw)
    acc = torch.zeros_like(dw)
    for q in range(4):
        sl = slice(8 * q, 8 * q + 8)
        ua.ops.conv_in_bwd_weight(ua.ops.Act(x0[sl].contiguous()), SLOPE, y[sl].contiguous(), dwq,
                                  0, 3, 1)
        acc += dwq
    check(dw[:, :C0], acc[:, :C0], 1e-5, "dw over the 2 GiB operand")


# --------------------------------------------------------------------------- up-sampling in the loader
UPFWD_CASES = [  # (N, H, W, C0, C1, Cout, act0, act1): H, W = output (skip) size
    (1, 256, 256, 64, 32, 128, True, True),      # 128-column tile
    (1, 256, 256, 32, 32, 64, True, True),       # 64-column tile
    (2, 256, 256, 64, 32, 32, True, False),      # 32 columns x 8 rows; plain skip
    (8, 32, 32, 64, 32, 512, True, True),        # small maps: every tile touches an image border
    (1, 256, 256, 32, 0, 64, False, True),       # no skip source, plain low-resolution operand
]


@pytest.mark.parametrize("case", UPFWD_CASES)
def test_conv_up_in_fwd(ua, case):
    """conv3x3(cat(upsample2x(act(low)), act(skip))) with the bilinear gather in the patch loader
    against F.interpolate + torch.cat + F.conv2d (Our_UNet/models/unet.py:215-231), statistics
    included."""
    N, H, W, C0, C1, Cout, act0, act1 = case
    low = rnd(N, C0, H // 2, W // 2, seed=1)
    skip = rnd(N, C1, H, W, seed=2) if C1 else None
    c0 = coeffs(N, C0, 10) if act0 else None
    c1 = coeffs(N, C1, 20) if (act1 and C1) else None
    w = rnd(Cout, C0 + C1, 3, 3, seed=3, scale=(2.0 / (9 * (C0 + C1))) ** 0.5)
    b = rnd(Cout, seed=4, scale=0.3)
    gamma, beta = rnd(Cout, seed=5) * 0.2 + 1.0, rnd(Cout, seed=6) * 0.2
    a0 = act_ref(low, *c0) if c0 else low.double()
    parts = [F.interpolate(a0, scale_factor=2, mode="bilinear", align_corners=False)]
    if C1:
        parts.append(act_ref(skip, *c1) if c1 else skip.double())
    y_ref = F.conv2d(torch.cat(parts, 1), w.double(), b.double(), padding=1)
    wf, _ = ua.ops.pack_conv3x3_weights(w.to(DEV), want_wd=False)
    s_low = make_src(ua, low, c0)
    if not C1:
        pytest.skip("single-source form is not exposed")
    s_skip = make_src(ua, skip, c1)
    assert ua.ops.conv_up_in_fwd_supported(s_low, s_skip, Cout)
    y, st = ua.ops.conv_up_in_fwd(s_low, s_skip, SLOPE, wf, b.to(DEV), gamma.to(DEV), beta.to(DEV),
                                  1e-5, None)
    check(from_nhwc(y), y_ref, 2e-5, "y")
    assert (st[0].cpu().double() - y_ref.mean(dim=(2, 3))).abs().max() <= 2e-5 * (y_ref.abs().max() + 1)
    check(st[1].cpu(), 1.0 / torch.sqrt(y_ref.var(dim=(2, 3), unbiased=False) + 1e-5), 5e-5, "rstd")
    # and it is the same as the two-kernel path (materialised up-sampling)
    up = ua.ops.Act(ua.ops.upsample2x_in_fwd(s_low, SLOPE))
    y2, _ = ua.ops.conv_in_fwd(up, s_skip, SLOPE, wf, b.to(DEV), 3, 1, gamma.to(DEV), beta.to(DEV),
                               1e-5, None)
    check(y, y2, 2e-6, "fused vs materialised up-sampling")


@pytest.mark.parametrize("case", [(2, 128, 128, 128, 64, 128), (8, 32, 32, 64, 32, 512),
                                  (1, 256, 256, 64, 32, 64), (4, 64, 64, 512, 256, 256)])
def test_conv_up_in_fwd_winograd(ua, case):
    """The decoder stage's first convolution on the Winograd kernel (bilinear gather of the
    low-resolution source inside its loader) against F.interpolate + torch.cat + F.conv2d and
    against the direct up-sampling-loader kernel; image borders on every side."""
    N, H, W, C0, C1, Cout = case
    assert ua.ops.conv_up_wino_supported(N, H, W, C0, C1, Cout)
    low = rnd(N, C0, H // 2, W // 2, seed=1)
    skip = rnd(N, C1, H, W, seed=2)
    c0, c1 = coeffs(N, C0, 10), coeffs(N, C1, 20)
    w = rnd(Cout, C0 + C1, 3, 3, seed=3, scale=(2.0 / (9 * (C0 + C1))) ** 0.5)
    b = rnd(Cout, seed=4, scale=0.3)
    gamma, beta = rnd(Cout, seed=5) * 0.2 + 1.0, rnd(Cout, seed=6) * 0.2
    parts = [F.interpolate(act_ref(low, *c0), scale_factor=2, mode="bilinear", align_corners=False),
             act_ref(skip, *c1)]
    y_ref = F.conv2d(torch.cat(parts, 1), w.double(), b.double(), padding=1)
    wf, _ = ua.ops.pack_conv3x3_weights(w.to(DEV), want_wd=False)
    uf, _ = ua.ops.pack_wino_weights(w.to(DEV), want_d=False)
    s_low, s_skip = make_src(ua, low, c0), make_src(ua, skip, c1)
    args = (s_low, s_skip, SLOPE, wf, b.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, None)
    y, st = ua.ops.conv_up_in_fwd(*args, wu=uf)
    check(from_nhwc(y), y_ref, 2e-5, "y (Winograd, up-sampling loader)")
    assert (st[0].cpu().double() - y_ref.mean(dim=(2, 3))).abs().max() <= 2e-5 * (y_ref.abs().max() + 1)
    check(st[1].cpu(), 1.0 / torch.sqrt(y_ref.var(dim=(2, 3), unbiased=False) + 1e-5), 5e-5, "rstd")
    y2, st2 = ua.ops.conv_up_in_fwd(*args)
    check(y, y2, 2e-5, "Winograd vs direct up-sampling loader")
    check(st[2], st2[2], 5e-5, "alpha")


@pytest.mark.parametrize("case", [(2, 64, 64), (1, 8, 32), (3, 40, 96), (2, 256, 256)])
def test_conv_up_in_fwd_c32_winograd(ua, case):
    """The last decoder stage's first convolution, (64 up-sampled + 32 skip) -> 32 channels, on
    its own Winograd kernel (csrc/conv_c32.hip: conv_wino_up32_kernel - three register-resident
    chunks, the bilinear up-sampling folded into the input transform of a 3 x 3 low-resolution
    window) against F.interpolate + torch.cat + F.conv2d in fp64 and against the direct
    up-sampling-loader kernel: one-tile images (every border rule at once), odd tile counts."""
    N, H, W = case
    C0, C1, Cout = 64, 32, 32
    low = rnd(N, C0, H // 2, W // 2, seed=1)
    skip = rnd(N, C1, H, W, seed=2)
    c0, c1 = coeffs(N, C0, 10), coeffs(N, C1, 20)
    w = rnd(Cout, C0 + C1, 3, 3, seed=3, scale=(2.0 / (9 * (C0 + C1))) ** 0.5)
    b = rnd(Cout, seed=4, scale=0.3)
    gamma, beta = rnd(Cout, seed=5) * 0.2 + 1.0, rnd(Cout, seed=6) * 0.2
    parts = [F.interpolate(act_ref(low, *c0), scale_factor=2, mode="bilinear", align_corners=False),
             act_ref(skip, *c1)]
    y_ref = F.conv2d(torch.cat(parts, 1), w.double(), b.double(), padding=1)
    wf, _ = ua.ops.pack_conv3x3_weights(w.to(DEV), want_wd=False)
    args = (make_src(ua, low, c0), make_src(ua, skip, c1), SLOPE, wf, b.to(DEV), gamma.to(DEV),
            beta.to(DEV), 1e-5, None)
    lib = ua._lib.lib()
    ua.ops.set_c32_winograd("always")
    try:
        assert lib.unet_conv_up_c32_is_winograd(N, H, W, C0, C1, Cout) == 1
        y, st = ua.ops.conv_up_in_fwd(*args)
    finally:
        ua.ops.set_c32_winograd(True)
    assert lib.unet_conv_up_c32_is_winograd(N, H, W, C0, C1, Cout) == int(N * (H // 8) * (W // 32) >= 256)
    check(from_nhwc(y), y_ref, 2e-5, "y (Winograd, folded up-sampling)")
    assert (st[0].cpu().double() - y_ref.mean(dim=(2, 3))).abs().max() <= 2e-5 * (y_ref.abs().max() + 1)
    check(st[1].cpu(), 1.0 / torch.sqrt(y_ref.var(dim=(2, 3), unbiased=False) + 1e-5), 5e-5, "rstd")
    if ua.ops.conv_up_in_fwd_supported(args[0], args[1], Cout):   # the direct kernel tiles it too
        ua.ops.set_c32_winograd(False)
        try:
            y2, st2 = ua.ops.conv_up_in_fwd(*args)
        finally:
            ua.ops.set_c32_winograd(True)
        check(y, y2, 2e-5, "Winograd vs direct up-sampling loader")
        check(st[2], st2[2], 5e-5, "alpha")


# --------------------------------------------------------------------------- reductions from the producer
def _next_norm(ua, N, C, H, W, seed):
    """A layer l as the producer of dL/da_l sees it: raw output, statistics, affine, mask."""
    y = to_nhwc(rnd(N, C, H, W, seed=seed) * 1.5 + 0.3)
    gamma, beta = (rnd(C, seed=seed + 1) * 0.2 + 1.0).to(DEV), (rnd(C, seed=seed + 2) * 0.2).to(DEV)
    st = ua.ops.instnorm_stats(y, gamma, beta, 1e-5)
    mask = ((torch.rand(N, C, generator=torch.Generator().manual_seed(seed + 3)) < 0.8).float()
            / 0.8).to(DEV)
    return y, st, gamma, beta, mask


def _in_bwd_both_ways(ua, g, nn, y, st, gamma, beta, mask):
    """InstanceNorm backward from the producer's summaries vs the stand-alone reduction pass."""
    C = y.shape[3]
    outs = []
    for partials in ((nn.partial, nn.tiles), None):
        dg, db, dbias = (torch.empty(C, device=DEV) for _ in range(3))
        dy = ua.ops.instnorm_lrelu_drop_bwd(g.clone(), y, st[0], st[1], gamma, beta, mask, SLOPE, dg,
                                            db, dbias, partials=partials)
        outs.append((dy, dg, db))
    for a, b, what in zip(outs[0], outs[1], ("dy", "dgamma", "dbeta")):
        check(a, b, 2e-5, what + " (producer summaries vs reduction pass)")


BS_DGRAD = [  # (N, H, W, Cout, Ccols, stride, accumulate): H, W = size of dx
    (1, 256, 256, 32, 64, 1, False),     # patch-staged 64 columns
    (1, 256, 256, 64, 128, 1, True),     # patch-staged 128 columns, accumulate (skip gradient)
    (1, 128, 128, 32, 32, 1, False),     # 32 -> 32 channel kernel
    (2, 256, 256, 32, 32, 1, True),      # the same, two tiles per workgroup, accumulate
    (1, 12, 128, 32, 32, 1, False),      # row-fused, K = 32
    (2, 16, 16, 64, 64, 1, False),       # gather-GEMM 64x64, two K groups
    (2, 16, 16, 128, 128, 1, True),      # four K groups, accumulate
    (2, 32, 32, 128, 128, 2, False),     # stride-2 per-class launches with 4 / 2 / 2 / 1 K groups
    (2, 256, 256, 32, 64, 2, True),      # stride-2 patch-staged kernel, 64 columns, accumulate
    (2, 512, 512, 64, 32, 2, False),     # stride-2 patch-staged kernel, 32 columns x 8 rows
    (1, 128, 256, 96, 64, 2, False),     # stride-2 patch-staged kernel, three K chunks, H != W
    (1, 64, 64, 32, 64, 2, False),       # stride-2 one-launch gather kernel (too few patch tiles)
    (2, 32, 32, 64, 64, 2, False),       # stride-2 per-class launches
    (3, 4, 4, 32, 32, 1, False),         # tiles span images: no epilogue, tiles == 0
]


@pytest.mark.parametrize("case", BS_DGRAD)
def test_data_gradient_emits_next_norm_reductions(ua, case):
    N, H, W, Cout, Ccols, stride, acc = case
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dy = to_nhwc(rnd(N, Cout, Ho, Wo, seed=1))
    w = rnd(Cout, Ccols, 3, 3, seed=2, scale=0.1)
    _, wd = ua.ops.pack_conv3x3_weights(w.to(DEV))
    y, st, gamma, beta, mask = _next_norm(ua, N, Ccols, H, W, 10)
    base = to_nhwc(rnd(N, Ccols, H, W, seed=3)) if acc else None
    ref = ua.ops.conv3x3_bwd_data(dy, wd, 0, Ccols, H, W, stride,
                                  out=base.clone() if acc else None, accumulate=acc)
    nn = ua.ops.NextNorm(y, st, gamma, beta, mask, SLOPE)
    g = ua.ops.conv3x3_bwd_data(dy, wd, 0, Ccols, H, W, stride, out=base.clone() if acc else None,
                                accumulate=acc, nxt=nn)
    assert torch.equal(g, ref)
    if H * W < 64:
        assert nn.tiles == 0
        return
    assert nn.tiles > 0
    _in_bwd_both_ways(ua, g, nn, y, st, gamma, beta, mask)


def test_low_resolution_gradient_emits_reductions(ua):
    N, h, w, Cx, Cout = 2, 16, 32, 64, 32
    D = ua.ops.upsample2x_bwd_taps(to_nhwc(rnd(N, Cout, 2 * h, 2 * w, seed=1)))
    _, wd = ua.ops.pack_conv3x3_weights(rnd(Cout, Cx, 3, 3, seed=2, scale=0.1).to(DEV))
    y, st, gamma, beta, mask = _next_norm(ua, N, Cx, h, w, 20)
    nn = ua.ops.NextNorm(y, st, gamma, beta, mask, SLOPE)
    g = ua.ops.conv3x3_up_bwd_data(D, wd, 0, Cx, nxt=nn)
    assert torch.equal(g, ua.ops.conv3x3_up_bwd_data(D, wd, 0, Cx)) and nn.tiles > 0
    _in_bwd_both_ways(ua, g, nn, y, st, gamma, beta, mask)


@pytest.mark.parametrize("case", [(8, 512, 512), (2, 256, 128), (1, 64, 64), (3, 8, 8), (1, 8, 12)])
def test_head_backward_emits_next_norm_reductions(ua, case):
    """unet_head1x1_in_bwd_bs (round 4): da is the final gradient of the last decoder layer, so the
    head's backward also sums gz and gz * xhat per workgroup (contiguous tile ranges that divide
    an image) - the stand-alone reduction pass over (da, y) goes.  Same da bits, and the
    summaries drive the InstanceNorm backward like the reduction pass does; shapes that do not
    split evenly report tiles == 0."""
    N, H, W = case
    C, K = 32, 3
    y, st, gamma, beta, mask = _next_norm(ua, N, C, H, W, 30)
    st = st.clone()
    st[2] *= mask          # alpha / beta of the forward carry the dropout factors
    st[3] *= mask
    x = ua.ops.Act(y, st[2].contiguous(), st[3].contiguous())
    dl = rnd(N, K, H, W, seed=5).to(DEV)
    w = (rnd(K, C, seed=6) * 0.2).to(DEV)
    dw0, db0, dw1, db1 = (torch.empty(K, C, device=DEV), torch.empty(K, device=DEV),
                          torch.empty(K, C, device=DEV), torch.empty(K, device=DEV))
    ref = ua.ops.head1x1_in_bwd(x, SLOPE, dl, w, dw0, db0)
    nn = ua.ops.NextNorm(y, st, gamma, beta, mask, SLOPE)
    g = ua.ops.head1x1_in_bwd(x, SLOPE, dl, w, dw1, db1, nxt=nn)
    assert torch.equal(g, ref)
    check(dw1, dw0, 1e-5, "head dw (contiguous tile ranges)")
    check(db1, db0, 1e-5, "head db")
    if (H * W) % 64:       # 96 pixels: no whole number of 64-pixel tiles per image
        assert nn.tiles == 0
        return
    assert nn.tiles > 0
    _in_bwd_both_ways(ua, g, nn, y, st, gamma, beta, mask)


def test_winograd_weight_packing_in_one_launch(ua):
    """unet_pack_wino_weights_batched (PackTable.run: every layer's U = G g G^T in one launch)
    against the per-layer entry point, forward and data-gradient forms, incl. a layer that only
    takes one of the two forms."""
    ws = [rnd(64, 64, 3, 3, seed=1).to(DEV), rnd(128, 192, 3, 3, seed=2).to(DEV),
          rnd(64, 40, 3, 3, seed=3).to(DEV), rnd(32, 32, 3, 3, seed=4).to(DEV)]
    flags = [(True, True), (True, True), (True, False), (False, False)]
    table = ua.ops.PackTable(ws, False, flags)
    table.run()
    for w, (ff, fd), uf, ud in zip(ws, flags, table.uf, table.ud):
        assert (uf is not None) == ff and (ud is not None) == fd
        if not (ff or fd):
            continue
        rf, rd = ua.ops.pack_wino_weights(w, want_f=ff, want_d=fd)
        if ff:
            assert torch.equal(uf, rf.view(-1))
        if fd:
            assert torch.equal(ud, rd.view(-1))


@pytest.mark.parametrize("case", [(4, 128, 128, 128, 128, 0), (2, 256, 256, 64, 64, 0),
                                  (4, 128, 128, 256, 64, 128)])
def test_instnorm_backward_applied_on_load_by_the_winograd_data_gradient(ua, case):
    """unet_instnorm_bwd_coefs + unet_conv3x3_bwd_data_dz_wino (the loader forms dL/dy from
    (g, y) and five coefficient planes, writes it for the weight gradient, emits the layer's
    parameter gradients) against the elementwise pass + the plain Winograd data gradient.
    Third case: a column slice of a wider weight (the skip half of a decoder stage's first
    convolution: ci_offset > 0)."""
    N, H, W, C, Ccols, ci_off = case
    cin_total = ci_off + Ccols
    y, st, gamma, beta, mask = _next_norm(ua, N, C, H, W, 30)
    # g = dL/da of that layer with its reductions, as a producing data gradient leaves them
    dyn = to_nhwc(rnd(N, 64, H, W, seed=1))
    _, wdn = ua.ops.pack_conv3x3_weights(rnd(64, C, 3, 3, seed=2, scale=0.1).to(DEV))
    nn = ua.ops.NextNorm(y, st, gamma, beta, mask, SLOPE)
    g = ua.ops.conv3x3_bwd_data(dyn, wdn, 0, C, H, W, 1, nxt=nn)
    assert nn.tiles > 0
    w = rnd(C, cin_total, 3, 3, seed=3, scale=0.1).to(DEV)
    _, wd = ua.ops.pack_conv3x3_weights(w)
    _, ud = ua.ops.pack_wino_weights(w, want_f=False)
    # the two-pass way
    dg0, db0, dbi0 = (torch.empty(C, device=DEV) for _ in range(3))
    dz0 = ua.ops.instnorm_lrelu_drop_bwd(g.clone(), y, st[0], st[1], gamma, beta, mask, SLOPE, dg0,
                                         db0, dbi0, partials=(nn.partial, nn.tiles))
    dx0 = ua.ops.conv3x3_bwd_data(dz0, wd, ci_off, Ccols, H, W, 1, ud=ud)
    # applied on load
    coef5, sums = ua.ops.instnorm_bwd_coefs(y, st[0], st[1], gamma, beta, mask,
                                            (nn.partial, nn.tiles))
    dg1, db1, dbi1 = (torch.empty(C, device=DEV) for _ in range(3))
    dx1, dz1 = ua.ops.conv3x3_bwd_data_dz(g, y, coef5, sums, gamma, st[1], SLOPE, dg1, db1, dbi1,
                                          ud, cin_total, ci_off, Ccols)
    check(dz1, dz0, 1e-5, "dz written by the data gradient")
    check(dx1, dx0, 2e-5, "dx")
    assert torch.equal(dg1, dg0) and torch.equal(db1, db0)
    assert (dbi1 - dbi0).abs().max() <= 1e-6 * (dz0.abs().max() * H * W)
