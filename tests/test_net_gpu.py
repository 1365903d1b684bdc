"""Whole-network parity (-m gpu): the drop-in UNet + SimpleLoss + FusedSGD through the C ABI
against (a) the reference-generated fixtures tests/golden/net64.npz and net512.npz and
(b) the oracle run on the same seeded inputs.

Tolerance: 1e-4 relative on fp32 logits (north_star), stated per assertion; argmax masks are
compared bit-exactly on every pixel whose reference top-2 logit margin is >= 1e-3 (pixels
below that margin are ties at fp32 resolution: SURVEY.md §7.2).

Gradients: LeakyReLU's derivative is discontinuous, so an element whose pre-activation is
within fp32 rounding of 0 can take the other branch than in the reference run (measured on
MI355X at 64x64: exactly 1 of 65,536 elements of encoder_stages.2.block.0, with the
reference's own fp32 gradients deviating from an fp64 run by the same 4e-4..2e-3).  Gradient
checks are therefore norm-wise (||g - ref|| / ||ref||) plus "all but a few sampled entries"."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import unet_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def sample_idx(numel, k=64, seed=5):
    rng = np.random.Generator(np.random.PCG64(seed + numel))
    return np.sort(rng.choice(numel, size=min(k, numel), replace=False))


def build(ua, g):
    sd = O.fill_state_dict(int(g["seed_w"]))
    model = ua.UNet()
    model.load_state_dict(sd)
    model = model.to(DEV)
    img, tgt = O.synthetic_batch(int(g["seed_x"]), int(g["n"]), int(g["hw"]), int(g["hw"]))
    return model, sd, img.to(DEV), tgt.to(DEV)


def unpack_argmax(g, shape):
    bits = np.unpackbits(g["eval_argmax"])[: 2 * int(np.prod(shape))].reshape(-1, 2)
    return (bits[:, 0] * 2 + bits[:, 1]).reshape(shape).astype(np.uint8)


def run_golden(ua, g, full, precision="fp32", fused=True):
    model, sd0, img, tgt = build(ua, g)
    model.matmul_precision = precision
    model.fused_pipeline = fused
    n, hw = int(g["n"]), int(g["hw"])
    names = [str(s) for s in g["param_names"]]
    assert names == [k for k, _ in model.named_parameters()]

    # ---- eval forward: logits within 1e-4 relative, argmax bit-exact off the tie pixels
    model.eval()
    with torch.no_grad():
        le = model(img)
    assert le.shape == (n, 3, hw, hw) and le.dtype == torch.float32
    if full:
        e = relerr(le, torch.from_numpy(g["eval_logits"]))
    else:
        e = relerr(le[:, :, ::16, ::16], torch.from_numpy(g["eval_logits_s16"]))
    assert e <= 1e-4, f"eval logits rel err {e:.3e}"
    am = le.argmax(dim=1).to(torch.uint8).cpu().numpy()
    ref_am = unpack_argmax(g, am.shape)
    assert hashlib.sha256(ref_am.tobytes()).hexdigest() == str(g["eval_argmax_sha256"])
    low = np.unpackbits(g["eval_lowmargin"])[: am.size].reshape(am.shape).astype(bool)
    assert low.mean() < 0.01
    assert np.array_equal(am[~low], ref_am[~low]), "argmax differs on a non-tie pixel"

    # ---- three train steps with injected dropout masks
    model.train()
    opt = ua.create_optimizer(model)
    lossf = ua.get_loss_function()
    steps = 3
    for s in range(steps):
        model.dropout_mask_override = O.draw_dropout_masks(int(g["seed_drop"]) + s, n)
        if s == 0:
            opt.zero_grad()
            logits = model(img)
            loss = lossf(logits, tgt)
            loss.backward()
            if full:
                e = relerr(logits, torch.from_numpy(g["train_logits"]))
            else:
                e = relerr(logits[:, :, ::16, ::16], torch.from_numpy(g["train_logits_s16"]))
            assert e <= 1e-4, f"train logits rel err {e:.3e}"
            bad = []
            for i, (k, p) in enumerate(model.named_parameters()):
                gk = p.grad.reshape(-1)
                ref_norm = float(g[f"gnorm_{i}"])
                got = gk.double().norm().item()
                if ref_norm < 1e-4:       # conv biases under InstanceNorm: exact 0 up to rounding
                    if got >= 1e-3:
                        bad.append(f"{k}: |grad| {got:.3e} should be ~0")
                    continue
                if abs(got - ref_norm) > 5e-3 * ref_norm:
                    bad.append(f"{k}: grad norm {got} vs {ref_norm}")
                idx = torch.from_numpy(sample_idx(gk.numel())).to(DEV)
                ref_s = torch.from_numpy(g[f"gsamp_{i}"])
                err = (gk[idx].cpu() - ref_s).abs()
                # the reference's own fp32 run deviates from fp64 by 1e-3..1.5e-2 of max|g| at
                # 512x512 (profiles/r01_grad_accuracy_vs_fp64_512.txt): LeakyReLU tie flips
                tol = 1e-2 * max(ref_s.abs().max().item(), ref_norm / gk.numel() ** 0.5)
                # an absolute cap on the outliers a few tie flips can cause: at most 3 + 1 % of
                # the samples beyond tol, none beyond 5 tol; the element-by-element checks are
                # test_tie_free_network_gradients_per_element (slope 1) and
                # test_default_slope_gradients_per_element_with_the_hip_branch_pattern
                n_out = int((err > tol).sum().item())
                if n_out > 3 + 0.01 * err.numel() or err.max().item() > 5 * tol:
                    bad.append(f"{k}: {n_out} of {err.numel()} sampled grads beyond tol, max err "
                               f"{err.max().item():.3e} (tol {tol:.3e})")
            assert not bad, "\n".join(bad)
            opt.step()
        else:
            loss = ua.train_step(model, opt, lossf, img, tgt)
        ref_loss = float(g[f"loss_{s}"])
        # The 3-step trajectory is chaotic in fp32: the reference's own fp32 run leaves an fp64
        # run of the same code by 1.4e-4 / 1.9e-2 relative loss at steps 1 / 2 on the 64x64
        # fixture (the HIP path stays closer: profiles/r01_trajectory_vs_fp64_64.txt), so only
        # step 0 is held to the tight tolerance.
        tol = (2e-4, 2e-3, 5e-2)[s]
        assert abs(loss.item() - ref_loss) <= tol * abs(ref_loss), \
            f"step {s}: loss {loss.item()} vs {ref_loss}"
        if s == 0:
            for i, (k, p) in enumerate(model.named_parameters()):
                if float(g[f"gnorm_{i}"]) < 1e-4:
                    continue   # conv biases: the update is rounding noise of a ~0 gradient
                d = (p.detach().cpu() - sd0[k]).double().norm().item()
                ref_d = float(g[f"dnorm_{s}_{i}"])
                assert abs(d - ref_d) <= 5e-3 * ref_d + 1e-7, f"step {s} {k}: |dp| {d} vs {ref_d}"


def test_net64_golden(ua, golden):
    run_golden(ua, golden("net64"), full=True)


def test_net512_golden(ua, golden):
    run_golden(ua, golden("net512"), full=False)


@pytest.mark.parametrize("fixture,full", [("net64", True), ("net512", False)])
def test_golden_on_the_standalone_pipeline(ua, golden, fixture, full):
    """`fused_pipeline = False`: separate statistics / apply passes, activated tensors in HBM
    (the pipeline the bf16 / bf16x3 operand modes use) against the same reference fixtures."""
    run_golden(ua, golden(fixture), full=full, fused=False)


@pytest.mark.parametrize("fixture,full", [("net64", True), ("net512", False)])
def test_golden_in_split_bf16_mode(ua, golden, fixture, full):
    """matmul_precision="bf16x3" (fp32 operands as three bf16 terms, six products on the bf16
    matrix cores, fp32 accumulation) is held to the SAME reference fixtures and tolerances as
    the fp32 matrix-core path: 1e-4 on logits, bit-exact argmax off the tie pixels, gradients,
    loss and first update."""
    run_golden(ua, golden(fixture), full=full, precision="bf16x3")


@pytest.mark.parametrize("fused", [True, False])
def test_tie_free_network_gradients_per_element(ua, golden, fused):
    """Whole-network gradients held element by element.  With nonlin_kwargs negative_slope = 1.0
    the reference's LeakyReLU is the identity, so there are no activation ties whose fp32 branch
    could flip: every one of the 90 gradient tensors must agree with the reference fixture
    (tests/golden/net64_slope1.npz, recorded from Our_UNet/models/unet.py) on its 256 recorded
    entries and with the oracle on EVERY entry to 1e-4 of the tensor's max magnitude.  (The
    LeakyReLU branch itself is covered by the kernel tests and the default-slope goldens.)"""
    g = golden("net64_slope1")
    n, hw = int(g["n"]), int(g["hw"])
    sd0 = O.fill_state_dict(int(g["seed_w"]))
    model = ua.UNet(nonlin_kwargs={"negative_slope": 1.0, "inplace": True})
    model.load_state_dict(sd0)
    model = model.to(DEV).train()
    model.fused_pipeline = fused
    img, tgt = O.synthetic_batch(int(g["seed_x"]), n, hw, hw)
    masks = O.draw_dropout_masks(int(g["seed_drop"]), n)
    model.dropout_mask_override = masks
    logits = model(img.to(DEV))
    loss = ua.get_loss_function()(logits, tgt.to(DEV))
    loss.backward()
    assert relerr(logits, torch.from_numpy(g["train_logits"])) <= 1e-4
    assert abs(loss.item() - float(g["loss_0"])) <= 2e-4 * abs(float(g["loss_0"]))
    osd = O.leaf_state_dict(sd0)
    _, _, ograds = O.train_step(osd, [None] * len(osd), img, tgt, masks, slope=1.0)
    bad = []
    for i, (k, p) in enumerate(model.named_parameters()):
        ref_norm = float(g[f"gnorm_{i}"])
        gk = p.grad.reshape(-1).cpu()
        if ref_norm < 1e-4:      # conv biases under InstanceNorm: exact 0 up to rounding
            if gk.double().norm().item() >= 1e-3:
                bad.append(f"{k}: should be ~0")
            continue
        scale = ograds[k].abs().max().item()
        e_all = (gk - ograds[k].reshape(-1)).abs().max().item() / scale
        idx = torch.from_numpy(sample_idx(gk.numel(), k=256))
        e_fix = (gk[idx] - torch.from_numpy(g[f"gsamp_{i}"])).abs().max().item() / scale
        if e_all > 1e-4 or e_fix > 1e-4 or abs(gk.double().norm().item() - ref_norm) > 1e-4 * ref_norm:
            bad.append(f"{k}: every-element err {e_all:.2e}, fixture entries {e_fix:.2e}")
    assert not bad, "\n".join(bad)


def _winograd_selection(ua, n, hw):
    """Which Winograd kernels the fused fp32 step picks at batch n, hw x hw (the library's own
    predicates - the ones ops.py / the C entry points consult)."""
    L = ua.lib()
    sel = {
        "c32 forward / data gradient (conv_wino32q)":
            bool(L.unet_conv_c32_is_winograd(n, hw, hw, 32, 32, 1)),
        "c32 up-sampling layer (conv_wino_up32)":
            bool(L.unet_conv_up_c32_is_winograd(n, hw, hw, 64, 32, 32)),
        "c32 weight gradient (conv_wgrad_wino32)":
            bool(L.unet_conv3x3_bwd_weight_is_winograd(n, hw, hw, 32, 32, 1)),
    }
    for c, s in ((64, 2), (128, 4), (256, 8), (512, 16)):
        h = hw // s
        sel[f"{c} channels forward / data gradient (conv_wino)"] = \
            bool(L.unet_conv_wino_supported(n, h, h, c, 0, c))
        sel[f"{c} channels weight gradient (conv_wgrad_wino)"] = \
            bool(L.unet_conv3x3_bwd_weight_is_winograd(n, h, h, c, c, 1))
        sel[f"{c} channels up-sampling loader (conv_wino UP)"] = \
            bool(L.unet_conv_up_wino_supported(n, h, h, min(2 * c, 512), c, c))
    return sel


@pytest.mark.parametrize("hw", [64, 128, 512])
def test_default_slope_gradients_per_element_with_the_hip_branch_pattern(ua, hw):
    """Whole-network gradients at the REFERENCE slope 0.01, element by element, away from ties.

    Two fp32 implementations of the network land on different sides of z = 0 for the few
    pre-activations that are at rounding level (|z| ~ 1e-6), and one such flip changes
    lrelu'(z) from 1 to 0.01 at that pixel - which is why the default-slope goldens can only be
    held norm-wise.  Here the oracle is evaluated with the HIP run's branch decisions on exactly
    those risky elements (|z_oracle| < 1e-4; the decision is sign(fma(y, gamma*rstd,
    beta - mean*gamma*rstd)) of the captured raw outputs, the expression of the backward
    kernels), so both differentiate the same piecewise-linear function: away from the risky
    elements the two must agree on every branch, and ALL 90 gradients are then held per element
    to 2e-4 of each tensor's max magnitude (measured: <= 1.2e-4, on the 512-channel layers whose
    InstanceNorm runs over 4 or 16 pixels at these image sizes; every other tensor <= 1e-4) - the check that `z > 0 ? 1 : slope` in the
    InstanceNorm-backward kernels, the data-gradient epilogues and the activation-on-load of the
    weight gradients is right at the reference slope.

    hw = 512 (round 4): the size at which the step really runs its Winograd kernels - the
    one-chunk 32-channel family (conv_wino32q, conv_wino_up32, conv_wgrad_wino32) and the 64- and
    128-channel conv_wino / conv_wgrad_wino launches; asserted below, so the reference-slope
    branch of their BSTATS epilogues and of their activation-on-load is held per element."""
    n = 2
    if hw == 512:
        sel = _winograd_selection(ua, n, hw)
        must = [k for k in sel if k.startswith(("c32", "64 ", "128 "))]
        assert all(sel[k] for k in must), {k: sel[k] for k in must}
    sd0 = O.fill_state_dict(29)
    model = ua.UNet()
    model.load_state_dict(sd0)
    model = model.to(DEV).train()
    img, tgt = O.synthetic_batch(13, n, hw, hw)
    masks = O.draw_dropout_masks(17, n)
    model.dropout_mask_override = masks
    model._debug_forward = []
    logits = model(img.to(DEV))
    loss = ua.get_loss_function()(logits, tgt.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    branches = []
    params = dict(model.named_parameters())
    for (name, y, st), row in zip(model._debug_forward, O.layer_table()):
        prefix, ci, ni = row[0], row[1], row[2]
        gamma = params[f"{prefix}.{ni}.weight"].detach().cpu()
        beta = params[f"{prefix}.{ni}.bias"].detach().cpu()
        mean, rstd = st[0].cpu(), st[1].cpu()                       # [N, C]
        al = gamma[None, :] * rstd                                   # fp32, one rounding
        b0 = (beta[None, :].double() - mean.double() * al.double()).float()
        z = y.detach().cpu().double() * al.double()[:, None, None, :] + b0.double()[:, None, None, :]
        branches.append((z > 0).permute(0, 3, 1, 2).contiguous())   # NCHW like the oracle
    assert len(branches) == len(O.layer_table())
    model._debug_forward = None

    # hw = 512: the oracle's arithmetic in fp64.  The InstanceNorm gradients there are sums of
    # 524,288 signed terms per channel; accumulated in fp32 - by the oracle's CPU kernels as much
    # as by the HIP ones - such a sum carries a few 1e-4 of the largest entry (first run of this
    # case against the fp32 oracle: 2.4e-4 / 3.8e-4 on decoder_stages.4's first norm, everything
    # else <= 2e-4), so the fp32 oracle cannot referee 2e-4 at this size; its fp64 evaluation can
    # (same precedent: test_gradient_accuracy_vs_fp64).
    dt = torch.float64 if hw == 512 else torch.float32
    osd = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd0.items()}
    diag = {}
    ologits = O.unet_forward(osd, img.to(dt), [m.to(dt) for m in masks], branches=branches,
                             tie_eps=1e-4, tie_diag=diag)
    if dt == torch.float64:
        oloss = torch.nn.functional.cross_entropy(ologits, tgt, weight=O.class_weights(tgt).to(dt),
                                                  ignore_index=255) + O.dice_loss(ologits, tgt)
    else:
        oloss = O.simple_loss(ologits, tgt)
    oloss.backward()
    assert diag["disagree_away_from_ties"] == 0, diag
    assert diag["risky"] > 0      # the hook did see elements near zero ...
    assert relerr(logits, ologits.detach()) <= 1e-4
    bad = []
    for k, p in model.named_parameters():
        og = osd[k].grad
        gk = p.grad.detach().cpu().to(og.dtype)
        scale = og.abs().max().item()
        if scale < 1e-4:         # conv biases under InstanceNorm: exact 0 up to rounding
            if gk.double().norm().item() >= 1e-3:
                bad.append(f"{k}: should be ~0")
            continue
        e = (gk - og).abs().max().item() / scale
        if e > 2e-4:
            bad.append(f"{k}: every-element err {e:.2e}")
    assert not bad, f"{diag}\n" + "\n".join(bad)


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16"])
def test_train_step_is_run_to_run_deterministic(ua, mode):
    """No float atomics anywhere: every partial sum (split-K slabs, K groups, statistics and
    backward-reduction tiles, loss reductions) is combined in a fixed order, so two train steps
    from the same state give bit-identical logits, loss, gradients and updated parameters - in
    all three operand modes, at a size where every kernel family is in play."""
    sd0 = O.fill_state_dict(21)
    img, tgt = O.synthetic_batch(4, 2, 256, 256)
    masks = O.draw_dropout_masks(8, 2)
    outs = []
    for _ in range(2):
        model = ua.UNet()
        model.load_state_dict(sd0)
        model = model.to(DEV).train()
        model.matmul_precision = mode
        model.dropout_mask_override = masks
        opt = ua.create_optimizer(model)
        logits = model(img.to(DEV))
        loss = ua.get_loss_function()(logits, tgt.to(DEV))
        opt.zero_grad()
        loss.backward()
        arena, garena = model.flat_parameters()
        g = garena.clone()
        opt.step()
        outs.append((logits.detach().clone(), loss.item(), g, arena.clone()))
        del model, opt
    a, b = outs
    assert torch.equal(a[0], b[0]) and a[1] == b[1]
    assert torch.equal(a[2], b[2]), "gradients differ between two identical runs"
    assert torch.equal(a[3], b[3])


EDGE_INPUTS = {
    "zeros": lambda g: torch.zeros(2, 3, 64, 64),
    "constant": lambda g: torch.full((2, 3, 64, 64), 2.5),
    "huge": lambda g: torch.randn(2, 3, 64, 64, generator=g) * 1e4,
    "tiny": lambda g: torch.randn(2, 3, 64, 64, generator=g) * 1e-6,
}


@pytest.mark.parametrize("kind", list(EDGE_INPUTS))
def test_degenerate_inputs_with_fp64_attribution(ua, kind):
    """All-zero, constant, huge and tiny images: the first InstanceNorm sees (near-)constant
    planes, rstd = 1/sqrt(eps) amplifies the rounding noise of the stem by ~300, and ANY fp32
    evaluation leaves the exact result by more than the 1e-4 of well-conditioned inputs (round 2
    measured 2.8e-4 / 4.1e-4 against the fp32 oracle with no attribution).  Here the error is
    measured against the oracle evaluated in fp64 and held to 3x the fp32 oracle's own error
    (plus 2e-5 of max |logit|): the HIP path may not be worse-conditioned than the reference's
    arithmetic on such inputs, and everything stays finite."""
    g = torch.Generator().manual_seed(11)
    img = EDGE_INPUTS[kind](g)
    sd0 = O.fill_state_dict(5)
    _, tgt = O.synthetic_batch(1, 2, 64, 64)
    masks = O.draw_dropout_masks(3, 2)
    model = ua.UNet()
    model.load_state_dict(sd0)
    model = model.to(DEV).train()
    model.dropout_mask_override = masks
    logits = model(img.to(DEV))
    loss = ua.get_loss_function()(logits, tgt.to(DEV))
    loss.backward()
    assert torch.isfinite(logits).all() and np.isfinite(loss.item())
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())
    with torch.no_grad():
        l32 = O.unet_forward(sd0, img, masks)
        sd64 = {k: v.double() for k, v in sd0.items()}
        l64 = O.unet_forward(sd64, img.double(), [None if m is None else m.double() for m in masks])
    scale = l64.abs().max().item()
    e_hip = (logits.detach().cpu().double() - l64).abs().max().item() / scale
    e_ref = (l32.double() - l64).abs().max().item() / scale
    assert e_hip <= 3.0 * e_ref + 2e-5, \
        f"{kind}: HIP {e_hip:.2e} vs fp32 oracle {e_ref:.2e} of max |logit| from the fp64 result"


@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_graph_captured_train_step_equals_the_eager_walk(ua, mode):
    """`GraphedTrainStep` replays the train step from one HIP graph.  With the dropout masks
    injected (static tensors) four replayed steps - the third and fourth after a learning-rate
    change, which the graph must follow through the device-side hyper-parameters without being
    re-captured - leave parameters, momentum and losses bit-identical to four eager steps, and
    the throw-away capture steps leave no trace in the model."""
    sd0 = O.fill_state_dict(23)
    img, tgt = O.synthetic_batch(6, 2, 128, 128)
    img2, tgt2 = O.synthetic_batch(7, 2, 128, 128)
    masks = [m.to(DEV) if m is not None else None for m in O.draw_dropout_masks(5, 2)]
    outs = []
    for graphed in (False, True):
        model = ua.UNet()
        model.load_state_dict(sd0)
        model = model.to(DEV).train()
        model.matmul_precision = mode
        model.dropout_mask_override = masks
        opt = ua.create_optimizer(model)
        lossf = ua.get_loss_function()
        a, b = (img.to(DEV), tgt.to(DEV)), (img2.to(DEV), tgt2.to(DEV))
        if graphed:
            step = ua.GraphedTrainStep(model, opt, lossf, *a)
            arena, _ = model.flat_parameters()
            assert torch.equal(arena.cpu(), outs[0][0]), "capture changed the parameters"
        else:
            arena, _ = model.flat_parameters()
            outs.append((arena.detach().cpu().clone(),))

            def step(x, y, model=model, opt=opt, lossf=lossf):
                return ua.train_step(model, opt, lossf, x, y)
        losses = []
        for k, batch in enumerate((a, b, a, b)):
            if k == 2:
                opt.param_groups[0]["lr"] *= 0.5
            losses.append(step(*batch).item())
        torch.cuda.synchronize()
        arena, _ = model.flat_parameters()
        outs.append((losses, arena.detach().cpu().clone(), opt._flat_buf.detach().cpu().clone()))
        del model, opt
    _, eager, graph = outs
    assert eager[0] == graph[0], f"losses differ: {eager[0]} vs {graph[0]}"
    assert torch.equal(eager[1], graph[1]), "parameters differ after four steps"
    assert torch.equal(eager[2], graph[2]), "momentum differs after four steps"


def test_graphed_step_keeps_momentum_loaded_from_a_checkpoint(ua, tmp_path):
    """Resume, then capture: FusedSGD.load_state_dict leaves the loaded momentum in
    state[p]['momentum_buffer'] until the flat arena adopts it.  GraphedTrainStep must adopt it
    BEFORE its throw-away steps and restore exactly that (round 3 zeroed it: mu = 0.99 of history
    silently lost).  Two replayed steps after the resume == two eager steps after the resume."""
    sd0 = O.fill_state_dict(41)
    img, tgt = O.synthetic_batch(12, 2, 64, 64)
    img, tgt = img.to(DEV), tgt.to(DEV)
    masks = [m.to(DEV) if m is not None else None for m in O.draw_dropout_masks(15, 2)]
    a = ua.UNet()
    a.load_state_dict(sd0)
    a = a.to(DEV).train()
    a.dropout_mask_override = masks
    opt_a = ua.create_optimizer(a)
    lossf = ua.get_loss_function()
    for _ in range(2):
        ua.train_step(a, opt_a, lossf, img, tgt)
    path = ua.save_checkpoint(a, opt_a, None, epoch=0, best_dice=0.0, output_dir=tmp_path)
    assert float(opt_a._flat_buf.abs().max()) > 0
    outs = []
    for graphed in (False, True):
        b = ua.UNet().to(DEV).train()
        b.dropout_mask_override = masks
        opt_b = ua.create_optimizer(b)
        ua.load_checkpoint(path, b, opt_b, None, device=DEV)
        assert opt_b._flat_buf is None
        if graphed:
            step = ua.GraphedTrainStep(b, opt_b, lossf, img, tgt)
            assert torch.equal(opt_b._flat_buf, opt_a._flat_buf), "capture lost the loaded momentum"
        else:
            def step(x, y, b=b, opt_b=opt_b):
                return ua.train_step(b, opt_b, lossf, x, y)
        losses = [step(img, tgt).item() for _ in range(2)]
        arena, _ = b.flat_parameters()
        outs.append((losses, arena.detach().clone(), opt_b._flat_buf.detach().clone()))
        # a second GraphedTrainStep on the same optimizer keeps the hyper-parameter tensor the
        # first graph reads, and an eager step afterwards runs on the CURRENT learning rate
        if graphed:
            hyper = opt_b._hyper
            ua.GraphedTrainStep(b, opt_b, lossf, img, tgt)
            assert opt_b._hyper is hyper
            opt_b.param_groups[0]["lr"] = 0.0
            opt_b.param_groups[0]["weight_decay"] = 0.0
            before = arena.detach().clone()
            ua.train_step(b, opt_b, lossf, img, tgt)          # eager, device-side hyper still on
            assert torch.equal(b.flat_parameters()[0], before), "eager step used a stale lr"
    (le, pe, be), (lg, pg, bg) = outs
    assert le == lg and torch.equal(pe, pg) and torch.equal(be, bg)


def test_graph_captured_step_draws_fresh_dropout_masks(ua):
    """Without injected masks the Bernoulli draw is part of the graph: consecutive replays must
    use different masks (torch's graph-safe generator advances the Philox offset per replay)."""
    torch.manual_seed(3)
    model = ua.create_model(DEV).train()
    opt = ua.create_optimizer(model)
    opt.param_groups[0]["lr"] = 0.0          # frozen weights: only the masks differ
    opt.param_groups[0]["weight_decay"] = 0.0
    img, tgt = O.synthetic_batch(6, 2, 64, 64)
    step = ua.GraphedTrainStep(model, opt, ua.get_loss_function(), img.to(DEV), tgt.to(DEV))
    losses = {round(step(step.images, step.masks).item(), 7) for _ in range(4)}
    assert len(losses) >= 3, f"replays reused the dropout masks: {losses}"


SWEEP = [(1, 64, 64), (3, 96, 160), (2, 128, 256), (1, 320, 192), (5, 256, 256), (2, 384, 384),
         (1, 512, 512)]


@pytest.mark.parametrize("shape", SWEEP, ids=[f"{n}x{h}x{w}" for n, h, w in SWEEP])
def test_shape_sweep_fused_vs_standalone_pipeline(ua, shape):
    """The fused pipeline picks its kernels by shape (patch / 32-channel / stride-2 patch /
    K-group gather-GEMM / up-sampling loader, each with tiling conditions).  Over batch sizes and
    image sizes that land on different combinations of them, the tie-free network
    (negative_slope = 1: no LeakyReLU branch to flip) must agree with the stand-alone pipeline -
    whose kernels are held to the oracle one by one - on the logits and on EVERY element of all
    90 gradients to 1e-4 of each tensor's max magnitude."""
    n, h, w = shape
    sd0 = O.fill_state_dict(17)
    img, tgt = O.synthetic_batch(3, n, h, w)
    masks = O.draw_dropout_masks(9, n)
    outs = []
    for fused in (True, False):
        model = ua.UNet(nonlin_kwargs={"negative_slope": 1.0, "inplace": True})
        model.load_state_dict(sd0)
        model = model.to(DEV).train()
        model.fused_pipeline = fused
        model.dropout_mask_override = masks
        logits = model(img.to(DEV))
        loss = ua.get_loss_function()(logits, tgt.to(DEV))
        loss.backward()
        outs.append((logits.detach().cpu(), loss.item(),
                     {k: p.grad.detach().cpu() for k, p in model.named_parameters()}))
        del model
    (lf, lossf, gf), (ls, losss, gs) = outs
    assert relerr(lf, ls) <= 1e-4
    assert abs(lossf - losss) <= 2e-4 * abs(losss)
    bad = []
    for k in gs:
        scale = gs[k].abs().max().item()
        if scale < 1e-6:      # conv biases under InstanceNorm: exact 0 up to rounding
            if gf[k].abs().max().item() > 1e-3:
                bad.append(f"{k}: should be ~0")
            continue
        e = (gf[k] - gs[k]).abs().max().item() / scale
        if e > 1e-4:
            bad.append(f"{k}: {e:.2e}")
    assert not bad, "\n".join(bad)


def test_instnorm_backward_applied_on_load_matches_the_elementwise_pass(ua):
    """`UNet.fold_instnorm_backward` (off by default: measured slower): the Winograd
    data gradients of the stride-1 layers form dL/dy from (g, y) in their loaders and write it
    for the weight gradients - no in_bwd_apply launch for those layers.  At 2 x 512 x 512 the 64-
    and 128-channel layers and the skip halves of the last two decoder stages take that route;
    every gradient element must agree with the elementwise pass to 1e-4 of the tensor's max
    (same LeakyReLU branch expression, so no tie can flip between the two)."""
    sd0 = O.fill_state_dict(23)
    img, tgt = O.synthetic_batch(4, 2, 512, 512)
    masks = O.draw_dropout_masks(6, 2)
    outs = []
    for fold in (False, True):
        model = ua.UNet()
        model.load_state_dict(sd0)
        model = model.to(DEV).train()
        model.fold_instnorm_backward = fold
        model.dropout_mask_override = masks
        loss = ua.get_loss_function()(model(img.to(DEV)), tgt.to(DEV))
        loss.backward()
        outs.append({k: p.grad.detach().cpu() for k, p in model.named_parameters()})
        del model
    bad = []
    for k in outs[0]:
        scale = outs[0][k].abs().max().item()
        if scale < 1e-6:
            continue
        e = (outs[1][k] - outs[0][k]).abs().max().item() / scale
        if e > 1e-4:
            bad.append(f"{k}: {e:.2e}")
    assert not bad, "\n".join(bad)


def test_net_vs_oracle_random_init(ua):
    """Reference-style random init (Kaiming weights, zero biases, unit gamma), 96x64 input,
    train mode with oracle-drawn masks: logits, loss and every gradient against the oracle."""
    sd0 = O.fill_state_dict(99, trained_like=False)
    model = ua.UNet()
    model.load_state_dict(sd0)
    model = model.to(DEV).train()
    img, tgt = O.synthetic_batch(5, 3, 96, 64)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(3, 3, 96, 64, generator=g)
    tgt = tgt[:, :96, :64].contiguous()
    masks = O.draw_dropout_masks(123, 3)
    osd = O.leaf_state_dict(sd0)
    ologits = O.unet_forward(osd, img, masks)
    oloss = O.simple_loss(ologits, tgt)
    oloss.backward()
    model.dropout_mask_override = masks
    logits = model(img.to(DEV))
    loss = ua.SimpleLoss()(logits, tgt.to(DEV))
    loss.backward()
    assert relerr(logits, ologits.detach()) <= 1e-4
    assert abs(loss.item() - oloss.item()) <= 1e-4 * abs(oloss.item())
    bad = []
    num = den = 0.0
    for k, p in model.named_parameters():
        ref = osd[k].grad.double()
        got = p.grad.cpu().double()
        if ref.abs().max() < 1e-4:
            if got.abs().max().item() >= 1e-3:
                bad.append(f"{k}: should be ~0")
            continue
        e = ((got - ref).norm() / ref.norm()).item()
        num += ((got - ref) ** 2).sum().item()
        den += (ref ** 2).sum().item()
        if e > 3e-2:
            bad.append(f"{k}: norm-wise grad rel err {e:.3e}")
    assert not bad, "\n".join(bad)
    assert (num / den) ** 0.5 <= 5e-3, f"whole-gradient rel err {(num / den) ** 0.5:.3e}"


def test_gradient_accuracy_vs_fp64(ua):
    """Principled accuracy check: the whole gradient of the HIP path - fp32 matrix cores and
    split-bf16 mode - must be as close to an fp64 run of the oracle as the oracle's own fp32
    run is.  A single sample is a lottery: the error is dominated by which LeakyReLU inputs
    change sign under fp32 rounding (a handful of elements, each worth an O(1) change of its
    gradient); over 12 batches all three fp32 paths range over 1e-5 ... 1e-2 with medians
    1.4e-3 / 1.4e-3 / 1.7e-3 (profiles/r01_d_tie_flip_lottery_vs_fp64.txt).  So seven batches
    are compared by their medians (x3 slack over the reference's, floor 3e-3 = twice the
    reference's own median), no sample may exceed 3e-2 (a wrong kernel gives O(0.1 - 1)), and
    the logits - which have no such lottery - must stay within 4x the reference's error."""
    sd0 = O.fill_state_dict(2024)
    errs = {"ref32": [], "fp32": [], "bf16x3": []}
    lerr = {"ref32": [], "fp32": [], "bf16x3": []}
    for trial in range(7):
        img, tgt = O.synthetic_batch(1234 + trial, 2, 64, 64)
        masks = O.draw_dropout_masks(77 + trial, 2)

        def run(dtype):
            osd = {k: v.to(dtype).clone().requires_grad_(True) for k, v in sd0.items()}
            lg = O.unet_forward(osd, img.to(dtype), [m.to(dtype) for m in masks])
            w = O.class_weights(tgt).to(dtype)
            loss = torch.nn.functional.cross_entropy(lg, tgt, weight=w, ignore_index=255) \
                + O.dice_loss(lg, tgt)
            loss.backward()
            return lg.detach(), torch.cat([v.grad.reshape(-1).double() for v in osd.values()])

        def hip(precision):
            model = ua.UNet()
            model.load_state_dict(sd0)
            model = model.to(DEV).train()
            model.matmul_precision = precision
            model.dropout_mask_override = masks
            logits = model(img.to(DEV))
            ua.SimpleLoss()(logits, tgt.to(DEV)).backward()
            return logits.detach(), torch.cat([p.grad.reshape(-1).double().cpu()
                                               for p in model.parameters()])

        l64, g64 = run(torch.float64)
        for name, (lg, g) in (("ref32", run(torch.float32)), ("fp32", hip("fp32")),
                              ("bf16x3", hip("bf16x3"))):
            errs[name].append(((g.double() - g64).norm() / g64.norm()).item())
            lerr[name].append(relerr(lg, l64))
    med = lambda v: sorted(v)[len(v) // 2]
    for mode in ("fp32", "bf16x3"):
        msg = f"{mode}: gradient error vs fp64 {errs[mode]}, reference fp32 {errs['ref32']}"
        assert med(errs[mode]) <= max(3 * med(errs["ref32"]), 3e-3), msg
        assert max(errs[mode]) <= 3e-2, msg
        assert med(lerr[mode]) <= max(4 * med(lerr["ref32"]), 2e-5), \
            f"{mode}: logits error vs fp64 {lerr[mode]}, reference fp32 {lerr['ref32']}"


def test_state_dict_roundtrip_and_eval_determinism(ua):
    model = ua.UNet().to(DEV).eval()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert len(sd) == 90
    x = torch.randn(1, 3, 64, 64, device=DEV)
    with torch.no_grad():
        y1 = model(x)
        y2 = model(x)
    assert torch.equal(y1, y2)
    other = ua.UNet().to(DEV).eval()
    other.load_state_dict(sd)
    with torch.no_grad():
        assert torch.equal(other(x), y1)


def test_cpu_input_fails_loudly(ua):
    model = ua.UNet()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.randn(1, 3, 64, 64))


def test_train_step_drives_loss_down(ua):
    torch.manual_seed(0)
    model = ua.create_model(DEV).train()
    opt = ua.create_optimizer(model)
    lossf = ua.get_loss_function()
    img, tgt = O.synthetic_batch(3, 2, 64, 64)
    img, tgt = img.to(DEV), tgt.to(DEV)
    losses = [ua.train_step(model, opt, lossf, img, tgt).item() for _ in range(12)]
    assert all(np.isfinite(losses))
    assert min(losses[-3:]) < losses[0]


def test_bf16_matmul_mode_tracks_fp32(ua):
    """matmul_precision="bf16" on the STAND-ALONE pipeline (fused_pipeline = False: bf16 MFMA
    operands, fp32 tensors in HBM; the default bf16 mode with bf16 tensors in HBM is covered by
    tests/test_bf16_gpu.py against the oracle's emulation of its rounding points).
    BASELINE config 4: no reference numerics exist for it (the
    reference's AMP is fp16 autocast), so it is held to a bf16-sized tolerance against the
    fp32 path on the same weights, inputs and masks, and must train.

    The bf16 kernels themselves are checked exactly (fp32 conv of bf16-rounded operands) in
    test_kernels_gpu.py.  End to end this 23-layer InstanceNorm/LeakyReLU stack amplifies
    operand rounding: fp32 vs fp64 logits already differ by ~1e-5 at eps 6e-8, and bf16
    (eps 2^-9) measured 5 % rms on logits, loss within 0.1 %, gradient cosine 0.95 at
    256x256 and 512x512 (tests/tools/debug_bf16.py)."""
    sd0 = O.fill_state_dict(2024)
    img, tgt = O.synthetic_batch(1234, 2, 256, 256)
    masks = O.draw_dropout_masks(77, 2)
    outs = {}
    for mode in ("fp32", "bf16"):
        model = ua.UNet()
        model.load_state_dict(sd0)
        model = model.to(DEV).train()
        model.matmul_precision = mode
        model.fused_pipeline = mode == "fp32"
        model.dropout_mask_override = masks
        logits = model(img.to(DEV))
        loss = ua.SimpleLoss()(logits, tgt.to(DEV))
        loss.backward()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).double().cpu()
        outs[mode] = (logits.detach().cpu().double(), loss.item(), g)
    d = outs["bf16"][0] - outs["fp32"][0]
    e = (d.norm() / outs["fp32"][0].norm()).item()     # rms-relative over all logits
    assert 1e-5 < e < 0.1, f"bf16 logits rms rel err {e:.3e}"
    assert abs(outs["bf16"][1] - outs["fp32"][1]) < 1e-2 * abs(outs["fp32"][1])
    ga, gb = outs["bf16"][2], outs["fp32"][2]
    cos = (ga @ gb / (ga.norm() * gb.norm())).item()
    assert cos > 0.9, f"bf16 gradient cosine {cos:.3f}"
    # and it trains
    model = ua.create_model(DEV).train()
    model.matmul_precision = "bf16"
    model.fused_pipeline = False
    opt = ua.create_optimizer(model)
    lossf = ua.get_loss_function()
    img, tgt = O.synthetic_batch(3, 2, 64, 64)
    losses = [ua.train_step(model, opt, lossf, img.to(DEV), tgt.to(DEV)).item() for _ in range(12)]
    assert all(np.isfinite(losses)) and min(losses[-3:]) < losses[0]


def test_clip_unet_in_split_bf16_mode(ua, golden):
    """CLIP_UNet with matmul_precision = "bf16x3" on the fused pipeline (the 1x1 fusion layer
    stays on the fp32 kernels, the 3x3 layers run the split kernels): the reference fixture's
    logits and loss to the fp32 tolerances, gradient norms to 5e-3."""
    g = golden("clip64")
    n, hw, clip_dim = int(g["n"]), int(g["hw"]), int(g["clip_dim"])
    sd0 = O.fill_state_dict(int(g["seed_w"]), clip_dim=clip_dim)
    model = ua.CLIPUNet(with_clip_features=True, clip_dim=clip_dim)
    model.load_state_dict(sd0)
    model = model.to(DEV).train()
    model.matmul_precision = "bf16x3"
    img, tgt = O.synthetic_batch(int(g["seed_x"]), n, hw, hw)
    clip = torch.from_numpy(g["clip_features"]).to(DEV)
    model.dropout_mask_override = O.draw_dropout_masks(int(g["seed_drop"]), n)
    logits = model(img.to(DEV), clip)
    loss = ua.SimpleLoss()(logits, tgt.to(DEV))
    loss.backward()
    assert relerr(logits, torch.from_numpy(g["train_logits"])) <= 1e-4
    assert abs(loss.item() - float(g["loss_0"])) <= 2e-4 * abs(float(g["loss_0"]))
    for i, (k, p) in enumerate(model.named_parameters()):
        ref_norm = float(g[f"gnorm_{i}"])
        if ref_norm >= 1e-4:
            assert abs(p.grad.double().norm().item() - ref_norm) <= 5e-3 * ref_norm, k


def test_clip_unet_golden(ua, golden):
    """CLIP_UNet variant (BASELINE config 5) against the fixture recorded from the reference's
    CLIP_UNet/models/unet.py with synthetic CLIP features: eval + train logits, loss, gradients."""
    g = golden("clip64")
    n, hw, clip_dim = int(g["n"]), int(g["hw"]), int(g["clip_dim"])
    sd0 = O.fill_state_dict(int(g["seed_w"]), clip_dim=clip_dim)
    model = ua.CLIPUNet(with_clip_features=True, clip_dim=clip_dim)
    assert [k for k, _ in model.named_parameters()] == [str(s) for s in g["param_names"]]
    model.load_state_dict(sd0)
    model = model.to(DEV)
    img, tgt = O.synthetic_batch(int(g["seed_x"]), n, hw, hw)
    img, tgt = img.to(DEV), tgt.to(DEV)
    clip = torch.from_numpy(g["clip_features"]).to(DEV)
    model.eval()
    with torch.no_grad():
        le = model(img, clip)
        plain = model(img)            # without features the fusion layer is skipped
    assert relerr(le, torch.from_numpy(g["eval_logits"])) <= 1e-4
    assert relerr(plain, le) > 1e-3
    model.train()
    model.dropout_mask_override = O.draw_dropout_masks(int(g["seed_drop"]), n)
    logits = model(img, clip)
    loss = ua.SimpleLoss()(logits, tgt)
    loss.backward()
    assert relerr(logits, torch.from_numpy(g["train_logits"])) <= 1e-4
    assert abs(loss.item() - float(g["loss_0"])) <= 2e-4 * abs(float(g["loss_0"]))
    bad = []
    for i, (k, p) in enumerate(model.named_parameters()):
        ref_norm = float(g[f"gnorm_{i}"])
        got = p.grad.double().norm().item()
        if ref_norm < 1e-4:
            if got >= 1e-3:
                bad.append(f"{k}: should be ~0")
            continue
        if abs(got - ref_norm) > 5e-3 * ref_norm:
            bad.append(f"{k}: grad norm {got} vs {ref_norm}")
        gk = p.grad.reshape(-1)
        idx = torch.from_numpy(sample_idx(gk.numel())).to(DEV)
        ref_s = torch.from_numpy(g[f"gsamp_{i}"])
        err = (gk[idx].cpu() - ref_s).abs()
        tol = 1e-2 * max(ref_s.abs().max().item(), ref_norm / gk.numel() ** 0.5)
        n_out = int((err > tol).sum().item())
        if n_out > 3 + 0.01 * err.numel() or err.max().item() > 5 * tol:
            bad.append(f"{k}: {n_out} of {err.numel()} sampled grads beyond tol, max err "
                       f"{err.max().item():.3e} (tol {tol:.3e})")
    assert not bad, "\n".join(bad)
    # one optimizer step through the flat arena (94 tensors)
    opt = ua.create_optimizer(model)
    loss2 = ua.train_step(model, opt, ua.get_loss_function(), img, tgt)
    assert np.isfinite(loss2.item())


def _grads_after_backward(ua, model, img, tgt, masks):
    model.dropout_mask_override = masks
    for p in model.parameters():
        p.grad = None
    loss = ua.SimpleLoss()(model(img), tgt)
    loss.backward()
    return {k: (None if p.grad is None else p.grad.clone()) for k, p in model.named_parameters()}


@pytest.mark.parametrize("frozen_stages", [6, 3])
def test_frozen_encoder_backward(ua, frozen_stages):
    """AE-transfer style freezing (AE_pretrained/transfer_learning/models/unet.py:448-454): the
    trainable parameters get exactly the gradients of the unfrozen run, frozen ones get none
    and are not touched by the optimizer."""
    sd0 = O.fill_state_dict(11)
    model = ua.UNet()
    model.load_state_dict(sd0)
    model = model.to(DEV).train()
    img, tgt = O.synthetic_batch(21, 2, 64, 64)
    img, tgt = img.to(DEV), tgt.to(DEV)
    masks = O.draw_dropout_masks(3, 2)
    full = _grads_after_backward(ua, model, img, tgt, masks)
    if frozen_stages == 6:
        missing = model.load_pretrained_encoder({"model_state_dict": sd0})
        assert missing == []
    else:
        for s in range(frozen_stages):
            for p in model.encoder_stages[s].parameters():
                p.requires_grad = False
    part = _grads_after_backward(ua, model, img, tgt, masks)
    for k, p in model.named_parameters():
        if not p.requires_grad:
            assert part[k] is None, k
        else:
            assert torch.equal(part[k], full[k]), f"{k}: gradient changed by freezing"
    opt = ua.create_optimizer(model)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    for k, p in model.named_parameters():
        changed = not torch.equal(p.detach(), before[k])
        assert changed == p.requires_grad, k


def test_checkpoint_roundtrip(ua, tmp_path):
    """save_checkpoint/load_checkpoint keep the reference's dictionary layout and resume the
    FusedSGD trajectory exactly (momentum buffers re-adopted into the flat arena)."""
    torch.manual_seed(5)
    img, tgt = O.synthetic_batch(8, 2, 64, 64)
    img, tgt = img.to(DEV), tgt.to(DEV)
    masks = [O.draw_dropout_masks(40 + s, 2) for s in range(3)]

    def run(model, opt, steps):
        out = None
        for s in steps:
            model.dropout_mask_override = masks[s]
            out = ua.train_step(model, opt, ua.get_loss_function(), img, tgt)
        return out

    a = ua.create_model(DEV).train()
    a.load_state_dict(O.fill_state_dict(6))
    opt_a = ua.create_optimizer(a)
    sched_a = ua.create_lr_scheduler(opt_a, 10)
    run(a, opt_a, [0, 1])
    path = ua.save_checkpoint(a, opt_a, sched_a, epoch=4, best_dice=0.5, output_dir=tmp_path,
                              is_best=True)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict",
                       "best_dice", "config"}
    assert len(ck["model_state_dict"]) == 90
    assert len(ck["optimizer_state_dict"]["state"]) == 90
    assert (tmp_path / "best_model.pth").exists()
    # a stock torch optimizer accepts the optimizer state (same layout)
    ref_opt = torch.optim.SGD(ua.UNet().parameters(), lr=0.005, momentum=0.99, nesterov=True,
                              weight_decay=1e-4)
    ref_opt.load_state_dict(ck["optimizer_state_dict"])
    loss_a = run(a, opt_a, [2])

    b = ua.create_model(DEV).train()
    opt_b = ua.create_optimizer(b)
    sched_b = ua.create_lr_scheduler(opt_b, 10)
    start, best = ua.load_checkpoint(path, b, opt_b, sched_b, device=DEV)
    assert (start, best) == (5, 0.5)
    loss_b = run(b, opt_b, [2])
    assert loss_a.item() == loss_b.item()
    for (k, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.equal(p.detach(), q.detach()), k


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_full_size_batch_split_invariance(ua, precision):
    """BASELINE's full size (bs 8, 512x512) selects tile instantiations that the N=2 fixtures
    never reach.  The network is per-sample (InstanceNorm, channel dropout), so a bs-8 pass must
    equal four bs-2 passes over the same images: logits per image, and parameter gradients for
    an injected dL/dlogits as the SUM of the four partial gradients.  This ties the bench-size
    kernels to the configuration the reference fixtures pin (net512 is N=2)."""
    N, hw = 8, 512
    sd0 = O.fill_state_dict(77, trained_like=True)
    img, _ = O.synthetic_batch(4321, N, hw, hw)
    img = img.to(DEV)
    g = torch.Generator(device="cpu").manual_seed(5)
    dlogits = (torch.randn(N, 3, hw, hw, generator=g) * 1e-3).to(DEV)
    masks = O.draw_dropout_masks(91, N)

    def run(sl):
        model = ua.UNet()
        model.load_state_dict(sd0)
        model = model.to(DEV).train()
        model.matmul_precision = precision
        model.dropout_mask_override = [m[sl] for m in masks]
        out = model(img[sl])
        out.backward(dlogits[sl])
        _, garena = model.flat_parameters()
        return out.detach(), garena.detach().clone()

    full_logits, full_grad = run(slice(0, N))
    part_grad = torch.zeros_like(full_grad)
    for i in range(0, N, 2):
        lg, gr = run(slice(i, i + 2))
        e = relerr(full_logits[i:i + 2], lg)
        assert e <= 2e-5, f"images {i}..{i + 1}: bs-8 vs bs-2 logits differ by {e:.2e}"
        part_grad += gr
    e = ((full_grad - part_grad).norm() / part_grad.norm()).item()
    # The two tilings sum in different orders, so activations differ by ~1e-6 and a fraction
    # f ~ 5e-6 of the LeakyReLU inputs changes sign; each flip changes that element's gradient by
    # O(1), i.e. a norm-wise error ~ sqrt(f) = 2e-3 at the last layer, growing to 6e-3 at the
    # first (measured; tests/tools/diag_batch_split.py) - the same effect that separates the
    # reference's own fp32 run from fp64 at this size (profiles/r01_grad_accuracy_vs_fp64_512.txt).
    # A wrong tile instantiation gives O(1) errors.
    assert e <= 2e-2, f"bs-8 gradient vs sum of bs-2 gradients: {e:.2e}"
    # the head's weight gradient has no LeakyReLU downstream of it: tight
    n_head = 96 + 4          # arena tail: head weight [3,32,1,1] + bias [3] padded to 4 floats
    eh = ((full_grad[-n_head:] - part_grad[-n_head:]).norm() / part_grad[-n_head:].norm()).item()
    assert eh <= 1e-4, f"head gradient: {eh:.2e}"


def _per_tensor_errors(model, full, part, names=None):
    """max |full - part| / max |part| per parameter tensor of two gradient arenas"""
    out = {}
    for (k, p), off in zip(model.named_parameters(), model._offsets):
        a, b = full[off:off + p.numel()], part[off:off + p.numel()]
        out[k] = ((a - b).abs().max().item(), b.abs().max().item())
    return out


def test_full_size_batch_split_tie_free_gradients_per_element(ua):
    """The bench configuration (bs 8, 512 x 512) is the only place where the 256- and 512-channel
    Winograd launches (conv_wino_kernel forward / data gradient / up-sampling loader,
    conv_wgrad_wino_kernel) are selected: at N = 2 those layers have too few tiles and run the
    direct kernels, which the reference fixtures and the per-element oracle tests pin.  With
    negative_slope = 1 there is no LeakyReLU branch that a different summation order could flip,
    so the bs-8 gradient arena must equal the SUM of the four bs-2 arenas on EVERY element of
    every tensor to 1e-4 of that tensor's max magnitude (and the logits to 2e-5): this holds each
    N=8-only instantiation against the direct kernels of the N=2 runs, element by element."""
    N, hw = 8, 512
    sel8, sel2 = _winograd_selection(ua, N, hw), _winograd_selection(ua, 2, hw)
    assert all(sel8.values()), sel8
    only8 = [k for k in sel8 if not sel2[k]]
    assert any(k.startswith("256 ") for k in only8) and any(k.startswith("512 ") for k in only8), \
        f"expected the 256/512-channel Winograd forms to be N=8-only: {only8}"
    sd0 = O.fill_state_dict(78, trained_like=True)
    img, _ = O.synthetic_batch(4322, N, hw, hw)
    img = img.to(DEV)
    g = torch.Generator(device="cpu").manual_seed(6)
    dlogits = (torch.randn(N, 3, hw, hw, generator=g) * 1e-3).to(DEV)
    masks = O.draw_dropout_masks(92, N)

    def run(sl):
        model = ua.UNet(nonlin_kwargs={"negative_slope": 1.0, "inplace": True})
        model.load_state_dict(sd0)
        model = model.to(DEV).train()
        model.dropout_mask_override = [m[sl] for m in masks]
        out = model(img[sl])
        out.backward(dlogits[sl])
        _, garena = model.flat_parameters()
        return model, out.detach(), garena.detach().clone()

    model, full_logits, full_grad = run(slice(0, N))
    part_grad = torch.zeros_like(full_grad)
    for i in range(0, N, 2):
        _, lg, gr = run(slice(i, i + 2))
        e = relerr(full_logits[i:i + 2], lg)
        assert e <= 2e-5, f"images {i}..{i + 1}: bs-8 vs bs-2 logits differ by {e:.2e}"
        part_grad += gr
    bad = []
    for k, (err, scale) in _per_tensor_errors(model, full_grad, part_grad).items():
        if scale < 1e-6:          # conv biases under InstanceNorm: exact 0 up to rounding
            continue
        if err > 1e-4 * scale:
            bad.append(f"{k}: {err / scale:.2e}")
    assert not bad, "\n".join(bad)


def test_clip_unet_full_size_batch_split(ua):
    """BASELINE config 5 at its stated size: CLIPUNet, bs 8, 512 x 512, CLIP features
    [8, 512, 16, 16] (the fixture clip64.npz covers a 2 x 2 bottleneck grid only).  At this size
    the fusion layer is unet_conv_in_fwd(ksize = 1) at M = 2048 rows, K = 1024, and the rest of
    the network runs its bs-8 selections.  Per-sample network => the bs-8 pass must equal four
    bs-2 passes: logits to 2e-5 and, on the tie-free network (negative_slope = 1), every element
    of all 94 gradients to 1e-4 of the tensor's max (CLIP_UNet/models/unet.py:441-478)."""
    N, hw, clip_dim = 8, 512, 512
    sd0 = O.fill_state_dict(79, trained_like=True, clip_dim=clip_dim)
    img, _ = O.synthetic_batch(4323, N, hw, hw)
    img = img.to(DEV)
    g = torch.Generator(device="cpu").manual_seed(7)
    clip = torch.randn(N, clip_dim, hw // 32, hw // 32, generator=g).to(DEV)
    dlogits = (torch.randn(N, 3, hw, hw, generator=g) * 1e-3).to(DEV)
    masks = O.draw_dropout_masks(93, N)

    def run(sl):
        model = ua.CLIPUNet(with_clip_features=True, clip_dim=clip_dim,
                            nonlin_kwargs={"negative_slope": 1.0, "inplace": True})
        model.load_state_dict(sd0)
        model = model.to(DEV).train()
        model.dropout_mask_override = [m[sl] for m in masks]
        out = model(img[sl], clip[sl])
        out.backward(dlogits[sl])
        _, garena = model.flat_parameters()
        return model, out.detach(), garena.detach().clone()

    model, full_logits, full_grad = run(slice(0, N))
    assert len(list(model.parameters())) == 94
    part_grad = torch.zeros_like(full_grad)
    for i in range(0, N, 2):
        _, lg, gr = run(slice(i, i + 2))
        e = relerr(full_logits[i:i + 2], lg)
        assert e <= 2e-5, f"images {i}..{i + 1}: bs-8 vs bs-2 logits differ by {e:.2e}"
        part_grad += gr
    bad = []
    errs = _per_tensor_errors(model, full_grad, part_grad)
    assert errs["clip_fusion_conv.0.weight"][1] > 0      # the fusion layer did get a gradient
    for k, (err, scale) in errs.items():
        if scale < 1e-6:
            continue
        if err > 1e-4 * scale:
            bad.append(f"{k}: {err / scale:.2e}")
    assert not bad, "\n".join(bad)
    # and the default-slope network at this size against the same split (logits only: the
    # gradients of a LeakyReLU network can flip ties between two tilings)
    def run01(sl):
        model = ua.CLIPUNet(with_clip_features=True, clip_dim=clip_dim)
        model.load_state_dict(sd0)
        model = model.to(DEV).eval()
        with torch.no_grad():
            return model(img[sl], clip[sl])
    full = run01(slice(0, N))
    for i in range(0, N, 4):
        assert relerr(full[i:i + 4], run01(slice(i, i + 4))) <= 2e-5


def test_drawn_dropout_masks_have_the_reference_distribution(ua):
    """Train mode without an override: one bernoulli draw covers all 16 SpatialDropout2d modules;
    every mask is [N, C] with values {0, 1/(1-p)} and keep rate 1-p (models/unet.py:22-35)."""
    from unet_implementations_amd import unet as U
    model = ua.create_model().train()
    enc, dec = model._build_plan() if model._plan is None else model._plan
    layers = [l for blk in enc for l in blk] + [l for blk in dec for l in blk]
    torch.manual_seed(0)
    n = 64
    masks = U._draw_masks(model, layers, n, torch.device(DEV))
    drops = [(l, m) for l, m in zip(layers, masks) if l.drop is not None and l.drop.drop_prob > 0]
    assert len(drops) == 16 and all(m is None for l, m in zip(layers, masks)
                                    if l.drop is None or l.drop.drop_prob == 0)
    for l, m in drops:
        p = l.drop.drop_prob
        assert m.shape == (n, l.conv.out_channels) and m.is_contiguous()
        vals = m.unique().tolist()
        assert all(abs(v) < 1e-6 or abs(v - 1 / (1 - p)) < 1e-5 for v in vals)
        keep = (m > 0).float().mean().item()
        assert abs(keep - (1 - p)) < 0.03, (l.name, keep)
    again = U._draw_masks(model, layers, n, torch.device(DEV))
    assert not torch.equal(again[layers.index(drops[0][0])], drops[0][1])      # fresh draw
    model.eval()
    assert all(m is None for m in U._draw_masks(model, layers, n, torch.device(DEV)))


def test_gradient_accumulation_and_in_place_zero_grad(ua):
    """`.grad` are views of the gradient arena the kernels write.  A second backward before the
    step must ACCUMULATE (reference semantics: loss.backward() twice adds), and
    optimizer.zero_grad(set_to_none=False) followed by backward must give g, not 2 g."""
    sd0 = O.fill_state_dict(3)
    model = ua.UNet()
    model.load_state_dict(sd0)
    model = model.to(DEV).train()
    lossf = ua.get_loss_function()
    (img1, tgt1), (img2, tgt2) = O.synthetic_batch(1, 2, 64, 64), O.synthetic_batch(2, 2, 64, 64)
    masks = O.draw_dropout_masks(4, 2)
    model.dropout_mask_override = masks

    def grads_of(img, tgt):
        for p in model.parameters():
            p.grad = None
        lossf(model(img.to(DEV)), tgt.to(DEV)).backward()
        return [p.grad.detach().clone() for p in model.parameters()]

    g1, g2 = grads_of(img1, tgt1), grads_of(img2, tgt2)
    for p in model.parameters():
        p.grad = None
    lossf(model(img1.to(DEV)), tgt1.to(DEV)).backward()
    lossf(model(img2.to(DEV)), tgt2.to(DEV)).backward()
    _, garena = model.flat_parameters()
    for p, a, b, off in zip(model.parameters(), g1, g2, model._offsets):
        assert p.grad.data_ptr() == garena.data_ptr() + 4 * off      # still the arena view
        assert torch.allclose(p.grad, a + b, rtol=1e-6, atol=1e-7 * (a + b).abs().max().item())
    opt = ua.create_optimizer(model)
    opt.zero_grad(set_to_none=False)
    assert all(float(p.grad.abs().max()) == 0.0 for p in model.parameters())
    lossf(model(img1.to(DEV)), tgt1.to(DEV)).backward()
    for p, a in zip(model.parameters(), g1):
        assert torch.equal(p.grad, a)
    assert opt._flat_ready()


def test_layer_that_did_not_run_gets_no_gradient(ua):
    """CLIPUNet.forward(x) without features skips the fusion layer: its four parameters keep
    .grad = None (as in the reference), so SGD leaves them - momentum and weight decay included -
    alone."""
    model = ua.CLIPUNet(with_clip_features=True, clip_dim=512).to(DEV).train()
    img, tgt = O.synthetic_batch(9, 2, 64, 64)
    opt = ua.create_optimizer(model)
    before = {k: p.detach().clone() for k, p in model.named_parameters() if "clip_fusion" in k}
    assert len(before) == 4
    loss = ua.train_step(model, opt, ua.get_loss_function(), img.to(DEV), tgt.to(DEV))
    assert torch.isfinite(loss)
    for k, p in model.named_parameters():
        if "clip_fusion" in k:
            assert p.grad is None and torch.equal(p.detach(), before[k])
        else:
            assert p.grad is not None


def test_uint8_batch_through_the_fused_stem(ua):
    """forward(x_u8, input_layout="nhwc_u8"): the dataset normalisation (Our_UNet/src/train.py:
    303-308) runs inside the loaders of the first convolution and of its weight gradient.  The
    loaders compute the very floats `ops.preprocess_u8` writes (bit-exact against
    oracle.preprocess_sample, test_preprocess_u8_bit_exact), so logits and every gradient must be
    IDENTICAL to the run on the preprocessed fp32 tensor."""
    g = torch.Generator().manual_seed(5)
    x_u8 = torch.randint(0, 256, (2, 128, 128, 3), generator=g, dtype=torch.uint8).to(DEV)
    tgt = torch.randint(0, 3, (2, 128, 128), generator=g).to(DEV)
    sd0 = O.fill_state_dict(3)
    model = ua.UNet()
    model.load_state_dict(sd0)
    model = model.to(DEV).train()
    model.dropout_mask_override = O.draw_dropout_masks(4, 2)
    lossf = ua.get_loss_function()

    def run(layout):
        for p in model.parameters():
            p.grad = None
        inp = x_u8 if layout == "nhwc_u8" else ua.ops.preprocess_u8(x_u8)[0]
        logits = model(inp, input_layout=layout)
        lossf(logits, tgt).backward()
        return logits.detach().clone(), [p.grad.detach().clone() for p in model.parameters()]

    l_ref, g_ref = run("nhwc")
    l_u8, g_u8 = run("nhwc_u8")
    assert torch.equal(l_u8, l_ref)
    for a, b in zip(g_u8, g_ref):
        assert torch.equal(a, b)
    # widths the raw-row stem does not cover fall back to the preprocessing kernel
    x96 = torch.randint(0, 256, (1, 96, 96, 3), generator=g, dtype=torch.uint8).to(DEV)
    model.eval()
    with torch.no_grad():
        a = model(x96, input_layout="nhwc_u8")
        b = model(ua.ops.preprocess_u8(x96)[0], input_layout="nhwc")
    assert torch.equal(a, b)


def test_loss_resizes_logits_to_the_target_like_the_reference(ua):
    """SimpleLoss.forward with logits at half the target's resolution
    (Our_UNet/models/losses.py:66-68): loss and the gradient w.r.t. the low-resolution logits
    against the oracle's loss on F.interpolate'd logits."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(8)
    logits = torch.randn(2, 3, 32, 48, generator=g)
    _, tgt = O.synthetic_batch(4, 2, 64, 96)
    lr = logits.clone().requires_grad_(True)
    ref = O.simple_loss(F.interpolate(lr, size=(64, 96), mode="bilinear", align_corners=False), tgt)
    ref.backward()
    ld = logits.to(DEV).requires_grad_(True)
    loss = ua.get_loss_function()(ld, tgt.to(DEV))
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 2e-5 * abs(ref.item())
    assert (ld.grad.cpu() - lr.grad).abs().max() <= 5e-5 * lr.grad.abs().max()


def test_clip_features_of_another_size_are_resized_to_the_bottleneck_grid(ua, golden):
    """CLIP_UNet/models/unet.py:444-450: features whose spatial size differs from the bottleneck's
    are resized bilinearly first.  Features given at twice the grid must produce the logits of
    their F.interpolate'd version passed directly."""
    import torch.nn.functional as F
    g = golden("clip64")
    n, hw, clip_dim = int(g["n"]), int(g["hw"]), int(g["clip_dim"])
    model = ua.CLIPUNet(with_clip_features=True, clip_dim=clip_dim)
    model.load_state_dict(O.fill_state_dict(int(g["seed_w"]), clip_dim=clip_dim))
    model = model.to(DEV).eval()
    img, _ = O.synthetic_batch(int(g["seed_x"]), n, hw, hw)
    big = torch.randn(n, clip_dim, 2 * (hw // 32), 2 * (hw // 32),
                      generator=torch.Generator().manual_seed(1))
    small = F.interpolate(big, size=(hw // 32, hw // 32), mode="bilinear", align_corners=False)
    with torch.no_grad():
        a = model(img.to(DEV), big.to(DEV))
        b = model(img.to(DEV), small.to(DEV))
    assert relerr(a, b) <= 1e-5


def test_grad_cam_style_hooks_on_stage_modules(ua):
    """The reference's Grad-CAM helper (Our_UNet/utils/visualize.py:392-412) registers a forward
    hook and a (legacy) backward hook on a target layer.  The fused walk never calls sub-modules,
    so stage-level modules get their hooks fired by the walk itself, with a materialised NCHW
    output / grad_output.  Checked on the last decoder stage - the head is linear in it, so
    logits = conv1x1(fmap) and d(logit[0, c].mean()) / d fmap = w_head[c] / (H W) for image 0 -
    and on the first encoder stage against the oracle's two units."""
    import torch.nn.functional as F
    sd0 = O.fill_state_dict(31)
    model = ua.UNet()
    model.load_state_dict(sd0)
    model = model.to(DEV).eval()
    img, _ = O.synthetic_batch(2, 2, 64, 64)
    got = {}
    target = model.decoder_stages[-1]
    h1 = target.register_forward_hook(lambda m, i, o: got.__setitem__("fmap", o.detach()))
    h2 = target.register_backward_hook(lambda m, gi, go: got.__setitem__("grad", go[0].detach()))
    h3 = model.encoder_stages[0].register_forward_hook(
        lambda m, i, o: got.__setitem__("enc0", o.detach()))
    h4 = model.encoder_stages[0].register_full_backward_hook(
        lambda m, gi, go: got.__setitem__("genc0", go[0].detach()))
    out = model(img.to(DEV))
    cls = 1
    model.zero_grad()
    out[0, cls].mean().backward(retain_graph=True)
    for h in (h1, h2, h3, h4):
        h.remove()
    fmap, grad = got["fmap"], got["grad"]
    assert fmap.shape == (2, 32, 64, 64) and grad.shape == fmap.shape
    head = model.segmentation_output
    logits2 = F.conv2d(fmap, head.weight.detach(), head.bias.detach())
    assert relerr(logits2, out.detach()) <= 1e-5
    want = torch.zeros_like(grad)
    want[0] = (head.weight.detach()[cls, :, 0, 0] / (64 * 64))[:, None, None]
    assert (grad - want).abs().max() <= 1e-6 * want.abs().max()
    # first encoder stage vs the oracle's two conv -> norm -> lrelu units (eval: no dropout)
    osd = O.leaf_state_dict(sd0)
    rows = O.layer_table()[:2]
    cur = img
    for prefix, ci, ni, _, _, stride, _, _ in rows:
        cur = O.conv_in_lrelu_drop(cur, osd[f"{prefix}.{ci}.weight"].detach(),
                                   osd[f"{prefix}.{ci}.bias"].detach(),
                                   osd[f"{prefix}.{ni}.weight"].detach(),
                                   osd[f"{prefix}.{ni}.bias"].detach(), stride)
    assert relerr(got["enc0"], cur) <= 1e-4
    assert got["genc0"].shape == got["enc0"].shape and torch.isfinite(got["genc0"]).all()
    # without hooks nothing is materialised and the output is unchanged
    assert torch.equal(model(img.to(DEV)).detach(), out.detach())
