"""CPU (-m "not gpu"): host-side logic of the drop-in surface: module tree, state_dict keys,
plan construction, loud failure off-GPU, optimizer state layout, flat-arena bookkeeping."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import unet_ref as O


def test_state_dict_keys_match_reference_fixture(ua, golden):
    g = golden("net64")
    names = [str(s) for s in g["param_names"]]
    model = ua.UNet()
    sd = model.state_dict()
    assert list(sd.keys()) == names and len(sd) == 90
    ref = O.fill_state_dict(1)
    assert all(sd[k].shape == ref[k].shape for k in names)
    assert sum(p.numel() for p in model.parameters()) == 19_655_235
    assert not list(model.buffers())           # InstanceNorm2d: no running stats
    model.load_state_dict(ref)                 # strict load of a reference-shaped checkpoint


def test_module_tree_and_attributes(ua):
    m = ua.create_model("cpu")
    assert (m.in_channels, m.num_classes, m.n_stages) == (3, 3, 6)
    assert m.features_per_stage == [32, 64, 128, 256, 512, 512]
    assert isinstance(m.encoder_stages, nn.ModuleList) and len(m.encoder_stages) == 6
    assert isinstance(m.decoder_stages, nn.ModuleList) and len(m.decoder_stages) == 5
    # what Grad-CAM / load_pretrained_encoder address in the reference
    assert isinstance(m.decoder_stages[0].conv_block.block[0], nn.Conv2d)
    assert m.decoder_stages[0].conv_block.block[0].in_channels == 1024
    enc_sd = m.encoder_stages.state_dict()
    other = ua.UNet()
    other.encoder_stages.load_state_dict(enc_sd)
    assert isinstance(m.segmentation_output, nn.Conv2d)
    assert m.segmentation_output.weight.shape == (3, 32, 1, 1)


def test_initialisation_follows_reference_rule(ua):
    torch.manual_seed(0)
    m = ua.UNet()
    for mod in m.modules():
        if isinstance(mod, nn.Conv2d):
            assert torch.all(mod.bias == 0)
            fan_out = mod.out_channels * mod.kernel_size[0] * mod.kernel_size[1]
            std = mod.weight.std().item()
            if mod.weight.numel() > 10_000:
                assert abs(std - (2.0 / fan_out) ** 0.5) < 0.05 * (2.0 / fan_out) ** 0.5
        elif isinstance(mod, nn.InstanceNorm2d):
            assert torch.all(mod.weight == 1) and torch.all(mod.bias == 0)


def test_plan_matches_oracle_layer_table(ua):
    m = ua.UNet()
    assert m.check_supported()
    enc, dec = m._plan
    layers = [l for b in enc for l in b] + [l for b in dec for l in b]
    rows = O.layer_table()
    assert len(layers) == len(rows) == 22
    for l, (prefix, ci, ni, cin, cout, stride, p, kind) in zip(layers, rows):
        assert (l.conv.in_channels, l.conv.out_channels, l.stride) == (cin, cout, stride)
        assert (l.drop.drop_prob if l.drop is not None else 0.0) == p
        assert l.first_of_decoder == (kind == "dec_first")
        assert l.slope == pytest.approx(0.01) and l.norm.eps == 1e-5


def test_unsupported_configurations_fail_loudly(ua):
    with pytest.raises(NotImplementedError):
        ua.UNet(norm_op=nn.BatchNorm2d, norm_op_kwargs={}).check_supported()
    with pytest.raises(NotImplementedError):
        ua.UNet(nonlin=nn.ReLU).check_supported()
    with pytest.raises(NotImplementedError):
        ua.UNet(kernel_sizes=[[5, 5]] * 6).check_supported()
    with pytest.raises(NotImplementedError):
        ua.UNet(dropout_op=nn.Dropout2d, dropout_op_kwargs={"p": 0.1}).check_supported()


def test_cpu_tensors_are_rejected(ua):
    m = ua.UNet()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 3, 64, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ua.SimpleLoss()(torch.randn(1, 3, 8, 8), torch.zeros(1, 8, 8, dtype=torch.long))
    # blocks are callable on their own (as in the reference), on the device only
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.encoder_stages[0](torch.randn(1, 3, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.decoder_stages[0](torch.randn(1, 512, 2, 2), torch.randn(1, 512, 4, 4))


def test_hooks_on_inner_modules_are_refused(ua):
    """The fused walk fires stage-level hooks only; a hook on an inner module (the reference's
    Grad-CAM helper accepts any layer, Our_UNet/utils/visualize.py:401-402) must raise instead of
    silently never firing.  The check runs before anything touches the device."""
    m = ua.UNet()
    h = m.decoder_stages[0].conv_block.block[0].register_forward_hook(lambda *a: None)
    with pytest.raises(NotImplementedError, match="decoder_stages.0.conv_block.block.0"):
        m._plan or m._build_plan()
        m._check_hooks()
    h.remove()
    m._check_hooks()
    h = m.encoder_stages[2].block.register_full_backward_hook(lambda *a: None)
    with pytest.raises(NotImplementedError, match="encoder_stages.2.block"):
        m._check_hooks()
    h.remove()
    for mod in (m.encoder_stages[1], m.decoder_stages[2], m.decoder_stages[2].conv_block,
                m.segmentation_output, m):
        mod.register_forward_hook(lambda *a: None)
    m._check_hooks()        # stage-level hooks are supported


def test_spatial_dropout_mask_draw_matches_reference_recipe(ua):
    d = ua.SpatialDropout2d(0.3)
    torch.manual_seed(11)
    mask = d.draw_mask(4, 64, "cpu")
    torch.manual_seed(11)
    ref = torch.empty(4, 64, 1, 1).bernoulli_(0.7).div_(0.7).view(4, 64)
    assert torch.equal(mask, ref)
    d.eval()
    x = torch.randn(2, 64, 4, 4)
    assert d(x) is x


def test_flat_arena_aliases_parameters(ua):
    m = ua.UNet()
    arena, garena = m.flat_parameters()
    assert arena.numel() == garena.numel() >= 19_655_235 and arena.numel() % 4 == 0
    for p, off in zip(m.parameters(), m._offsets):
        assert off % 4 == 0 and p.data_ptr() == arena.data_ptr() + 4 * off
    # load_state_dict copies into the views: aliasing survives
    m.load_state_dict(O.fill_state_dict(3))
    w = m.encoder_stages[0].block[0].weight
    assert torch.equal(arena[:w.numel()].view_as(w), w.data)
    a2, _ = m.flat_parameters()
    assert a2.data_ptr() == arena.data_ptr()


def test_fused_sgd_state_dict_layout(ua):
    m = ua.UNet()
    opt = ua.create_optimizer(m)
    sd = opt.state_dict()
    g = sd["param_groups"][0]
    assert (g["lr"], g["momentum"], g["weight_decay"], g["nesterov"]) == (0.005, 0.99, 1e-4, True)
    assert len(g["params"]) == 90
    ref = torch.optim.SGD(ua.UNet().parameters(), lr=0.005, momentum=0.99, nesterov=True,
                          weight_decay=1e-4).state_dict()["param_groups"][0]
    assert set(ref.keys()) <= set(g.keys()) | {"maximize", "foreach", "differentiable", "fused"}
    with pytest.raises(NotImplementedError):
        ua.FusedSGD(m.parameters(), lr=0.1, momentum=0.9, nesterov=False)


def test_poly_lr_schedule(ua):
    m = ua.UNet()
    opt = ua.create_optimizer(m)
    sched = ua.create_lr_scheduler(opt, 10)
    lrs = []
    for _ in range(3):
        lrs.append(opt.param_groups[0]["lr"])
        sched.step()
    assert np.allclose(lrs, [0.005 * (1 - e / 10) ** 0.9 for e in range(3)])


def test_clip_unet_surface(ua, golden):
    g = golden("clip64")
    m = ua.CLIPUNet()
    assert [k for k, _ in m.named_parameters()] == [str(s) for s in g["param_names"]]
    assert len(m.state_dict()) == 94 and sum(p.numel() for p in m.parameters()) == 20_181_059
    assert (m.with_clip_features, m.clip_dim) == (True, 512)
    assert m.check_supported() and m._fusion_layer.ksize == 1
    assert isinstance(m.clip_fusion_conv[0], nn.Conv2d) and m.clip_fusion_conv[0].in_channels == 1024
    plain = ua.CLIPUNet(with_clip_features=False)
    assert len(plain.state_dict()) == 90 and not hasattr(plain, "clip_fusion_conv")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.randn(1, 3, 64, 64), torch.randn(1, 512, 2, 2))
