"""CPU (-m "not gpu"): pins the ORACLE (oracle/unet_ref.py) against the fixtures that
tests/tools/make_golden.py recorded from the reference's own Our_UNet/models/{unet,losses}.py.

The fixtures were produced on the build container's CPU; another host may pick different
oneDNN kernels, so comparisons allow fp32 summation-order noise (1e-5 relative on forward
values, norm-wise 1e-2 on whole-network gradients: see tests/test_net_gpu.py for why)."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import unet_ref as O


def relerr(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def test_layer_table_matches_fixture_names(golden):
    g = golden("net64")
    names = [str(s) for s in g["param_names"]]
    sd = O.fill_state_dict(int(g["seed_w"]))
    assert list(sd.keys()) == names
    assert len(names) == 90
    assert sum(v.numel() for v in sd.values()) == 19_655_235


def test_convblock_restatement(golden):
    g = golden("ops_small")
    p = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("cb_p_")}
    x = torch.from_numpy(g["cb_x"]).requires_grad_(True)
    m0, m1 = torch.from_numpy(g["cb_mask0"]), torch.from_numpy(g["cb_mask1"])
    t = O.conv_in_lrelu_drop(x, p["block.0.weight"], p["block.0.bias"], p["block.1.weight"],
                             p["block.1.bias"], 2, m0)
    t = O.conv_in_lrelu_drop(t, p["block.4.weight"], p["block.4.bias"], p["block.5.weight"],
                             p["block.5.bias"], 1, m1)
    assert relerr(t.detach(), torch.from_numpy(g["cb_y"])) <= 1e-5
    t.backward(torch.from_numpy(g["cb_gy"]))
    assert relerr(x.grad, torch.from_numpy(g["cb_gx"])) <= 1e-4


def test_upblock_restatement(golden):
    g = golden("ops_small")
    p = {k[16:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("ub_p_conv_block.")}
    x = torch.from_numpy(g["ub_x"])
    skip = torch.from_numpy(g["ub_skip"])
    t = O.upsample_concat(x, skip)
    t = O.conv_in_lrelu_drop(t, p["block.0.weight"], p["block.0.bias"], p["block.1.weight"],
                             p["block.1.bias"], 1)
    t = O.conv_in_lrelu_drop(t, p["block.3.weight"], p["block.3.bias"], p["block.4.weight"],
                             p["block.4.bias"], 1)
    assert relerr(t, torch.from_numpy(g["ub_y"])) <= 1e-5


def test_loss_restatement(golden):
    g = golden("ops_small")
    lg = torch.from_numpy(g["loss_logits"]).requires_grad_(True)
    tg = torch.from_numpy(g["loss_target"])
    loss = O.simple_loss(lg, tg)
    assert abs(loss.item() - float(g["loss_value"])) <= 1e-6 * abs(float(g["loss_value"]))
    loss.backward()
    assert relerr(lg.grad, torch.from_numpy(g["loss_dlogits"])) <= 1e-5
    lg2 = torch.from_numpy(g["loss_logits"]).requires_grad_(True)
    loss2 = O.simple_loss(lg2, tg, dynamic_weights=False,
                          fixed_weights=torch.from_numpy(g["loss2_weights"]))
    assert abs(loss2.item() - float(g["loss2_value"])) <= 1e-6 * abs(float(g["loss2_value"]))


def test_class_weights_missing_class():
    tg = torch.zeros(1, 4, 4, dtype=torch.int64)
    tg[0, 0, :] = 1
    tg[0, 3, :] = 255
    w = O.class_weights(tg)
    # counts: class0 = 8, class1 = 4, class2 = 0 -> 1 ; total valid = 12
    raw = torch.tensor([12 / 8, 12 / 4, 12 / 1.0])
    assert torch.allclose(w, raw * 3 / raw.sum())


def test_sgd_restatement(golden):
    g = golden("ops_small")
    p = [torch.from_numpy(g["sgd_p0"]).clone()]
    bufs = [None]
    for s in range(3):
        O.sgd_nesterov_(p, [torch.from_numpy(g["sgd_grads"][s])], bufs)
        assert relerr(p[0], torch.from_numpy(g["sgd_traj"][s])) <= 1e-6


def test_net64_forward_and_first_step(golden):
    g = golden("net64")
    n, hw = int(g["n"]), int(g["hw"])
    sd0 = O.fill_state_dict(int(g["seed_w"]))
    img, tgt = O.synthetic_batch(int(g["seed_x"]), n, hw, hw)
    with torch.no_grad():
        le = O.unet_forward(sd0, img)
    assert relerr(le, torch.from_numpy(g["eval_logits"])) <= 1e-5
    am = le.argmax(dim=1).to(torch.uint8).numpy()
    low = np.unpackbits(g["eval_lowmargin"])[: am.size].reshape(am.shape).astype(bool)
    bits = np.unpackbits(g["eval_argmax"])[: 2 * am.size].reshape(-1, 2)
    ref_am = (bits[:, 0] * 2 + bits[:, 1]).reshape(am.shape).astype(np.uint8)
    assert hashlib.sha256(ref_am.tobytes()).hexdigest() == str(g["eval_argmax_sha256"])
    assert np.array_equal(am[~low], ref_am[~low])

    masks = O.draw_dropout_masks(int(g["seed_drop"]), n)
    osd = O.leaf_state_dict(sd0)
    loss, logits, grads = O.train_step(osd, [None] * len(osd), img, tgt, masks)
    assert relerr(logits, torch.from_numpy(g["train_logits"])) <= 1e-5
    assert abs(loss.item() - float(g["loss_0"])) <= 1e-5 * abs(float(g["loss_0"]))
    for i, k in enumerate(osd):
        ref = float(g[f"gnorm_{i}"])
        if ref < 1e-4:
            continue
        assert abs(grads[k].double().norm().item() - ref) <= 1e-2 * ref, k


def test_dropout_replay_is_deterministic():
    a = O.draw_dropout_masks(5, 3)
    b = O.draw_dropout_masks(5, 3)
    assert len(a) == 16 and all(torch.equal(x, y) for x, y in zip(a, b))
    assert [tuple(m.shape) for m in a[:2]] == [(3, 128), (3, 128)]
    for m, (c, p) in zip(a, O.dropout_layers()):
        vals = set(np.round(m.unique().tolist(), 5))
        assert vals <= {0.0, round(1 / (1 - p), 5)}


@pytest.mark.parametrize("hw", [64])
def test_synthetic_batch_structure(hw):
    img, tgt = O.synthetic_batch(1, 4, hw, hw)
    assert img.shape == (4, 3, hw, hw) and tgt.shape == (4, hw, hw) and tgt.dtype == torch.int64
    for i in range(4):
        vals = set(tgt[i].unique().tolist())
        assert vals <= {0, 1 + (i % 2), 255} and 255 in vals and 0 in vals


def test_clip_variant_restatement(golden):
    g = golden("clip64")
    n, hw, clip_dim = int(g["n"]), int(g["hw"]), int(g["clip_dim"])
    sd0 = O.fill_state_dict(int(g["seed_w"]), clip_dim=clip_dim)
    assert list(sd0.keys()) == [str(s) for s in g["param_names"]] and len(sd0) == 94
    base = O.fill_state_dict(int(g["seed_w"]))
    assert all(torch.equal(sd0[k], v) for k, v in base.items())   # base tensors unchanged
    img, tgt = O.synthetic_batch(int(g["seed_x"]), n, hw, hw)
    clip = torch.from_numpy(g["clip_features"])
    with torch.no_grad():
        le = O.unet_forward(sd0, img, clip_features=clip)
    assert relerr(le, torch.from_numpy(g["eval_logits"])) <= 1e-5
    masks = O.draw_dropout_masks(int(g["seed_drop"]), n)
    osd = O.leaf_state_dict(sd0)
    loss, logits, grads = O.train_step(osd, [None] * len(osd), img, tgt, masks, clip_features=clip)
    assert relerr(logits, torch.from_numpy(g["train_logits"])) <= 1e-5
    assert abs(loss.item() - float(g["loss_0"])) <= 1e-5 * abs(float(g["loss_0"]))


METRIC_FIELDS = ("intersections", "unions", "true_positives", "false_positives", "false_negatives")


def test_segmentation_metrics_restatement(golden):
    """oracle.segmentation_metrics against the accumulators and scores recorded from the
    reference's SegmentationMetrics (Our_UNet/utils/metrics.py:59-151): integer-exact counts,
    identical quotients (nan where the reference returns nan), per batch and accumulated; the
    stored predictions are torch.argmax of the stored logits (first maximum wins on ties)."""
    g = golden("metrics")
    preds, targets = [], []
    for k in range(int(g["n_batches"])):
        lg, t = torch.from_numpy(g[f"b{k}_logits"]), g[f"b{k}_target"]
        p = lg.argmax(dim=1).numpy()
        assert np.array_equal(p, g[f"b{k}_pred"])
        r = O.segmentation_metrics([p], [t])
        for f in METRIC_FIELDS:
            assert np.array_equal(r[f], g[f"b{k}_{f}"]), (k, f)
        assert r["total_pixels"] == int(g[f"b{k}_total_pixels"])
        assert r["correct_pixels"] == int(g[f"b{k}_correct_pixels"])
        for f in ("iou", "dice"):
            assert np.array_equal(r[f], g[f"b{k}_{f}"], equal_nan=True), (k, f)
        preds.append(p)
        targets.append(t)
    r = O.segmentation_metrics(preds, targets)
    for f in METRIC_FIELDS:
        assert np.array_equal(r[f], g[f"acc_{f}"]), f
    for f in ("iou", "dice"):
        assert np.array_equal(r[f], g[f"acc_{f}"], equal_nan=True)
    for f in ("pixel_accuracy", "mean_iou", "mean_dice"):
        assert r[f] == float(g[f"acc_{f}"])
    # the fixture does contain the edge cases it claims
    assert np.isnan(g["b0_iou"][2]) or g["b0_unions"][2] > 0
    assert g["b0_true_positives"][2] == 0 and g["b1_true_positives"][1] == 0
