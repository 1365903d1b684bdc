"""CPU (-m "not gpu"): libunet_hip.so loads, exports every symbol include/unet_hip.h declares,
and rejects bad arguments before any GPU work (no compute calls are made here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "unet_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(unet_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    names = declared_functions()
    for must in ["unet_conv3x3_fwd", "unet_conv3x3_bwd_data", "unet_conv3x3_bwd_weight",
                 "unet_instnorm_stats", "unet_instnorm_lrelu_drop_fwd",
                 "unet_instnorm_lrelu_drop_bwd", "unet_upsample2x_fwd", "unet_upsample2x_bwd",
                 "unet_head1x1_fwd", "unet_head1x1_bwd", "unet_dice_wce_loss_fwd_bwd",
                 "unet_sgd_nesterov_step", "unet_pack_conv3x3_weights", "unet_last_error"]:
        assert must in names


def test_library_exports_every_declared_symbol(ua):
    handle = ctypes.CDLL(ua.LIB_PATH)
    for name in declared_functions():
        assert hasattr(handle, name), f"{name} declared in unet_hip.h but not exported"


def test_python_binding_covers_every_declared_symbol(ua):
    assert sorted(ua._lib.SIGNATURES.keys()) == declared_functions()


def test_abi_version_and_device_count(ua):
    lib = ua.lib()
    header_version = int(re.search(r"#define\s+UNET_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.unet_abi_version() == header_version == ua._lib.ABI_VERSION
    assert lib.unet_device_count() >= 0


def test_argument_validation_without_gpu(ua):
    lib = ua.lib()
    # null pointers / bad shapes are rejected by the host-side checks (no launch happens)
    rc = lib.unet_conv3x3_fwd(None, 32, None, 0, None, None, None, 1, 8, 8, 32, 1, None)
    assert rc == -1 and b"null" in lib.unet_last_error()
    rc = lib.unet_conv3x3_fwd(1, 32, None, 0, 1, None, 1, 1, 8, 8, 48, 1, None)
    assert rc == -1 and b"multiple of 32" in lib.unet_last_error()
    rc = lib.unet_conv3x3_fwd(1, 32, None, 0, 1, None, 1, 1, 8, 8, 32, 3, None)
    assert rc == -1 and b"stride" in lib.unet_last_error()
    rc = lib.unet_conv3x3_bwd_data(1, 1, 64, 48, 1, 1, 8, 8, 32, 32, 1, 0, None)
    assert rc == -1   # slice 48..80 exceeds Cin_total 64
    rc = lib.unet_head1x1_fwd(1, 1, 1, 1, 1, 64, 16, 3, None)
    assert rc == -1 and b"C == 32" in lib.unet_last_error()
    rc = lib.unet_sgd_nesterov_step(4, 16, 16, 100, 0.1, 0.9, 0.0, 1, 1.0, None)
    assert rc == -1 and b"aligned" in lib.unet_last_error()


def test_workspace_queries(ua):
    lib = ua.lib()
    assert lib.unet_conv3x3_bwd_weight_workspace_bytes(8, 512, 512, 32, 32, 1) > 9 * 32 * 32 * 4
    assert lib.unet_conv3x3_bwd_weight_workspace_bytes(8, 512, 512, 3, 32, 1) > 0
    assert lib.unet_conv3x3_bwd_weight_workspace_bytes(0, 512, 512, 32, 32, 1) == 0
    assert lib.unet_instnorm_workspace_bytes(8, 512 * 512, 32) > 0
    assert lib.unet_head1x1_bwd_workspace_bytes(8, 512 * 512, 32, 3) > 0
    assert lib.unet_dice_wce_loss_workspace_bytes(8, 512, 512) > 0


def test_missing_library_fails_loudly(ua, monkeypatch, tmp_path):
    monkeypatch.setattr(ua._lib, "_lib", None)
    monkeypatch.setattr(ua._lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ua.UNetHipError, match="no CPU / eager fallback"):
        ua._lib.lib()
