"""Thin torch-tensor wrappers over the C ABI (one Python function per export).

Tensors are plumbing only: device memory, the current HIP stream and nothing
else.  Activations are NHWC fp32 (`[N, H, W, C]` contiguous).  Every function
launches asynchronously on `torch.cuda.current_stream()`.
"""
import torch

import ctypes

from ._lib import ActSrc, BwdStats, check, lib


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  Every entry-point call needs
    it (~330 per train step): the raw accessors cost ~0.3 us, `torch.cuda.current_stream()`
    builds a Stream object for ~10 us (3 ms of host time per step)."""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """HIP-event timing of kernel launches, recorded on the stream the kernels are launched on
    (torch's current stream).  bench.py installs one over its timed region to report
    `roofline.achieved` = algorithmic FLOPs / measured kernel time, the MFMA FLOPs actually
    issued (`executed`: lower where a gradient is reassociated onto the low-resolution grid) and
    the HBM rates of the streaming kernels (`nbytes` = algorithmic bytes)."""

    def __init__(self, only=None):
        # only: set of call-site classes ("conv" = 3x3 forward / data gradient, "wgrad") to
        # bracket; None = every entry point.  An event pair costs ~2.5 us of stream time (the
        # marker packets serialise the command processor), 0.8 ms per step over all ~330
        # calls - so bench.py brackets only the roofline group inside its timed region.
        self.only = only
        self.records = []  # (tag, flops, launches, start_event, end_event, executed, nbytes)

    def begin(self, kind=None):
        if self.only is not None and kind not in self.only:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, tag, flops, launches, start, executed=None, nbytes=0.0):
        if start is None:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.append((tag, flops, launches, start, ev,
                             flops if executed is None else executed, nbytes))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, flops, launches, e0, e1, executed, nbytes in self.records:
            d = out.setdefault(tag, dict(flops=0.0, executed=0.0, bytes=0.0, ms=0.0, launches=0,
                                         calls=0))
            d["flops"] += flops
            d["executed"] += executed
            d["bytes"] += nbytes
            d["ms"] += e0.elapsed_time(e1)
            d["launches"] += launches
            d["calls"] += 1
        return out


_timer = None
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def set_timer(timer):
    global _timer
    _timer = timer


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("unet-implementations_amd: the HIP path needs CUDA/ROCm tensors "
                           "(no CPU fallback exists)")
    if t.dtype not in (torch.float32, torch.int64, torch.uint8, torch.bfloat16):
        raise TypeError(f"unsupported dtype {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return t.data_ptr()


def _f32(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _b16(shape, like):
    return torch.empty(shape, dtype=torch.bfloat16, device=like.device)


def _is_b16(t):
    return t is not None and t.dtype == torch.bfloat16


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


# ---- deferred weight-gradient reductions (unet_wgrad_defer_*) -----------------------------------
import threading

_defer = threading.local()


def _wgrad_ws(nbytes, like):
    """Workspace of a weight-gradient call: while its reduction is only queued (wgrad_deferral)
    it must outlive the flush."""
    ws = _ws(nbytes, like)
    held = getattr(_defer, "held", None)
    if held is not None:
        held.append(ws)
    return ws


class wgrad_deferral:
    """`with wgrad_deferral() as d:` - the weight-gradient entry points called inside only queue
    the reductions of their slabs; `d.flush()` (and the end of the block) launches everything
    queued so far as 2-3 batched launches instead of 2-3 per layer.  A weight gradient is valid
    only after the flush that follows its call.  Per thread; not re-entrant."""

    def __enter__(self):
        if getattr(_defer, "held", None) is not None:
            raise RuntimeError("wgrad_deferral is not re-entrant")
        check(lib().unet_wgrad_defer_begin())
        _defer.held = []
        return self

    def flush(self):
        n = lib().unet_wgrad_defer_pending()
        if n:
            t0 = _timer.begin("wgrad") if _timer is not None else None
            check(lib().unet_wgrad_defer_flush(_stream()))
            if t0 is not None:
                _timer.end("conv_wgrad_reduce", 0.0, 2, t0)
        # (stream order: the allocator hands the blocks out again only behind the flush)
        _defer.held.clear()
        return n

    def __exit__(self, *exc):
        try:
            check(lib().unet_wgrad_defer_end(_stream()))
        finally:
            _defer.held = None
        return False


# ---- layout -------------------------------------------------------------------
def nchw_to_nhwc(x):
    N, C, H, W = x.shape
    y = _f32((N, H, W, C), x)
    check(lib().unet_nchw_to_nhwc(_ptr(x), _ptr(y), N, C, H, W, _stream()))
    return y


def nhwc_to_nchw(x):
    N, H, W, C = x.shape
    y = _f32((N, C, H, W), x)
    check(lib().unet_nhwc_to_nchw(_ptr(x), _ptr(y), N, C, H, W, _stream()))
    return y


def pack_conv3x3_weights(w_oihw, wf=None, wd=None, want_wd=True):
    Cout, Cin, kh, kw = w_oihw.shape
    assert kh == 3 and kw == 3
    if wf is None:
        wf = _f32((9, Cout, Cin), w_oihw)   # forward: reduction axis (ci) contiguous
    if wd is None and want_wd:
        wd = _f32((9, Cin, Cout), w_oihw)   # data gradient: reduction axis (co) contiguous
    check(lib().unet_pack_conv3x3_weights(_ptr(w_oihw), _ptr(wf), _ptr(wd), Cout, Cin, _stream()))
    return wf, wd


def pack_conv3x3_weights_bf16x3(w_oihw, want_wd=True, want_wf=True):
    """The two packed layouts as three bf16 planes each (the split-bf16 operand mode):
    wf3 [3, 9, Cout, Cin], wd3 [3, 9, Cin, Cout]."""
    Cout, Cin, kh, kw = w_oihw.shape
    assert kh == 3 and kw == 3
    wf3 = torch.empty((3, 9, Cout, Cin), dtype=torch.bfloat16, device=w_oihw.device) \
        if want_wf else None
    wd3 = torch.empty((3, 9, Cin, Cout), dtype=torch.bfloat16, device=w_oihw.device) \
        if want_wd else None
    check(lib().unet_pack_conv3x3_weights_bf16x3(_ptr(w_oihw), _ptr(wf3), _ptr(wd3), Cout, Cin,
                                                 _stream()))
    return wf3, wd3


class PackTable:
    """Device table for `unet_pack_conv3x3_weights_batched`: one entry per 3x3 layer with
    persistent destination buffers (fp32 layouts always, bf16x3 planes on request)."""

    def __init__(self, weights, planes, wino=None):
        """wino: per weight a pair (forward, data gradient) of flags - also keep the Winograd
        forms U = G g G^T of that layer (csrc/conv_wino.hip), refreshed by run()."""
        import struct
        self.wf, self.wd, self.wf3, self.wd3 = [], [], [], []
        self.src_ptrs = [w.data_ptr() for w in weights]
        self.planes = planes
        self.wino = tuple(wino) if wino is not None else tuple((False, False) for _ in weights)
        self.uf, self.ud, self._wino_jobs = [], [], []
        for w, (ff, fd) in zip(weights, self.wino):
            cout, cin = w.shape[0], w.shape[1]
            uf = _f32((16 * cout * cin,), w) if ff else None
            ud = _f32((16 * cout * cin,), w) if fd else None
            self.uf.append(uf)
            self.ud.append(ud)
            if uf is not None or ud is not None:
                self._wino_jobs.append((w, uf, ud, cout, cin))
        # device table of unet_pack_wino_weights_batched: one launch for every Winograd form
        self._wino_table, self._wino_blocks = None, 0
        if self._wino_jobs:
            rawu = b""
            for w, uf, ud, cout, cin in self._wino_jobs:
                rawu += struct.pack("<3Q4i", w.data_ptr(), 0 if uf is None else uf.data_ptr(),
                                    0 if ud is None else ud.data_ptr(), cout, cin,
                                    self._wino_blocks, 0)
                self._wino_blocks += (cout * cin // 8 + 255) // 256
            self._wino_table = torch.frombuffer(bytearray(rawu), dtype=torch.uint8).to(
                weights[0].device)
        raw = b""
        tiles = 0
        for w in weights:
            cout, cin = w.shape[0], w.shape[1]
            if cout % 32:
                raise ValueError("batched packing needs Cout to be a multiple of 32")
            wf = _f32((9, cout, cin), w)
            wd = _f32((9, cin, cout), w)
            wf3 = wd3 = None
            if planes and cin != 3:
                # planes: True / 3 = the three split planes, 1 = the bf16-rounded weight alone
                npl = 1 if planes == 1 and planes is not True else 3
                wf3 = torch.empty((npl, 9, cout, cin), dtype=torch.bfloat16, device=w.device)
                wd3 = torch.empty((npl, 9, cin, cout), dtype=torch.bfloat16, device=w.device)
            self.wf.append(wf); self.wd.append(wd); self.wf3.append(wf3); self.wd3.append(wd3)
            raw += struct.pack("<5Q4i", w.data_ptr(), wf.data_ptr(), wd.data_ptr(),
                               0 if wf3 is None else wf3.data_ptr(),
                               0 if wd3 is None else wd3.data_ptr(), cout, cin, tiles,
                               1 if (wf3 is not None and wf3.shape[0] == 1) else 0)
            tiles += (cout // 32) * ((cin + 31) // 32)
        self.n, self.tiles = len(weights), tiles
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(weights[0].device)

    def matches(self, weights, planes, wino=None):
        wino = tuple(wino) if wino is not None else tuple((False, False) for _ in weights)
        return planes == self.planes and len(weights) == self.n and wino == self.wino and \
            all(w.data_ptr() == p for w, p in zip(weights, self.src_ptrs))

    def run(self):
        check(lib().unet_pack_conv3x3_weights_batched(self.table.data_ptr(), self.n, self.tiles,
                                                      _stream()))
        if self._wino_table is not None:
            check(lib().unet_pack_wino_weights_batched(self._wino_table.data_ptr(),
                                                       len(self._wino_jobs), self._wino_blocks,
                                                       _stream()))


# ---- convolution ---------------------------------------------------------------
_PREC = {False: 0, True: 1, 0: 0, 1: 1, 3: 3, "fp32": 0, "bf16": 1, "bf16x3": 3}
_SUFFIX = {0: "", 1: "_bf16", 3: "_bf16x3"}
_GROUP = {0: "", 1: "_bf16", 3: "_bf16x3"}


def _prec(bf16):
    """Matrix-core operand mode of a conv call: fp32 MFMA (default), bf16 operands, or the
    split-bf16 emulation of fp32 ("bf16x3": 3 bf16 terms per operand, 6 products)."""
    try:
        return _PREC[bf16]
    except (KeyError, TypeError):
        raise ValueError("precision must be 'fp32', 'bf16' or 'bf16x3'") from None


def conv3x3_fwd(x0, x1, wf, bias, stride, out=None, bf16=False, wf3=None):
    N, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    if x1 is not None:
        assert x1.shape[:3] == x0.shape[:3]
    Cout = wf.shape[1]
    assert wf.shape[0] == 9 and wf.shape[2] == C0 + C1
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = out if out is not None else _f32((N, Ho, Wo, Cout), x0)
    t0 = _timer.begin("conv") if _timer is not None else None
    pr = _prec(bf16)
    if pr == 3 and C0 == 3:
        pr = 0                      # the RGB stem has no split form (K = 27, HBM-bound)
    fn = getattr(lib(), "unet_conv3x3_fwd" + _SUFFIX[pr])
    if pr == 3:
        if wf3 is None:
            raise ValueError("bf16x3 needs wf3 from pack_conv3x3_weights_bf16x3")
        check(fn(_ptr(x0), C0, _ptr(x1), C1, _ptr(wf), _ptr(wf3), _ptr(bias), _ptr(y), N, H, W,
                 Cout, stride, _stream()))
    else:
        check(fn(_ptr(x0), C0, _ptr(x1), C1, _ptr(wf), _ptr(bias), _ptr(y), N, H, W, Cout, stride,
                 _stream()))
    if t0 is not None:
        _timer.end("conv_stem_fwd" if C0 == 3 else "conv_igemm" + _GROUP[pr],
                   2.0 * N * Ho * Wo * 9 * (C0 + C1) * Cout, 1, t0)
    return y


class NextNorm:
    """Layer l as seen by the producer of g = dL/da_l: raw output y, statistics, affine
    parameters, dropout mask.  A data gradient whose output is FINAL for layer l takes one and
    leaves the per-tile reductions of l's InstanceNorm backward in `.partial` / `.tiles`
    (tiles == 0: that launch had no such epilogue)."""

    __slots__ = ("y", "st", "gamma", "beta", "mask", "slope", "partial", "tiles")

    def __init__(self, y, st, gamma, beta, mask, slope):
        self.y, self.st, self.gamma, self.beta, self.mask, self.slope = y, st, gamma, beta, mask, slope
        self.partial, self.tiles = None, 0

    def c_struct(self):
        N, H, W, C = self.y.shape
        nbytes = N * ((H * W + 63) // 64) * C * 8
        self.partial = torch.empty(nbytes, dtype=torch.uint8, device=self.y.device)
        return BwdStats(_ptr(self.y), _ptr(self.st[0]), _ptr(self.st[1]), _ptr(self.gamma),
                        _ptr(self.beta), _ptr(self.mask), self.slope, _ptr(self.partial), nbytes, 0)


def _c32_winograd(N, H, W, Cin, Cout, stride):
    """True when the fp32 32 -> 32 channel kernel of this shape runs in its Winograd form
    (csrc/conv_c32.hip; `set_c32_winograd`): 16/36 of the direct kernel's MFMA FLOPs."""
    return bool(lib().unet_conv_c32_is_winograd(N, H, W, Cin, Cout, stride))


_c32_override = None


def c32_winograd_override():
    """The setting a direct `set_c32_winograd(...)` call left behind (tests, benchmarks), or None."""
    return _c32_override


class c32_winograd_scope:
    """`with c32_winograd_scope(mode):` - the calling thread's switch set to `mode` (True /
    False / "always") for the block and put back afterwards: how UNet.forward / backward pass
    their model's choice to the entry points they call without leaving it behind."""

    def __init__(self, mode):
        self.code = 2 if mode == "always" else (1 if mode else 0)

    def __enter__(self):
        self.prev = lib().unet_set_c32_winograd(self.code)
        return self

    def __exit__(self, *exc):
        lib().unet_set_c32_winograd(self.prev)
        return False


def set_c32_winograd(on, override=True):
    """Choice for the 32 -> 32 channel stride-1 layers on the fused fp32 pipeline: True (default)
    = the Winograd kernel for launches that fill the chip (>= 512 tiles of 8 x 32 pixels),
    "always" = for every shape it tiles, False = the direct kernel.  Returns the previous setting
    of the calling thread in the same terms.  The library keeps the switch PER THREAD;
    `UNet.forward` applies `model.winograd` to its own calls only (`c32_winograd_scope`) and
    hands the decision to its backward.  A direct call (override=True) is a test / benchmark
    override for kernel-level calls: "always" then also wins over `model.winograd`."""
    global _c32_override
    code = 2 if on == "always" else (1 if on else 0)
    prev = lib().unet_set_c32_winograd(code)
    if override:
        _c32_override = on if on == "always" else None
    return "always" if prev == 2 else bool(prev)


def conv3x3_bwd_data(dy, wd, ci_offset, ccols, H, W, stride, out=None, accumulate=False,
                     bf16=False, wd3=None, nxt=None, ud=None):
    """dx[N,H,W,ccols] (+)= transpose-conv of dy for input channels [ci_offset, ci_offset+ccols).
    nxt (NextNorm, fp32 path): dx is final for that layer - also emit its backward reductions.
    ud: the Winograd data-gradient form of the weight (fp32 tensors, stride 1, no accumulate,
    shape checked by the caller with conv_wino_supported)."""
    N, Ho, Wo, Cout = dy.shape
    cin_total = wd.shape[1]
    assert wd.shape[0] == 9 and wd.shape[2] == Cout
    if ud is not None and not _is_b16(dy) and stride == 1 and not accumulate and _prec(bf16) == 0:
        dx = out if out is not None else _f32((N, H, W, ccols), dy)
        assert dx.shape == (N, H, W, ccols)
        bs = nxt.c_struct() if nxt is not None else None
        t0 = _timer.begin("conv") if _timer is not None else None
        check(lib().unet_conv3x3_bwd_data_bs_wino(_ptr(dy), _ptr(ud), cin_total, ci_offset,
                                                  _ptr(dx), N, H, W, Cout, ccols,
                                                  ctypes.byref(bs) if bs is not None else None,
                                                  _stream()))
        if nxt is not None:
            nxt.tiles = bs.tiles_out
        if t0 is not None:
            alg = 2.0 * N * H * W * 9 * ccols * Cout
            _timer.end("conv_igemm", alg, 1, t0, executed=alg * 16.0 / 36.0)
        return dx
    if _is_b16(dy):     # mixed-precision pipeline: bf16 tensors, bf16 matrix cores
        dx = out if out is not None else _b16((N, H, W, ccols), dy)
        assert dx.shape == (N, H, W, ccols) and _is_b16(dx)
        t0 = _timer.begin("conv") if _timer is not None else None
        if nxt is not None or wd3 is not None:
            # (wd3: the weights also pre-rounded to bf16 - plane 0 of the pack's planes)
            bs = nxt.c_struct() if nxt is not None else None
            check(lib().unet_conv3x3_bwd_data_bs_b16_wb(
                _ptr(dy), _ptr(wd), _ptr(wd3), cin_total, ci_offset, _ptr(dx), N, H, W, Cout, ccols,
                stride, 1 if accumulate else 0, ctypes.byref(bs) if bs is not None else None,
                _stream()))
            if nxt is not None:
                nxt.tiles = bs.tiles_out
        else:
            check(lib().unet_conv3x3_bwd_data_b16(_ptr(dy), _ptr(wd), cin_total, ci_offset, _ptr(dx),
                                                  N, H, W, Cout, ccols, stride,
                                                  1 if accumulate else 0, _stream()))
        if t0 is not None:
            _timer.end("conv_igemm_bf16", 2.0 * N * Ho * Wo * 9 * ccols * Cout,
                       1 if stride == 1 else 4, t0)
        return dx
    dx = out if out is not None else _f32((N, H, W, ccols), dy)
    assert dx.shape == (N, H, W, ccols)
    pr = _prec(bf16)
    if nxt is not None and pr in (0, 3):
        bs = nxt.c_struct()
        t0 = _timer.begin("conv") if _timer is not None else None
        if pr == 3:
            check(lib().unet_conv3x3_bwd_data_bs_bf16x3(
                _ptr(dy), _ptr(wd), _ptr(wd3), cin_total, ci_offset, _ptr(dx), N, H, W, Cout,
                ccols, stride, 1 if accumulate else 0, ctypes.byref(bs), _stream()))
        else:
            check(lib().unet_conv3x3_bwd_data_bs(_ptr(dy), _ptr(wd), cin_total, ci_offset, _ptr(dx),
                                                 N, H, W, Cout, ccols, stride,
                                                 1 if accumulate else 0, ctypes.byref(bs),
                                                 _stream()))
        nxt.tiles = bs.tiles_out
        if t0 is not None:
            alg = 2.0 * N * Ho * Wo * 9 * ccols * Cout
            c32w = pr == 0 and _c32_winograd(N, H, W, ccols, Cout, stride)
            _timer.end("conv_igemm" + _GROUP[pr], alg, 1, t0,
                       executed=alg * 16.0 / 36.0 if c32w else None)
        return dx
    t0 = _timer.begin("conv") if _timer is not None else None
    fn = getattr(lib(), "unet_conv3x3_bwd_data" + _SUFFIX[pr])
    if pr == 3:
        if wd3 is None:
            raise ValueError("bf16x3 needs wd3 from pack_conv3x3_weights_bf16x3")
        check(fn(_ptr(dy), _ptr(wd), _ptr(wd3), cin_total, ci_offset, _ptr(dx), N, H, W, Cout,
                 ccols, stride, 1 if accumulate else 0, _stream()))
    else:
        check(fn(_ptr(dy), _ptr(wd), cin_total, ci_offset, _ptr(dx), N, H, W, Cout, ccols, stride,
                 1 if accumulate else 0, _stream()))
    if t0 is not None:
        launches = 1
        if stride == 2:  # fp32: one launch when >= 512 tiles, else one per output parity class
            tiles = -(-(N * Ho * Wo) // 128) * (ccols // 32)
            launches = 1 if (pr == 0 and tiles >= 512) else 4
        alg = 2.0 * N * Ho * Wo * 9 * ccols * Cout
        c32w = pr == 0 and _c32_winograd(N, H, W, ccols, Cout, stride)
        _timer.end("conv_igemm" + _GROUP[pr], alg, launches, t0,
                   executed=alg * 16.0 / 36.0 if c32w else None)
    return dx


def conv3x3_bwd_weight(x, dy, dw_oihw, ci_offset, stride, db=None, bf16=False):
    N, H, W, Cx = x.shape
    Cout = dy.shape[3]
    cin_total = dw_oihw.shape[1]
    assert dw_oihw.shape[0] == Cout and dw_oihw.is_contiguous()
    nbytes = lib().unet_conv3x3_bwd_weight_workspace_bytes(N, H, W, Cx, Cout, stride)
    ws = _wgrad_ws(nbytes, x)
    t0 = _timer.begin("wgrad") if _timer is not None else None
    pr = _prec(bf16)
    fn = getattr(lib(), "unet_conv3x3_bwd_weight" + _SUFFIX[pr])
    check(fn(_ptr(x), Cx, _ptr(dy), _ptr(dw_oihw), ci_offset, cin_total, _ptr(db), _ptr(ws),
             ws.numel(), N, H, W, Cout, stride, _stream()))
    if t0 is not None:  # wgrad kernel + slab reduce
        Ho, Wo = dy.shape[1], dy.shape[2]
        _timer.end("conv_stem_wgrad" if Cx == 3 else "conv_wgrad" + _GROUP[pr],
                   2.0 * N * Ho * Wo * 9 * Cx * Cout, 2, t0)
    return dw_oihw


# ---- 1x1 convolution (CLIP fusion layer) ------------------------------------------------
def conv1x1_fwd(x0, x1, w2d, bias):
    """y = conv1x1(cat(x0, x1)) + bias; w2d is [Cout, C0+C1]."""
    N, H, W, C0 = x0.shape
    C1 = 0 if x1 is None else x1.shape[3]
    Cout = w2d.shape[0]
    assert w2d.shape[1] == C0 + C1 and w2d.is_contiguous()
    y = _f32((N, H, W, Cout), x0)
    check(lib().unet_conv1x1_fwd(_ptr(x0), C0, _ptr(x1), C1, _ptr(w2d), _ptr(bias), _ptr(y), N, H,
                                 W, Cout, _stream()))
    return y


def transpose2d(w2d):
    R, C = w2d.shape
    out = _f32((C, R), w2d)
    check(lib().unet_transpose2d(_ptr(w2d), _ptr(out), R, C, _stream()))
    return out


def conv1x1_bwd_data(dy, wT, ci_offset, ccols, accumulate=False, out=None):
    N, H, W, Cout = dy.shape
    cin_total = wT.shape[0]
    assert wT.shape[1] == Cout
    dx = out if out is not None else _f32((N, H, W, ccols), dy)
    check(lib().unet_conv1x1_bwd_data(_ptr(dy), _ptr(wT), cin_total, ci_offset, _ptr(dx), N, H, W,
                                      Cout, ccols, 1 if accumulate else 0, _stream()))
    return dx


def conv1x1_bwd_weight(x, dy, dw2d, ci_offset):
    N, H, W, Cx = x.shape
    Cout = dy.shape[3]
    assert dw2d.shape[0] == Cout and dw2d.is_contiguous()
    ws = _wgrad_ws(lib().unet_conv3x3_bwd_weight_workspace_bytes(N, H, W, Cx, Cout, 1), x)
    check(lib().unet_conv1x1_bwd_weight(_ptr(x), Cx, _ptr(dy), _ptr(dw2d), ci_offset,
                                        dw2d.shape[1], _ptr(ws), ws.numel(), N, H, W, Cout,
                                        _stream()))
    return dw2d


# ---- InstanceNorm + LeakyReLU + channel dropout -----------------------------------
def instnorm_stats(y, gamma, beta, eps):
    N, H, W, C = y.shape
    st = _f32((4, N, C), y)  # mean, rstd, alpha, beta2
    ws = _ws(lib().unet_instnorm_workspace_bytes(N, H * W, C), y)
    check(lib().unet_instnorm_stats(_ptr(y), _ptr(gamma), _ptr(beta), eps, _ptr(st[0]),
                                    _ptr(st[1]), _ptr(st[2]), _ptr(st[3]), _ptr(ws), ws.numel(), N,
                                    H * W, C, _stream()))
    return st


def instnorm_lrelu_drop_fwd(y, alpha, beta2, mask, slope, out=None):
    N, H, W, C = y.shape
    a = out if out is not None else torch.empty_like(y)
    check(lib().unet_instnorm_lrelu_drop_fwd(_ptr(y), _ptr(alpha), _ptr(beta2), _ptr(mask), slope,
                                             _ptr(a), N, H * W, C, _stream()))
    return a


def instnorm_lrelu_drop_bwd(ga, y, mean, rstd, gamma, beta, mask, slope, dgamma, dbeta, dbias,
                            out=None, partials=None):
    """Returns dy (in place over `ga` unless `out` is given).  partials = (buffer, tiles): the
    reductions were already summarised per tile by the producer of ga (NextNorm)."""
    N, H, W, C = y.shape
    dy = ga if out is None else out
    ws = _ws(lib().unet_instnorm_workspace_bytes(N, H * W, C), y)
    if partials is not None:
        fnp = lib().unet_instnorm_lrelu_drop_bwd_partials_b16 if _is_b16(y) else \
            lib().unet_instnorm_lrelu_drop_bwd_partials
        t0 = _timer.begin() if _timer is not None else None
        check(fnp(
            _ptr(ga), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(mask), slope,
            _ptr(dy), _ptr(dgamma), _ptr(dbeta), _ptr(dbias), _ptr(partials[0]), partials[1],
            _ptr(ws), ws.numel(), N, H * W, C, _stream()))
        if t0 is not None:   # apply pass only: ga + y in, dy out
            _timer.end("instnorm_bwd", 0.0, 2, t0, nbytes=y.element_size() * 3.0 * y.numel())
        return dy
    fn = lib().unet_instnorm_lrelu_drop_bwd_b16 if _is_b16(y) else lib().unet_instnorm_lrelu_drop_bwd
    if _is_b16(y) != _is_b16(ga):
        raise TypeError("ga and y must share their storage type")
    t0 = _timer.begin() if _timer is not None else None
    check(fn(_ptr(ga), _ptr(y), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), _ptr(mask),
             slope, _ptr(dy), _ptr(dgamma), _ptr(dbeta), _ptr(dbias), _ptr(ws), ws.numel(), N,
             H * W, C, _stream()))
    if t0 is not None:   # reduce pass: ga + y; apply pass: ga + y in, dy out (+ tiny finalizers)
        _timer.end("instnorm_bwd", 0.0, 5, t0, nbytes=y.element_size() * 5 * y.numel())
    return dy


def instnorm_bwd_coefs(y, mean, rstd, gamma, beta, mask, partials):
    """Apply-on-load form of the InstanceNorm + LeakyReLU + dropout backward: from the per-tile
    reductions `partials` = (buffer, tiles) the coefficient planes coef5 [5, N, C] and the sums
    [N, C, 2] that `conv3x3_bwd_data_dz` consumes (no elementwise pass over the layer tensor)."""
    N, H, W, C = y.shape
    coef5 = _f32((5, N, C), y)
    sums = _f32((N, C, 2), y)
    check(lib().unet_instnorm_bwd_coefs(_ptr(partials[0]), partials[1], _ptr(mean), _ptr(rstd),
                                        _ptr(gamma), _ptr(beta), _ptr(mask), _ptr(coef5),
                                        _ptr(sums), N, H * W, C, _stream()))
    return coef5, sums


def conv_in_bwd_weight_dz_supported(N, H, W, Cx, Cout):
    """The fp32 weight gradient of this (stride-1, 3x3) layer can apply the layer's InstanceNorm
    backward on load (32 -> 32 channels on the Winograd kernel: csrc/conv_wgrad.hip)?"""
    return bool(lib().unet_conv_in_bwd_weight_dz_supported(N, H, W, Cx, Cout))


def conv_in_bwd_weight_dz(x, slope, g, y, coef5, sums, gamma, rstd, dz_slope, dgamma, dbeta, dbias,
                          dw_oihw, ci_offset, in_place=True):
    """Weight gradient of a fused 32 -> 32 channel layer whose dy side forms dz = dL/dy from
    (g, y) on load (unet_conv_in_bwd_weight_dz).  Returns dz (written over g unless in_place is
    False) for the layer's data gradient; dgamma / dbeta / dbias are filled."""
    x, rx = _act(x)
    N, H, W, Cx = x.shape
    Cout = g.shape[3]
    assert dw_oihw.shape[0] == Cout and dw_oihw.is_contiguous() and y.shape == g.shape
    ws = _wgrad_ws(lib().unet_conv3x3_bwd_weight_workspace_bytes(N, H, W, Cx, Cout, 1), g)
    dz = g if in_place else _f32(tuple(g.shape), g)
    t0 = _timer.begin("wgrad") if _timer is not None else None
    check(lib().unet_conv_in_bwd_weight_dz(
        rx, slope, _ptr(g), _ptr(y), _ptr(coef5), _ptr(sums), _ptr(gamma), _ptr(rstd), dz_slope,
        _ptr(dz), _ptr(dgamma), _ptr(dbeta), _ptr(dbias), _ptr(dw_oihw), ci_offset,
        dw_oihw.shape[1], _ptr(ws), ws.numel(), N, H, W, Cout, _stream()))
    if t0 is not None:
        alg = 2.0 * N * H * W * 9 * Cx * Cout
        _timer.end("conv_wgrad", alg, 2, t0, executed=alg * 16.0 / 36.0)
    return dz


def conv3x3_bwd_data_dz(g, y, coef5, sums, gamma, rstd, slope, dgamma, dbeta, dbias, ud,
                        cin_total, ci_offset, ccols, nxt=None):
    """Winograd data gradient whose loader applies the InstanceNorm backward of the layer to
    (g, y) on the fly (unet_conv3x3_bwd_data_dz_wino).  Returns (dx, dz): dz = dL/dy of the
    layer, written as a by-product for its weight gradient; dgamma / dbeta / dbias are filled."""
    N, H, W, Cout = g.shape
    dx = _f32((N, H, W, ccols), g)
    dz = _f32((N, H, W, Cout), g)
    bs = nxt.c_struct() if nxt is not None else None
    t0 = _timer.begin("conv") if _timer is not None else None
    check(lib().unet_conv3x3_bwd_data_dz_wino(
        _ptr(g), _ptr(y), _ptr(coef5), _ptr(sums), _ptr(gamma), _ptr(rstd), slope, _ptr(dz),
        _ptr(dgamma), _ptr(dbeta), _ptr(dbias), _ptr(ud), cin_total, ci_offset, _ptr(dx), N, H, W,
        Cout, ccols, ctypes.byref(bs) if bs is not None else None, _stream()))
    if nxt is not None:
        nxt.tiles = bs.tiles_out
    if t0 is not None:
        alg = 2.0 * N * H * W * 9 * ccols * Cout
        _timer.end("conv_igemm", alg, 1, t0, executed=alg * 16.0 / 36.0)
    return dx, dz


class _ResizeBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        n, c, h, w = x.shape
        x = x.contiguous().float()
        y = _f32((n, c, H, W), x)
        check(lib().unet_resize_bilinear_fwd(_ptr(x), _ptr(y), n * c, h, w, H, W, _stream()))
        ctx.hw = (h, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        n, c, H, W = gy.shape
        h, w = ctx.hw
        gy = gy.contiguous().float()
        gx = _f32((n, c, h, w), gy)
        check(lib().unet_resize_bilinear_bwd(_ptr(gy), _ptr(gx), n * c, h, w, H, W, _stream()))
        return gx, None, None


def resize_bilinear(x, size):
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False) of an NCHW fp32 tensor
    on the HIP kernels (differentiable; the backward is the deterministic gather-form adjoint)."""
    return _ResizeBilinear.apply(x, int(size[0]), int(size[1]))


# ---- fused layer pipeline (include/unet_hip.h) ---------------------------------------------
class Act:
    """An operand that is activated on load: `x` is the RAW output of a convolution (NHWC)
    and the consumer applies `lrelu(x * alpha[n, c] + beta[n, c])` while staging it, or a
    plain tensor (`alpha is None`).  alpha / beta are rows 2 and 3 of the statistics tensor
    `conv_in_fwd` returns (InstanceNorm scale / shift with the dropout mask folded in)."""

    __slots__ = ("x", "alpha", "beta")

    def __init__(self, x, alpha=None, beta=None):
        self.x, self.alpha, self.beta = x, alpha, beta

    @property
    def shape(self):
        return self.x.shape

    def c_struct(self):
        return ActSrc(_ptr(self.x), self.x.shape[3], _ptr(self.alpha), _ptr(self.beta))


class U8Image:
    """The dataset's uint8 HWC batch [N, H, W, 3] as the operand of the RGB stem: normalised
    ((v / 255) - mean) / std inside the loaders of the first convolution and of its weight
    gradient (unet_stem_u8_fwd / _bwd_weight), never materialised as fp32."""

    __slots__ = ("x", "mean", "std")

    def __init__(self, x, mean=None, std=None):
        if x.dtype != torch.uint8 or x.dim() != 4 or x.shape[3] != 3 or not x.is_contiguous():
            raise TypeError("U8Image takes a contiguous uint8 [N,H,W,3] tensor")
        if x.shape[2] % 128:
            raise ValueError("the fused uint8 stem needs W % 128 == 0 (use preprocess_u8 otherwise)")
        self.x = x
        self.mean = tuple(IMAGENET_MEAN if mean is None else mean)
        self.std = tuple(IMAGENET_STD if std is None else std)

    @property
    def shape(self):
        return self.x.shape

    def c_mean_std(self):
        return (ctypes.c_float * 3)(*self.mean), (ctypes.c_float * 3)(*self.std)


def _act(a):
    if a is None:
        return None, None
    if not isinstance(a, Act):
        a = Act(a)
    st = a.c_struct()
    return a, ctypes.byref(st)


def conv_wino_supported(N, H, W, C0, C1, Cout):
    """Does the Winograd kernel tile conv3x3(stride 1) of [N,H,W,C0+C1] -> Cout (forward), or the
    data gradient with K = C0 reduction channels and Cout columns?"""
    return bool(lib().unet_conv_wino_supported(N, H, W, C0, C1, Cout))


def pack_wino_weights(w_oihw, want_f=True, want_d=True):
    """(uf, ud): the Winograd forms of a 3x3 weight (flat fp32 tensors of 16*Cout*Cin floats)."""
    cout, cin = w_oihw.shape[0], w_oihw.shape[1]
    uf = _f32((16 * cout * cin,), w_oihw) if want_f else None
    ud = _f32((16 * cout * cin,), w_oihw) if want_d else None
    check(lib().unet_pack_wino_weights(_ptr(w_oihw), _ptr(uf), _ptr(ud), cout, cin, _stream()))
    return uf, ud


def conv_in_fwd(s0, s1, slope, w, bias, ksize, stride, gamma, beta, eps, mask, b16=False,
                w3=None, wu=None):
    """Fused layer forward: y = conv(cat(act(s0), act(s1))) + bias and the InstanceNorm
    statistics of y.  Returns (y, st) with st = [mean, rstd, alpha, beta] as [4, N, Cout];
    alpha / beta carry the dropout `mask` [N, Cout] (or None) folded in.
    b16: the mixed-precision pipeline - y (and the sources other than the fp32 RGB image) are
    bf16 tensors, bf16 matrix cores, fp32 statistics.
    w3 (pre-split weight planes): the split-bf16 operand mode on fp32 tensors."""
    u8 = s0 if isinstance(s0, U8Image) else None
    if u8 is None:
        s0, r0 = _act(s0)
    s1, r1 = _act(s1)
    N, H, W, C0 = s0.shape
    C1 = 0 if s1 is None else s1.shape[3]
    if s1 is not None:
        assert s1.shape[:3] == s0.shape[:3]
    Cout = w.shape[1] if ksize == 3 else w.shape[0]
    assert (w.shape[0] == 9 and w.shape[2] == C0 + C1) if ksize == 3 else w.shape[1] == C0 + C1
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if b16:
        for src in (s0, s1):
            if src is not None and src.shape[3] != 3 and not _is_b16(src.x):
                raise TypeError("the bf16 pipeline takes bf16 layer tensors")
        if u8 is not None:
            raise NotImplementedError("uint8 stem on the bf16 pipeline")
    y = (_b16 if b16 else _f32)((N, Ho, Wo, Cout), s0.x)
    st = _f32((4, N, Cout), s0.x)
    ws = _ws(lib().unet_conv_in_fwd_workspace_bytes(N, H, W, Cout, stride), s0.x)
    fwd = lib().unet_conv_in_fwd_b16 if b16 else lib().unet_conv_in_fwd
    fin = lib().unet_conv_in_stats_finalize_b16 if b16 else lib().unet_conv_in_stats_finalize
    px = ctypes.c_int(0)
    t0 = _timer.begin("conv") if _timer is not None else None
    if u8 is not None:
        assert s1 is None and ksize == 3 and stride == 1
        m3, s3 = u8.c_mean_std()
        check(lib().unet_stem_u8_fwd(_ptr(u8.x), m3, s3, _ptr(w), _ptr(bias), _ptr(y), _ptr(ws),
                                     ws.numel(), ctypes.byref(px), N, H, W, Cout, _stream()))
    elif wu is not None and not b16 and ksize == 3 and stride == 1:
        # Winograd F(2x2, 3x3) form (the caller checked conv_wino_supported)
        check(lib().unet_conv_in_fwd_wino(r0, r1, slope, _ptr(wu), _ptr(bias), _ptr(y), _ptr(ws),
                                          ws.numel(), ctypes.byref(px), N, H, W, Cout, _stream()))
    elif w3 is not None and b16 and ksize == 3 and C0 != 3:
        # mixed precision with the weights also pre-rounded to bf16 (plane 0 of the planes)
        check(lib().unet_conv_in_fwd_b16_wb(r0, r1, slope, _ptr(w), _ptr(w3), _ptr(bias), ksize,
                                            stride, _ptr(y), _ptr(ws), ws.numel(),
                                            ctypes.byref(px), N, H, W, Cout, _stream()))
    elif w3 is not None and not b16 and C0 != 3:
        check(lib().unet_conv_in_fwd_bf16x3(r0, r1, slope, _ptr(w), _ptr(w3), _ptr(bias), ksize,
                                            stride, _ptr(y), _ptr(ws), ws.numel(),
                                            ctypes.byref(px), N, H, W, Cout, _stream()))
    else:
        check(fwd(r0, r1, slope, _ptr(w), _ptr(bias), ksize, stride, _ptr(y), _ptr(ws),
                  ws.numel(), ctypes.byref(px), N, H, W, Cout, _stream()))
    if t0 is not None:   # the convolution launch alone (its epilogue includes the statistics)
        alg = 2.0 * N * Ho * Wo * ksize * ksize * (C0 + C1) * Cout
        wino = wu is not None and not b16 and ksize == 3 and stride == 1
        if not wino and not b16 and w3 is None and u8 is None and ksize == 3 and s1 is None:
            wino = _c32_winograd(N, H, W, C0, Cout, stride)
        _timer.end("conv_stem_fwd" if C0 == 3 else
                   ("conv_igemm_bf16" if b16 else ("conv_igemm_bf16x3" if w3 is not None else "conv_igemm")),
                   alg, 1, t0, executed=alg * 16.0 / 36.0 if wino else None)
    check(fin(_ptr(y), _ptr(ws), ws.numel(), px.value, _ptr(gamma), _ptr(beta), eps, _ptr(mask),
              _ptr(st[0]), _ptr(st[1]), _ptr(st[2]), _ptr(st[3]), N, Ho * Wo, Cout, _stream()))
    return y, st


def conv_up_in_fwd_supported(low, skip, Cout):
    N, H, W, C1 = skip.shape
    if _is_b16(skip.x):     # mixed-precision pipeline: both sources activated bf16 tensors
        return low.alpha is not None and skip.alpha is not None and _is_b16(low.x) and \
            bool(lib().unet_conv_up_in_fwd_b16_supported(N, H, W, low.shape[3], C1, Cout))
    return bool(lib().unet_conv_up_in_fwd_supported(N, H, W, low.shape[3], C1, Cout))


def conv_up_wino_supported(N, H, W, C0, C1, Cout):
    """Winograd form of conv_up_in_fwd for output [N,H,W,Cout], low-res source C0, skip C1?"""
    return bool(lib().unet_conv_up_wino_supported(N, H, W, C0, C1, Cout))


def conv_up_in_fwd(low, skip, slope, wf, bias, gamma, beta, eps, mask, wu=None, w3=None):
    """y = conv3x3(cat(upsample2x(act(low)), act(skip))) + bias with the up-sampling in the
    loader, plus the InstanceNorm statistics of y (as conv_in_fwd).  wu: the Winograd forward
    form of the weight (shape checked by the caller with conv_up_wino_supported).  bf16 sources
    (the mixed-precision pipeline): y is bf16, w3 = the bf16-rounded weight plane or None."""
    low, rl = _act(low)
    skip, rs = _act(skip)
    N, H, W, C1 = skip.shape
    C0 = low.shape[3]
    assert low.shape[1] * 2 == H and low.shape[2] * 2 == W and low.shape[0] == N
    Cout = wf.shape[1]
    assert wf.shape[0] == 9 and wf.shape[2] == C0 + C1
    if _is_b16(skip.x):
        y = _b16((N, H, W, Cout), skip.x)
        st = _f32((4, N, Cout), skip.x)
        ws = _ws(lib().unet_conv_in_fwd_workspace_bytes(N, H, W, Cout, 1), skip.x)
        px = ctypes.c_int(0)
        t0 = _timer.begin("conv") if _timer is not None else None
        check(lib().unet_conv_up_in_fwd_b16(rl, rs, slope, _ptr(wf), _ptr(w3), _ptr(bias), _ptr(y),
                                            _ptr(ws), ws.numel(), ctypes.byref(px), N, H, W, Cout,
                                            _stream()))
        if t0 is not None:
            _timer.end("conv_igemm_bf16", 2.0 * N * H * W * 9 * (C0 + C1) * Cout, 1, t0)
        check(lib().unet_conv_in_stats_finalize_b16(_ptr(y), _ptr(ws), ws.numel(), px.value,
                                                    _ptr(gamma), _ptr(beta), eps, _ptr(mask),
                                                    _ptr(st[0]), _ptr(st[1]), _ptr(st[2]),
                                                    _ptr(st[3]), N, H * W, Cout, _stream()))
        return y, st
    y = _f32((N, H, W, Cout), skip.x)
    st = _f32((4, N, Cout), skip.x)
    ws = _ws(lib().unet_conv_in_fwd_workspace_bytes(N, H, W, Cout, 1), skip.x)
    px = ctypes.c_int(0)
    t0 = _timer.begin("conv") if _timer is not None else None
    if wu is not None:
        check(lib().unet_conv_up_in_fwd_wino(rl, rs, slope, _ptr(wu), _ptr(bias), _ptr(y),
                                             _ptr(ws), ws.numel(), ctypes.byref(px), N, H, W, Cout,
                                             _stream()))
    else:
        check(lib().unet_conv_up_in_fwd(rl, rs, slope, _ptr(wf), _ptr(bias), _ptr(y), _ptr(ws),
                                        ws.numel(), ctypes.byref(px), N, H, W, Cout, _stream()))
    if t0 is not None:
        alg = 2.0 * N * H * W * 9 * (C0 + C1) * Cout
        wino = wu is not None or (low.alpha is not None and skip.alpha is not None and bool(
            lib().unet_conv_up_c32_is_winograd(N, H, W, C0, C1, Cout)))
        _timer.end("conv_igemm", alg, 1, t0, executed=alg * 16.0 / 36.0 if wino else None)
    check(lib().unet_conv_in_stats_finalize(_ptr(y), _ptr(ws), ws.numel(), px.value, _ptr(gamma),
                                            _ptr(beta), eps, _ptr(mask), _ptr(st[0]), _ptr(st[1]),
                                            _ptr(st[2]), _ptr(st[3]), N, H * W, Cout, _stream()))
    return y, st


def conv_in_bwd_weight(x, slope, dy, dw_oihw, ci_offset, ksize, stride, x3=False):
    """Weight gradient of a fused layer: dw[:, ci_offset : ci_offset + Cx] = act(x) (x) dy.
    x3: the split-bf16 operand mode (fp32 tensors)."""
    if isinstance(x, U8Image):
        N, H, W, _ = x.shape
        Cout = dy.shape[3]
        ws = _wgrad_ws(lib().unet_conv3x3_bwd_weight_workspace_bytes(N, H, W, 3, Cout, 1), dy)
        m3, s3 = x.c_mean_std()
        t0 = _timer.begin("wgrad") if _timer is not None else None
        check(lib().unet_stem_u8_bwd_weight(_ptr(x.x), m3, s3, _ptr(dy), _ptr(dw_oihw), _ptr(ws),
                                            ws.numel(), N, H, W, Cout, _stream()))
        if t0 is not None:
            _timer.end("conv_stem_wgrad", 2.0 * N * H * W * 27 * Cout, 2, t0)
        return dw_oihw
    x, rx = _act(x)
    N, H, W, Cx = x.shape
    Cout = dy.shape[3]
    assert dw_oihw.shape[0] == Cout and dw_oihw.is_contiguous()
    ws = _wgrad_ws(lib().unet_conv3x3_bwd_weight_workspace_bytes(N, H, W, Cx, Cout, stride), dy)
    b16 = _is_b16(dy)
    fn = lib().unet_conv_in_bwd_weight_b16 if b16 else \
        (lib().unet_conv_in_bwd_weight_bf16x3 if x3 else lib().unet_conv_in_bwd_weight)
    t0 = _timer.begin("wgrad") if _timer is not None else None
    check(fn(rx, slope, _ptr(dy), _ptr(dw_oihw), ci_offset, dw_oihw.shape[1], ksize, stride,
             _ptr(ws), ws.numel(), N, H, W, Cout, _stream()))
    if t0 is not None:
        alg = 2.0 * N * dy.shape[1] * dy.shape[2] * ksize * ksize * Cx * Cout
        wino = not b16 and not x3 and ksize == 3 and \
            bool(lib().unet_conv3x3_bwd_weight_is_winograd(N, H, W, Cx, Cout, stride))
        _timer.end("conv_stem_wgrad" if Cx == 3 else
                   ("conv_wgrad_bf16" if b16 else ("conv_wgrad_bf16x3" if x3 else "conv_wgrad")),
                   alg, 2, t0, executed=alg * 16.0 / 36.0 if wino else None)
    return dw_oihw


def upsample2x_in_fwd(x, slope):
    x, rx = _act(x)
    N, h, w, C = x.shape
    b16 = _is_b16(x.x)
    up = (_b16 if b16 else _f32)((N, 2 * h, 2 * w, C), x.x)
    t0 = _timer.begin() if _timer is not None else None
    fn = lib().unet_upsample2x_in_fwd_b16 if b16 else lib().unet_upsample2x_in_fwd
    check(fn(rx, slope, _ptr(up), N, h, w, _stream()))
    if t0 is not None:
        _timer.end("upsample2x_fwd", 0.0, 1, t0,
                   nbytes=x.x.element_size() * (x.x.numel() + up.numel()))
    return up


def upsample2x_bwd_taps(dy):
    """D[N, h, w, 9*C] = the nine transposed-upsampled shifts of dy[N, 2h, 2w, C] (tap-major)."""
    N, H2, W2, C = dy.shape
    b16 = _is_b16(dy)
    D = (_b16 if b16 else _f32)((N, H2 // 2, W2 // 2, 9 * C), dy)
    t0 = _timer.begin() if _timer is not None else None
    fn = lib().unet_upsample2x_bwd_taps_b16 if b16 else lib().unet_upsample2x_bwd_taps
    check(fn(_ptr(dy), _ptr(D), N, H2 // 2, W2 // 2, C, _stream()))
    if t0 is not None:   # reads dy once, writes 9/4 of it
        _timer.end("upsample2x_bwd_taps", 0.0, 1, t0,
                   nbytes=dy.element_size() * dy.numel() * (1 + 9 / 4))
    return D


def conv3x3_up_bwd_weight(x, slope, D, dw_oihw, ci_offset):
    """dw[:, ci_offset : ci_offset + Cx] of conv3x3(upsample2x(act(x))) from D (low-res GEMM)."""
    x, rx = _act(x)
    N, h, w, Cx = x.shape
    Cout = D.shape[3] // 9
    assert D.shape[:3] == x.shape[:3] and dw_oihw.shape[0] == Cout and dw_oihw.is_contiguous()
    ws = _wgrad_ws(lib().unet_conv3x3_up_bwd_weight_workspace_bytes(N, h, w, Cx, Cout), D)
    fn = lib().unet_conv3x3_up_bwd_weight_b16 if _is_b16(D) else lib().unet_conv3x3_up_bwd_weight
    t0 = _timer.begin("wgrad") if _timer is not None else None
    check(fn(rx, slope, _ptr(D), _ptr(dw_oihw), ci_offset, dw_oihw.shape[1], _ptr(ws), ws.numel(),
             N, h, w, Cout, _stream()))
    if t0 is not None:   # algorithmic FLOPs: the 3x3 weight gradient on the up-sampled grid
        _timer.end("conv_wgrad_bf16" if _is_b16(D) else "conv_wgrad",
                   2.0 * N * 4 * h * w * 9 * Cx * Cout, 2, t0,
                   executed=2.0 * N * h * w * 9 * Cx * Cout)
    return dw_oihw


def conv3x3_up_bwd_data(D, wd, ci_offset, ccols, out=None, accumulate=False, nxt=None, wd3=None):
    """g[N, h, w, ccols] (+)= dL/d(low-res operand) of conv3x3(upsample2x(.)) from D.
    nxt (NextNorm): g is final for that layer - also emit its backward reductions.
    wd3 (bf16 tensors only): the data-gradient weights pre-rounded to bf16 (plain-GEMM form)."""
    N, h, w, C9 = D.shape
    Cout = C9 // 9
    cin_total = wd.shape[1]
    assert wd.shape[0] == 9 and wd.shape[2] == Cout
    b16 = _is_b16(D)
    g = out if out is not None else (_b16 if b16 else _f32)((N, h, w, ccols), D)
    fn = lib().unet_conv3x3_up_bwd_data_b16 if b16 else lib().unet_conv3x3_up_bwd_data
    t0 = _timer.begin("conv") if _timer is not None else None
    if b16 and wd3 is not None:
        bs = nxt.c_struct() if nxt is not None else None
        check(lib().unet_conv3x3_up_bwd_data_bs_b16_wb(
            _ptr(D), _ptr(wd), _ptr(wd3), cin_total, ci_offset, _ptr(g), N, h, w, Cout, ccols,
            1 if accumulate else 0, ctypes.byref(bs) if bs is not None else None, _stream()))
        if nxt is not None:
            nxt.tiles = bs.tiles_out
    elif nxt is not None:
        bs = nxt.c_struct()
        fbs = lib().unet_conv3x3_up_bwd_data_bs_b16 if b16 else lib().unet_conv3x3_up_bwd_data_bs
        check(fbs(_ptr(D), _ptr(wd), cin_total, ci_offset, _ptr(g), N, h, w, Cout, ccols,
                  1 if accumulate else 0, ctypes.byref(bs), _stream()))
        nxt.tiles = bs.tiles_out
    else:
        check(fn(_ptr(D), _ptr(wd), cin_total, ci_offset, _ptr(g), N, h, w, Cout, ccols,
                 1 if accumulate else 0, _stream()))
    if t0 is not None:   # algorithmic FLOPs: the 3x3 data gradient on the up-sampled grid
        _timer.end("conv_igemm_bf16" if b16 else "conv_igemm",
                   2.0 * N * 4 * h * w * 9 * ccols * Cout, 1, t0,
                   executed=2.0 * N * h * w * 9 * ccols * Cout)
    return g


def head1x1_in_fwd(x, slope, w, b):
    x, rx = _act(x)
    N, H, W, C = x.shape
    K = w.shape[0]
    logits = _f32((N, K, H, W), x.x)
    fn = lib().unet_head1x1_in_fwd_b16 if _is_b16(x.x) else lib().unet_head1x1_in_fwd
    t0 = _timer.begin() if _timer is not None else None
    check(fn(rx, slope, _ptr(w), _ptr(b), _ptr(logits), N, H * W, K, _stream()))
    if t0 is not None:
        _timer.end("head_fwd", 0.0, 1, t0,
                   nbytes=x.x.element_size() * x.x.numel() + 4.0 * logits.numel())
    return logits


def head1x1_in_bwd(x, slope, dlogits, w, dw, db, nxt=None):
    """nxt (NextNorm of the layer whose raw output x.x is): also leave the reductions of that
    layer's InstanceNorm backward (nxt.tiles == 0: this shape has no such epilogue)."""
    x, rx = _act(x)
    N, H, W, C = x.shape
    K = w.shape[0]
    da = torch.empty_like(x.x)
    ws = _ws(lib().unet_head1x1_bwd_workspace_bytes(N, H * W, C, K), x.x)
    b16 = _is_b16(x.x)
    t0 = _timer.begin() if _timer is not None else None
    if nxt is not None:
        bs = nxt.c_struct()
        fn = lib().unet_head1x1_in_bwd_bs_b16 if b16 else lib().unet_head1x1_in_bwd_bs
        check(fn(rx, slope, _ptr(dlogits), _ptr(w), _ptr(da), _ptr(dw), _ptr(db), _ptr(ws),
                 ws.numel(), N, H * W, K, ctypes.byref(bs), _stream()))
        nxt.tiles = bs.tiles_out
    else:
        fn = lib().unet_head1x1_in_bwd_b16 if b16 else lib().unet_head1x1_in_bwd
        check(fn(rx, slope, _ptr(dlogits), _ptr(w), _ptr(da), _ptr(dw), _ptr(db), _ptr(ws),
                 ws.numel(), N, H * W, K, _stream()))
    if t0 is not None:
        _timer.end("head_bwd", 0.0, 2, t0,
                   nbytes=da.element_size() * 2 * da.numel() + 4.0 * dlogits.numel())
    return da


# ---- bilinear 2x ---------------------------------------------------------------------
def upsample2x_fwd(x):
    N, h, w, C = x.shape
    y = _f32((N, 2 * h, 2 * w, C), x)
    check(lib().unet_upsample2x_fwd(_ptr(x), _ptr(y), N, h, w, C, _stream()))
    return y


def upsample2x_bwd(gy, out=None, accumulate=False):
    N, H2, W2, C = gy.shape
    h, w = H2 // 2, W2 // 2
    gx = out if out is not None else _f32((N, h, w, C), gy)
    check(lib().unet_upsample2x_bwd(_ptr(gy), _ptr(gx), N, h, w, C, 1 if accumulate else 0,
                                    _stream()))
    return gx


# ---- head + loss ---------------------------------------------------------------------
def head1x1_fwd(a, w, b):
    N, H, W, C = a.shape
    K = w.shape[0]
    logits = _f32((N, K, H, W), a)
    check(lib().unet_head1x1_fwd(_ptr(a), _ptr(w), _ptr(b), _ptr(logits), N, H * W, C, K,
                                 _stream()))
    return logits


def head1x1_bwd(a, dlogits, w, dw, db):
    N, H, W, C = a.shape
    K = w.shape[0]
    da = torch.empty_like(a)
    ws = _ws(lib().unet_head1x1_bwd_workspace_bytes(N, H * W, C, K), a)
    check(lib().unet_head1x1_bwd(_ptr(a), _ptr(dlogits), _ptr(w), _ptr(da), _ptr(dw), _ptr(db),
                                 _ptr(ws), ws.numel(), N, H * W, C, K, _stream()))
    return da


def dice_wce_loss_fwd_bwd(logits, target, smooth, w_dice, w_ce, ignore_index, dynamic_weights,
                          class_weights=None, grad_scale=1.0, want_grad=True, ws=None):
    """ws: a caller-kept workspace (dice_wce_loss_workspace) - needed when the gradient is taken
    later with dice_wce_loss_grad (want_grad=False here)."""
    N, K, H, W = logits.shape
    if K != 3:
        raise ValueError("the fused loss kernel handles exactly 3 classes")
    out = _f32((8,), logits)
    dl = torch.empty_like(logits) if want_grad else None
    if ws is None:
        ws = _ws(lib().unet_dice_wce_loss_workspace_bytes(N, H, W), logits)
    check(lib().unet_dice_wce_loss_fwd_bwd(_ptr(logits), _ptr(target), _ptr(out), _ptr(dl),
                                           _ptr(ws), ws.numel(), N, H, W, smooth, w_dice, w_ce,
                                           ignore_index, 1 if dynamic_weights else 0,
                                           _ptr(class_weights), grad_scale, _stream()))
    return out, dl


def dice_wce_loss_workspace(logits):
    N, K, H, W = logits.shape
    return _ws(lib().unet_dice_wce_loss_workspace_bytes(N, H, W), logits)


def dice_wce_loss_grad(logits, target, ws, upstream, ignore_index):
    """dL/dlogits of a loss whose forward ran with want_grad=False on the workspace `ws`
    (dice_wce_loss_fwd_bwd(..., ws=ws) or dice_wce_loss_shard_apply): upstream (a device float,
    dL/dloss from autograd, or None = 1) is applied inside the gradient kernel."""
    N, K, H, W = logits.shape
    dl = torch.empty_like(logits)
    if upstream is not None:
        upstream = upstream.reshape(1).float().contiguous()
    check(lib().unet_dice_wce_loss_grad(_ptr(logits), _ptr(target), _ptr(ws), ws.numel(),
                                        _ptr(upstream), _ptr(dl), N, H, W, ignore_index,
                                        _stream()))
    return dl


def dice_wce_loss_shard_stats(logits, target, smooth, ignore_index):
    """Phase 1 of the sharded-batch loss: (stats float64[10] on the device, workspace)."""
    N, K, H, W = logits.shape
    if K != 3:
        raise ValueError("the fused loss kernel handles exactly 3 classes")
    stats = torch.empty((10,), dtype=torch.float64, device=logits.device)
    ws = _ws(lib().unet_dice_wce_loss_workspace_bytes(N, H, W), logits)
    check(lib().unet_dice_wce_loss_shard_stats(_ptr(logits), _ptr(target), stats.data_ptr(),
                                               _ptr(ws), ws.numel(), N, H, W, smooth, ignore_index,
                                               _stream()))
    return stats, ws


def dice_wce_loss_shard_apply(logits, target, global_stats, n_global, ws, smooth, w_dice, w_ce,
                              ignore_index, dynamic_weights, class_weights=None, grad_scale=1.0,
                              want_grad=True):
    """Phase 2: loss of the concatenated batch (out[0]) and this shard's dL/dlogits."""
    N, K, H, W = logits.shape
    out = _f32((8,), logits)
    dl = torch.empty_like(logits) if want_grad else None
    check(lib().unet_dice_wce_loss_shard_apply(
        _ptr(logits), _ptr(target), global_stats.data_ptr(), int(n_global), _ptr(out), _ptr(dl),
        _ptr(ws), ws.numel(), N, H, W, smooth, w_dice, w_ce, ignore_index,
        1 if dynamic_weights else 0, _ptr(class_weights), grad_scale, _stream()))
    return out, dl


# ---- validation metrics / input pipeline ---------------------------------------------------
def argmax_dice_counts(logits, target, ignore_index=255, want_preds=True):
    """Returns (preds uint8 [N,H,W] or None, counts int64 [3,3] = per class
    {intersection, predicted, labelled}); everything stays on the device."""
    N, K, H, W = logits.shape
    if K != 3:
        raise ValueError("3 classes expected")
    logits, target = logits.contiguous(), target.contiguous()
    if logits.dtype != torch.float32 or target.dtype != torch.int64:
        raise TypeError("argmax_dice_counts takes fp32 logits and int64 targets")
    preds = torch.empty((N, H, W), dtype=torch.uint8, device=logits.device) if want_preds else None
    counts = torch.empty((3, 3), dtype=torch.int64, device=logits.device)
    check(lib().unet_argmax_dice_counts(_ptr(logits), _ptr(target), _ptr(preds), counts.data_ptr(),
                                        N, H, W, ignore_index, _stream()))
    return preds, counts


def argmax_classes(logits):
    """uint8 [N,H,W] class map of fp32 logits [N,3,H,W] (first maximum wins, like torch.argmax)."""
    N, K, H, W = logits.shape
    if K != 3 or logits.dtype != torch.float32:
        raise TypeError("argmax_classes takes fp32 logits with 3 classes")
    logits = logits.contiguous()
    preds = torch.empty((N, H, W), dtype=torch.uint8, device=logits.device)
    check(lib().unet_argmax_dice_counts(_ptr(logits), None, _ptr(preds), None, N, H, W, 255,
                                        _stream()))
    return preds


def preprocess_u8(image_hwc_u8, mask_u8=None, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """uint8 [N,H,W,3] (+ uint8 [N,H,W]) on the device -> (fp32 NHWC image, int64 target)."""
    import ctypes
    N, H, W, C = image_hwc_u8.shape
    if C != 3 or image_hwc_u8.dtype != torch.uint8 or not image_hwc_u8.is_contiguous():
        raise TypeError("preprocess_u8 takes a contiguous uint8 [N,H,W,3] tensor")
    out = _f32((N, H, W, 3), image_hwc_u8)
    tgt = None
    if mask_u8 is not None:
        if mask_u8.dtype != torch.uint8 or tuple(mask_u8.shape) != (N, H, W) or \
                not mask_u8.is_contiguous():
            raise TypeError("mask must be a contiguous uint8 [N,H,W] tensor")
        tgt = torch.empty((N, H, W), dtype=torch.int64, device=image_hwc_u8.device)
    m3 = (ctypes.c_float * 3)(*mean)
    s3 = (ctypes.c_float * 3)(*std)
    check(lib().unet_preprocess_u8(_ptr(image_hwc_u8), _ptr(mask_u8), _ptr(out), _ptr(tgt), N, H,
                                   W, m3, s3, _stream()))
    return out, tgt


# ---- optimizer -------------------------------------------------------------------------
def sgd_nesterov_step(params, grads, momentum, lr, mu, weight_decay, first_step, grad_scale=1.0):
    n = params.numel()
    t0 = _timer.begin() if _timer is not None else None
    check(lib().unet_sgd_nesterov_step(_ptr(params), _ptr(grads), _ptr(momentum), n, lr, mu,
                                       weight_decay, 1 if first_step else 0, grad_scale,
                                       _stream()))
    if t0 is not None:   # p, g, buf in; p, buf out
        _timer.end("sgd_nesterov", 0.0, 1, t0, nbytes=4.0 * (5 if not first_step else 4) * n)


def sgd_nesterov_step_dev(params, grads, momentum, hyper, first_step):
    """As sgd_nesterov_step with {lr, mu, weight_decay, grad_scale} read from the device tensor
    `hyper` (fp32 [4]): what a graph-captured train step launches."""
    n = params.numel()
    check(lib().unet_sgd_nesterov_step_dev(_ptr(params), _ptr(grads), _ptr(momentum), n,
                                           _ptr(hyper), 1 if first_step else 0, _stream()))


def add_inplace(a, b):
    check(lib().unet_add_inplace(_ptr(a), _ptr(b), a.numel(), _stream()))
    return a
