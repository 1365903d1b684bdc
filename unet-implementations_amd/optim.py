"""Fused SGD with Nesterov momentum over the model's flat arenas.

Counterpart of `optim.SGD(model.parameters(), lr, momentum, nesterov=True,
weight_decay)` as configured in Our_UNet/src/train.py:445-451.  When every
parameter and gradient is a view into the `UNet` arenas the whole step is ONE
kernel launch over 19.66 M floats; otherwise each parameter gets its own launch
of the same kernel.  `state_dict()` keeps torch's SGD layout
(`state[i]['momentum_buffer']`, `param_groups`) so checkpoints interchange.
"""
import torch

from . import ops


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0,
                 nesterov=False, model=None):
        if dampening != 0.0:
            raise NotImplementedError("FusedSGD implements dampening == 0 (the reference setting)")
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        if not nesterov:
            raise NotImplementedError("FusedSGD implements the Nesterov form used by the reference")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                        nesterov=nesterov)
        super().__init__(params, defaults)
        self._model = model
        self._flat_buf = None
        self._steps = 0
        self.grad_scale = 1.0
        # graph-captured steps (train.GraphedTrainStep): lr / momentum / weight decay / grad_scale
        # are read by the kernel from this device tensor, refreshed by sync_device_hyper()
        self._hyper = None
        self._hyper_host = None

    def use_device_hyper(self, on=True):
        """Make the flat-arena step read its hyper-parameters from device memory, so that a step
        captured in a HIP graph follows later changes of `param_groups[0]['lr']` (LR schedule).
        The device tensor is created once and kept: a graph captured earlier keeps reading the
        same address when a second GraphedTrainStep is built on this optimizer."""
        if not on:
            self._hyper = self._hyper_host = None
            return
        if self._hyper is None:
            arena, _ = self._model.flat_parameters()
            self._hyper = torch.zeros(4, dtype=torch.float32, device=arena.device)
            self._hyper_host = None
        self.sync_device_hyper()

    def sync_device_hyper(self):
        """Copy {lr, momentum, weight_decay, grad_scale} to the device tensor if they changed."""
        if self._hyper is None:
            return
        g = self.param_groups[0]
        vals = (float(g["lr"]), float(g["momentum"]), float(g["weight_decay"]),
                float(self.grad_scale))
        if vals != self._hyper_host:
            self._hyper.copy_(torch.tensor(vals, dtype=torch.float32))
            self._hyper_host = vals

    def _flat_ready(self):
        """True when one launch over the arenas is equivalent to the per-parameter update."""
        m = self._model
        if m is None or len(self.param_groups) != 1:
            return False
        arena, garena = m.flat_parameters()
        params = self.param_groups[0]["params"]
        if len(params) != len(m._offsets):
            return False
        base, gbase = arena.data_ptr(), garena.data_ptr()
        for p, off in zip(params, m._offsets):
            if p.grad is None or p.data_ptr() != base + 4 * off or \
                    p.grad.data_ptr() != gbase + 4 * off:
                return False
        return True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat_buf = None      # adopt the loaded momentum buffers on the next step

    @torch.no_grad()
    def adopt_flat_momentum(self):
        """Build the flat momentum arena (one buffer aliased by every state[p]['momentum_buffer'])
        without taking a step.  Buffers that already exist - a checkpoint loaded through
        load_state_dict - are copied in.  Returns True when none existed (zero momentum: the next
        step is torch SGD's "first step", buf <- g).  No-op (returns None) once the arena exists."""
        if self._flat_buf is not None:
            return None
        m = self._model
        g = self.param_groups[0]
        arena, _ = m.flat_parameters()
        if len(self.param_groups) != 1 or len(g["params"]) != len(m._offsets):
            raise RuntimeError("the flat momentum arena needs one parameter group holding every "
                               "parameter of the model")
        self._flat_buf = torch.zeros_like(arena)
        fresh = True
        for p, off in zip(g["params"], m._offsets):
            st = self.state[p]
            if "momentum_buffer" in st and st["momentum_buffer"] is not None:
                # resumed from a checkpoint: adopt the loaded buffers
                self._flat_buf[off:off + p.numel()].view_as(p).copy_(st["momentum_buffer"])
                fresh = False
            st["momentum_buffer"] = self._flat_buf[off:off + p.numel()].view_as(p)
        return fresh

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flat = self._flat_ready()
        if not flat and self.grad_scale != 1.0 and self._model is not None:
            # grad_scale is set by the data-parallel wrapper, which all-reduces the ARENA: a
            # gradient that left the arena was not reduced
            m = self._model
            _, garena = m.flat_parameters()
            for (name, p), off in zip(m.named_parameters(), m._offsets):
                if p.grad is not None and p.grad.data_ptr() != garena.data_ptr() + 4 * off:
                    raise RuntimeError(f"{name}.grad does not alias the gradient arena the "
                                       "all-reduce operates on")
        if flat:
            g = self.param_groups[0]
            arena, garena = self._model.flat_parameters()
            first = bool(self.adopt_flat_momentum())
            if self._hyper is not None:
                # (an eager step after a GraphedTrainStep was built - a last batch of another
                # shape, a data-parallel path that sets grad_scale - must not run on stale
                # values; no-op when nothing changed, so also inside a capture)
                self.sync_device_hyper()
                ops.sgd_nesterov_step_dev(arena, garena, self._flat_buf, self._hyper, first)
            else:
                ops.sgd_nesterov_step(arena, garena, self._flat_buf, g["lr"], g["momentum"],
                                      g["weight_decay"], first, self.grad_scale)
        else:
            if self._hyper is not None:
                raise RuntimeError("device-side hyper-parameters need the flat-arena step (every "
                                   "parameter and gradient a view of the UNet arenas)")
            for g in self.param_groups:
                for p in g["params"]:
                    if p.grad is None:
                        continue
                    st = self.state[p]
                    first = "momentum_buffer" not in st or st["momentum_buffer"] is None
                    if first:
                        st["momentum_buffer"] = torch.zeros_like(p)
                    if p.data_ptr() % 16 or p.grad.data_ptr() % 16 or not p.is_contiguous() \
                            or not p.grad.is_contiguous():
                        raise RuntimeError("FusedSGD needs contiguous, 16-byte aligned tensors")
                    ops.sgd_nesterov_step(p.data.view(-1), p.grad.view(-1),
                                          st["momentum_buffer"].view(-1), g["lr"], g["momentum"],
                                          g["weight_decay"], first, self.grad_scale)
        self._steps += 1
        return loss
