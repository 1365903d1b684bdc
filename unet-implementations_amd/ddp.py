"""Data-parallel training over one 8xMI355X node: one process per GPU, RCCL over xGMI.

The reference is single-process; this layer is new.  Every rank holds
a full replica and a shard of the minibatch; the only exchange is a sum all-reduce of
the flat fp32 gradient arena (19.66 M floats = 78.6 MB).  Backward completes the arena
back to front, so the all-reduce is issued in a few large buckets as suffixes become
final (decoder_stages.0 + head ~40 MB first, then encoder_stages.5, .4, then the small
high-resolution encoder stages) and overlaps the FLOP-heavy encoder backward kernels:
`torch.distributed` (backend "nccl" = RCCL) runs each bucket on its own stream.
The 1/world averaging is folded into the SGD kernel (`FusedSGD.grad_scale`).

Loss semantics (`mode`):
  "ddp"          each rank normalises its CE / class weights over its own shard, gradients
                 are averaged (standard DDP; oracle = reference per shard, mean of grads).
  "global-exact" `SimpleLoss(batch_sync="global")` + `GradBucketAllReduce(average=False)`:
                 class weights, CE denominator and the Dice batch mean are taken over the
                 concatenated batch (one extra 80-byte all-reduce inside the loss), gradients
                 are summed; N ranks x batch b then reproduce one process at batch N*b
                 (oracle = reference on the concatenated batch).
"""
import torch
import torch.distributed as dist


class GradBucketAllReduce:
    """Bucketed, overlapped all-reduce of a UNet's gradient arena."""

    def __init__(self, model, optimizer=None, process_group=None, bucket_bytes=16 << 20,
                 average=True, always_reduce=False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed must be initialised (backend nccl or gloo)")
        self.model = model
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self._works = []
        self._hi = None
        self._prefix = 0         # frozen prefix of the arena, looked up once per backward
        # (lo, hi) arena ranges handed to all_reduce during the CURRENT / most recent step,
        # newest last (reset when a new backward starts, so a long run keeps one step's worth)
        self.sent_ranges = []
        # always_reduce: issue the collectives even in a one-rank group (a rehearsal of the RCCL
        # call path - streams, async work handles - on a single GPU; the sums are identities)
        self.always_reduce = always_reduce
        model.grad_ready_hook = self._on_ready
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world if average else 1.0

    # called from UNet backward: every gradient at arena offsets >= lo is final
    def _on_ready(self, lo):
        garena = getattr(self.model, "_grad_arena", None)   # built by forward before any hook call
        if garena is None:
            _, garena = self.model.flat_parameters()
        if self._hi is None:                 # first hook call of this backward
            self._hi = garena.numel()
            self._prefix = self._frozen_prefix()
            self.sent_ranges = []
        if self._hi - lo >= self.bucket_elems or lo <= self._prefix:
            self._launch(garena, lo, self._hi)
            self._hi = lo

    def _launch(self, garena, lo, hi):
        if hi <= lo:
            return
        self.sent_ranges.append((lo, hi))
        if self.world == 1 and not self.always_reduce:
            return
        self._works.append(dist.all_reduce(garena[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                           async_op=True))

    def _frozen_prefix(self):
        """Arena offset of the first trainable parameter: a frozen encoder prefix (AE transfer,
        Our_UNet/src/train.py:800-859) produces no gradient and is left out of the exchange.
        A 90-parameter walk: called once per backward (`_on_ready` caches it in `_prefix`)."""
        if getattr(self.model, "_offsets", None) is None:
            self.model.flat_parameters()
        offsets = getattr(self.model, "_offsets", None)
        if offsets is None or not hasattr(self.model, "parameters"):
            return 0
        for p, off in zip(self.model.parameters(), offsets):
            if p.requires_grad:
                return off
        return 0

    def check_gradients_alias_arena(self):
        """The exchange ships the gradient ARENA: every trainable parameter's .grad must be a
        view into it (it is when autograd adopted the tensor UNet.backward returned).  A
        detached copy would silently train on un-reduced gradients, so fail loudly."""
        _, garena = self.model.flat_parameters()
        if getattr(self.model, "_offsets", None) is None or \
                not hasattr(self.model, "named_parameters"):
            return      # a bare arena (tests): nothing to cross-check
        gbase = garena.data_ptr()
        for (name, p), off in zip(self.model.named_parameters(), self.model._offsets):
            if p.requires_grad and p.grad is not None and p.grad.data_ptr() != gbase + 4 * off:
                raise RuntimeError(
                    f"{name}.grad does not alias the gradient arena, so the all-reduce did not "
                    "cover it (was .grad assigned by hand, or accumulated into a foreign tensor?)")

    def finish(self):
        """Call between backward and optimizer.step(): flushes the tail bucket and makes the
        current stream wait for every outstanding all-reduce."""
        self.check_gradients_alias_arena()
        if self._hi is not None and self._hi > 0 and (self.world > 1 or self.always_reduce):
            _, garena = self.model.flat_parameters()
            self._launch(garena, self._prefix, self._hi)
        for w in self._works:
            w.wait()
        self._works = []
        self._hi = None


def broadcast_parameters(model, src=0, process_group=None):
    """Make every replica start from rank `src`'s weights (one broadcast of the flat arena)."""
    arena, _ = model.flat_parameters()
    dist.broadcast(arena, src=src, group=process_group)
