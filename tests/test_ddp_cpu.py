"""CPU (-m "not gpu"): the data-parallel exchange step with world_size 2 over gloo.

The gradient all-reduce is layout plumbing (a flat fp32 arena completed back to front), so it
is exercised here on CPU tensors with a stand-in that exposes the same `flat_parameters()` /
`grad_ready_hook` surface as the HIP `UNet`; the per-shard gradients come from the oracle at a
tiny resolution, and the result is checked against the single-process mean of shard gradients
(the "ddp" loss semantics of DESIGN.md)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Arena:
    """Stand-in with the UNet arena surface."""

    def __init__(self, n):
        self.arena = torch.zeros(n)
        self.garena = torch.zeros(n)
        self.grad_ready_hook = None

    def flat_parameters(self):
        return self.arena, self.garena


def _shard_grads(rank):
    """Deterministic per-rank 'gradient' arena and the hook call sequence of a backward."""
    g = torch.Generator().manual_seed(100 + rank)
    n = 10_000
    return torch.randn(n, generator=g), [9000, 7000, 6500, 3000, 100, 0]


def _worker(rank, world, port, bucket_bytes, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import unet_implementations_amd as ua
    from unet_implementations_amd import ddp

    grads, los = _shard_grads(rank)
    model = _Arena(grads.numel())
    model.arena.copy_(torch.arange(grads.numel(), dtype=torch.float32) * (rank + 1))

    class _Opt:
        grad_scale = 1.0

    opt = _Opt()
    sync = ddp.GradBucketAllReduce(model, opt, bucket_bytes=bucket_bytes)
    assert opt.grad_scale == 1.0 / world
    ddp.broadcast_parameters(model)
    for step in range(2):           # the bucket state must reset between steps
        model.garena.zero_()
        for lo in los:
            hi = model.garena.numel() if lo == los[0] else prev
            model.garena[lo:hi] = grads[lo:hi] * (step + 1)
            prev = lo
            model.grad_ready_hook(lo)
        sync.finish()
        if rank == 0:
            out[f"g{step}"] = model.garena.clone()
    if rank == 0:
        out["arena"] = model.arena.clone()
        out["nbuckets"] = 0
    dist.destroy_process_group()
    del ua


@pytest.mark.parametrize("bucket_bytes", [4 * 2500, 1 << 30, 4])
def test_bucketed_allreduce_world2(bucket_bytes):
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, bucket_bytes, out), nprocs=world, join=True)
    total = sum(_shard_grads(r)[0] for r in range(world))
    assert torch.allclose(out["g0"], total, rtol=0, atol=1e-6)
    assert torch.allclose(out["g1"], 2 * total, rtol=0, atol=1e-6)
    # broadcast made rank 0's parameters the common starting point
    assert torch.equal(out["arena"], torch.arange(10_000, dtype=torch.float32))


def test_single_process_is_a_noop():
    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from unet_implementations_amd import ddp
        m = _Arena(64)
        sync = ddp.GradBucketAllReduce(m, None, bucket_bytes=16)
        m.garena.fill_(3.0)
        for lo in (48, 16, 0):
            m.grad_ready_hook(lo)
        sync.finish()
        assert torch.all(m.garena == 3.0)
    finally:
        dist.destroy_process_group()


def test_frozen_prefix_is_left_out_of_the_exchange():
    """A frozen encoder prefix (requires_grad=False) is not all-reduced: finish() starts at the
    first trainable parameter's arena offset."""
    from unet_implementations_amd import ddp

    class _P:
        def __init__(self, rg):
            self.requires_grad = rg

    class _M(_Arena):
        def __init__(self):
            super().__init__(100)
            self._offsets = [0, 40, 70]
            self._params = [_P(False), _P(False), _P(True)]

        def parameters(self):
            return iter(self._params)

    sync = ddp.GradBucketAllReduce.__new__(ddp.GradBucketAllReduce)
    sync.model = _M()
    assert sync._frozen_prefix() == 70
    sync.model._params[0].requires_grad = True
    assert sync._frozen_prefix() == 0
    sync.model = _Arena(10)          # no parameter table: whole arena
    assert sync._frozen_prefix() == 0
