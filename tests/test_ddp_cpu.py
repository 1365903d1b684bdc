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


class _P:
    def __init__(self, rg, grad=None):
        self.requires_grad = rg
        self.grad = grad


class _M(_Arena):
    """Arena stand-in with a parameter table: three parameters at offsets 0 / 40 / 70."""

    def __init__(self, frozen=(True, True, False)):
        super().__init__(100)
        self._offsets = [0, 40, 70]
        self._params = [_P(not f) for f in frozen]

    def parameters(self):
        return iter(self._params)

    def named_parameters(self):
        return iter([(f"p{i}", p) for i, p in enumerate(self._params)])


def test_frozen_prefix_is_left_out_of_the_exchange():
    """A frozen encoder prefix (requires_grad=False) is not all-reduced: the ranges actually
    handed to all_reduce start at the first trainable parameter's arena offset."""
    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from unet_implementations_amd import ddp
        m = _M()
        sync = ddp.GradBucketAllReduce(m, None, bucket_bytes=4 * 20)
        assert sync._frozen_prefix() == 70
        # UNet.backward reports suffixes, the last call carries the frozen-prefix offset
        for lo in (90, 70):
            m.grad_ready_hook(lo)
        sync.finish()
        assert sync.sent_ranges == [(70, 100)] or sync.sent_ranges == [(90, 100), (70, 90)]
        assert min(lo for lo, _ in sync.sent_ranges) == 70
        # nothing frozen: the whole arena is covered exactly once, back to front
        m2 = _M(frozen=(False, False, False))
        sync2 = ddp.GradBucketAllReduce(m2, None, bucket_bytes=4 * 20)
        for lo in (90, 70, 40, 0):
            m2.grad_ready_hook(lo)
        sync2.finish()
        covered = sorted(sync2.sent_ranges)
        assert covered[0][0] == 0 and covered[-1][1] == 100
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
        sync.model = _Arena(10)          # no parameter table: whole arena
        assert sync._frozen_prefix() == 0
    finally:
        dist.destroy_process_group()


def test_finish_refuses_gradients_outside_the_arena():
    """If a parameter's .grad is not a view of the arena the all-reduce did not cover it:
    finish() must raise instead of training on un-reduced gradients."""
    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from unet_implementations_amd import ddp
        m = _M(frozen=(False, False, False))
        for p, off, n in zip(m._params, m._offsets, (40, 30, 30)):
            p.grad = m.garena[off:off + n]
        sync = ddp.GradBucketAllReduce(m, None)
        m.grad_ready_hook(0)
        sync.finish()                                   # all views: fine
        m._params[1].grad = m.garena[40:70].clone()     # a detached copy
        m.grad_ready_hook(0)
        with pytest.raises(RuntimeError, match="does not alias the gradient arena"):
            sync.finish()
    finally:
        dist.destroy_process_group()


def test_sent_ranges_keep_one_step_only():
    """`sent_ranges` serves the tests; it must not grow with the number of steps."""
    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from unet_implementations_amd import ddp
        m = _M(frozen=(False, False, False))
        sync = ddp.GradBucketAllReduce(m, None, bucket_bytes=4 * 20)
        for _ in range(5):
            for lo in (90, 70, 40, 0):
                m.grad_ready_hook(lo)
            sync.finish()
            covered = sorted(sync.sent_ranges)
            assert covered[0][0] == 0 and covered[-1][1] == 100
            assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
            assert len(sync.sent_ranges) <= 4
    finally:
        dist.destroy_process_group()


def test_bench_self_launch_builds_the_launcher_command(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE starts its own N ranks through
    torch.distributed.run as a CHILD process (the parent never touches the GPU)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    seen = {}

    class _R:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return _R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
