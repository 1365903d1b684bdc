"""GPU (-m gpu): the data-parallel train step with two ranks sharing cuda:0 over gloo
(RCCL refuses two ranks on one device; the exchange code path - hooks, buckets, finish,
1/world scaling in the SGD kernel - is backend independent).

Checks: (a) both replicas end bit-identical, (b) the all-reduced gradient equals the sum of
the two shard gradients obtained without any exchange ("ddp" loss semantics: each rank
normalises its loss over its own shard, gradients are averaged)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HW, N_PER_RANK = 64, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(rank):
    import unet_implementations_amd as ua
    from oracle import unet_ref as O
    sd0 = O.fill_state_dict(31)
    model = ua.UNet()
    model.load_state_dict(sd0)
    model = model.to("cuda").train()
    img, tgt = O.synthetic_batch(500 + rank, N_PER_RANK, HW, HW)
    model.dropout_mask_override = O.draw_dropout_masks(900 + rank, N_PER_RANK)
    return ua, model, img.cuda(), tgt.cuda()


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_implementations_amd import ddp
    ua, model, img, tgt = _setup(rank)
    opt = ua.create_optimizer(model)
    sync = ddp.GradBucketAllReduce(model, opt, bucket_bytes=8 << 20)
    ddp.broadcast_parameters(model)
    loss = ua.train_step(model, opt, ua.get_loss_function(), img, tgt, grad_sync=sync.finish)
    torch.cuda.synchronize()
    arena, garena = model.flat_parameters()
    out[f"grad{rank}"] = garena.cpu()
    out[f"param{rank}"] = arena.cpu()
    out[f"loss{rank}"] = loss.item()
    dist.destroy_process_group()


def test_two_rank_step_matches_shard_sum():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert torch.equal(out["param0"], out["param1"]), "replicas diverged"
    assert torch.equal(out["grad0"], out["grad1"])
    # shard gradients without exchange, same kernels, same masks
    total = None
    for rank in range(world):
        ua, model, img, tgt = _setup(rank)
        loss = ua.get_loss_function()(model(img), tgt)
        loss.backward()
        assert abs(loss.item() - out[f"loss{rank}"]) <= 1e-6 * abs(loss.item())
        _, g = model.flat_parameters()
        total = g.cpu().clone() if total is None else total + g.cpu()
    ref = total
    err = ((out["grad0"] - ref).norm() / ref.norm()).item()
    assert err <= 1e-6, f"all-reduced gradient differs from the shard sum: {err:.3e}"
    # the SGD kernel applied grad/world: first step p1 = p0 - lr*(1+mu)*(g/2 + wd*p0)
    from oracle import unet_ref as O
    sd0 = O.fill_state_dict(31)
    k = "decoder_stages.0.conv_block.block.0.weight"
    ua, model, _, _ = _setup(0)
    model.flat_parameters()
    idx = [n for n, _ in model.named_parameters()].index(k)
    off = model._offsets[idx]
    n = sd0[k].numel()
    g = ref[off:off + n].view_as(sd0[k]) / world + 1e-4 * sd0[k]
    expect = sd0[k] - 0.005 * (1 + 0.99) * g
    got = out["param0"][off:off + n].view_as(sd0[k])
    assert ((got - expect).abs().max() / expect.abs().max()).item() <= 1e-6


def _worker_global(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unet_implementations_amd import ddp
    ua, model, img, tgt = _setup(rank)
    opt = ua.create_optimizer(model)
    sync = ddp.GradBucketAllReduce(model, opt, bucket_bytes=8 << 20, average=False)
    ddp.broadcast_parameters(model)
    loss_fn = ua.SimpleLoss(batch_sync="global")
    loss = ua.train_step(model, opt, loss_fn, img, tgt, grad_sync=sync.finish)
    torch.cuda.synchronize()
    arena, garena = model.flat_parameters()
    out[f"grad{rank}"] = garena.cpu()
    out[f"param{rank}"] = arena.cpu()
    out[f"loss{rank}"] = loss.item()
    dist.destroy_process_group()


def test_two_rank_global_exact_equals_one_process_on_concatenated_batch():
    """batch_sync="global": 2 ranks x batch 2 == 1 process x batch 4 (loss, gradient, step)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_global, args=(world, _free_port(), out), nprocs=world, join=True)
    assert out["loss0"] == out["loss1"]
    assert torch.equal(out["param0"], out["param1"])
    import unet_implementations_amd as ua
    from oracle import unet_ref as O
    parts = [_setup(r) for r in range(world)]
    model = parts[0][1]
    img = torch.cat([p[2] for p in parts])
    tgt = torch.cat([p[3] for p in parts])
    m0, m1 = parts[0][1].dropout_mask_override, parts[1][1].dropout_mask_override
    model.dropout_mask_override = [torch.cat([a, b]) for a, b in zip(m0, m1)]
    opt = ua.create_optimizer(model)
    loss = ua.train_step(model, opt, ua.get_loss_function(), img, tgt)
    arena, garena = model.flat_parameters()
    assert abs(loss.item() - out["loss0"]) <= 2e-6 * abs(loss.item())
    g, ref = out["grad0"], garena.cpu()
    assert ((g - ref).norm() / ref.norm()).item() <= 1e-5
    p, pref = out["param0"], arena.cpu()
    assert ((p - pref).abs().max() / pref.abs().max()).item() <= 1e-6
    # and the loss is the reference's loss of the 4-image batch
    sd0 = O.fill_state_dict(31)
    masks = [torch.cat([a, b]) for a, b in zip(m0, m1)]
    with torch.no_grad():
        logits = O.unet_forward(sd0, img.cpu(), masks)
        ref_loss = O.simple_loss(logits, tgt.cpu()).item()
    assert abs(out["loss0"] - ref_loss) <= 2e-5 * abs(ref_loss)


def test_bench_two_ranks_through_torch_distributed_run():
    """The driver's multi-GPU command line (`python -m torch.distributed.run --nproc-per-node N
    bench.py --gpus N`) rehearsed with 2 ranks sharing cuda:0 over gloo: the JSON line must carry
    the contract fields, the world size and the backend that ran."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UNET_SHARE_GPU="1", UNET_DIST_BACKEND="gloo",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--hw", "64", "--no-alt", "--no-cpu-baseline"]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1
    assert r["scaling"] == "weak" and r["higher_is_better"] is True and r["value"] > 0
    assert r["config"]["global_batch"] == 4 and r["config"]["parallelism"] == "dp2"
    assert r["config"]["world"] == 2 and r["config"]["dist_backend"] == "gloo"
    assert abs(r["value"] - 4 * 2 / (r["ms_per_step"] * 2e-3)) < 1e-6 * r["value"]


def _run_bench(args, extra_env):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py")] + args
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_2_without_a_launcher():
    """`python bench.py --gpus 2` exactly as a driver might type it (no torch.distributed.run):
    bench.py starts its own two ranks as child processes and relays rank 0's JSON line."""
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--hw", "64",
                    "--no-alt", "--no-cpu-baseline"],
                   dict(UNET_SHARE_GPU="1", UNET_DIST_BACKEND="gloo"))
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1
    assert r["config"]["world"] == 2 and r["config"]["dist_backend"] == "gloo"
    assert r["config"]["global_batch"] == 4 and r["value"] > 0


def test_two_ranks_at_full_size_share_one_gpu():
    """The benchmark configuration itself (bs 8, 512 x 512: the Winograd kernels and every
    hand-scheduled loop run) on two ranks time-slicing cuda:0.  Memory latencies under sharing are
    several times the exclusive ones, which is what it takes to expose a register read or reused
    before a hand-waited load has landed (one did: "Memory access fault by GPU", found by exactly
    this rehearsal; tools/asm_hazard_check.py is the static side of the same check)."""
    r = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--no-alt", "--no-cpu-baseline",
                    "--no-graph"], dict(UNET_SHARE_GPU="1", UNET_DIST_BACKEND="gloo"))
    assert r["n_gpus"] == 2 and r["config"]["world"] == 2 and r["config"]["global_batch"] == 16
    assert r["value"] > 0


def test_bench_single_gpu_through_a_one_rank_rccl_group():
    """UNET_BENCH_RCCL=1: the N=1 bench runs its fence barrier, the MAX all-reduce of the step
    time and the bucketed gradient all-reduces through RCCL (backend nccl, one rank)."""
    r = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "2", "--hw", "64",
                    "--no-alt", "--no-cpu-baseline"], dict(UNET_BENCH_RCCL="1"))
    assert r["n_gpus"] == 1 and r["config"]["world"] == 1
    assert r["config"]["dist_backend"] == "nccl (RCCL)" and r["value"] > 0
    assert r["roofline"]["frac"] > 0 and "enqueue_ms_per_step" in r


def _worker_rccl(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from unet_implementations_amd import ddp
    ua, model, img, tgt = _setup(0)
    opt = ua.create_optimizer(model)
    sync = ddp.GradBucketAllReduce(model, opt, bucket_bytes=8 << 20, always_reduce=True)
    ddp.broadcast_parameters(model)
    loss = ua.train_step(model, opt, ua.get_loss_function(), img, tgt, grad_sync=sync.finish)
    torch.cuda.synchronize()
    arena, garena = model.flat_parameters()
    out["grad"] = garena.cpu()
    out["param"] = arena.cpu()
    out["loss"] = loss.item()
    out["ranges"] = list(sync.sent_ranges)
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_call_path_with_a_single_rank():
    """Backend "nccl" (= RCCL) on the one GPU this box has: a one-rank group whose bucketed
    all-reduces, broadcast and barrier are really issued (`always_reduce`).  The sums are
    identities, so the step must equal the plain single-process step bit for bit - what this
    covers is the RCCL initialisation with `device_id`, the async work handles on RCCL's stream
    and the wait in finish(), which the gloo tests cannot."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_rccl, args=(1, _free_port(), out), nprocs=1, join=True)
    ua, model, img, tgt = _setup(0)
    opt = ua.create_optimizer(model)
    loss = ua.train_step(model, opt, ua.get_loss_function(), img, tgt)
    arena, garena = model.flat_parameters()
    assert out["loss"] == loss.item()
    assert torch.equal(out["grad"], garena.cpu())
    assert torch.equal(out["param"], arena.cpu())
    covered = sorted(out["ranges"])
    assert covered[0][0] == 0 and covered[-1][1] == garena.numel()
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), "buckets must tile the arena"


def _worker_rccl_graph(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    from unet_implementations_amd import ddp
    from oracle import unet_ref as O
    for graphed in (False, True):
        ua, model, img, tgt = _setup(0)
        img2, tgt2 = O.synthetic_batch(777, N_PER_RANK, HW, HW)
        img2, tgt2 = img2.cuda(), tgt2.cuda()
        model.dropout_mask_override = [m.cuda() if m is not None else None
                                       for m in model.dropout_mask_override]
        opt = ua.create_optimizer(model)
        lossf = ua.get_loss_function()
        sync = ddp.GradBucketAllReduce(model, opt, bucket_bytes=8 << 20, always_reduce=True)
        ddp.broadcast_parameters(model)
        if graphed:
            gstep = ua.GraphedTrainStep(model, opt, lossf, img, tgt, grad_sync=sync)
            out["capture_ranges"] = list(sync.sent_ranges)
            step = gstep
        else:
            def step(x, y, model=model, opt=opt, lossf=lossf, sync=sync):
                return ua.train_step(model, opt, lossf, x, y, grad_sync=sync.finish)
        losses = []
        for k, batch in enumerate(((img, tgt), (img2, tgt2), (img, tgt), (img2, tgt2))):
            if k == 2:
                opt.param_groups[0]["lr"] *= 0.5
            losses.append(step(*batch).item())
        torch.cuda.synchronize()
        arena, garena = model.flat_parameters()
        tag = "graph" if graphed else "eager"
        out[tag] = (losses, arena.cpu(), garena.cpu(), opt._flat_buf.cpu())
        del model, opt
    dist.barrier()
    dist.destroy_process_group()


def test_graph_captured_data_parallel_step_equals_the_eager_one():
    """GraphedTrainStep(..., grad_sync=GradBucketAllReduce) with backend nccl (= RCCL, one rank
    with `always_reduce`, so every bucketed all-reduce is really issued): the collectives are
    captured with the step (forked onto RCCL's stream where a bucket becomes final, joined before
    the SGD launch).  Four replayed steps - the last two after a learning-rate change - must
    leave losses, parameters, the last gradient arena and the momentum bit-identical to four
    eager data-parallel steps, and the captured step must have shipped the whole arena."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_rccl_graph, args=(1, _free_port(), out), nprocs=1, join=True)
    eager, graph = out["eager"], out["graph"]
    assert eager[0] == graph[0], f"losses differ: {eager[0]} vs {graph[0]}"
    assert torch.equal(eager[1], graph[1]), "parameters differ after four steps"
    assert torch.equal(eager[2], graph[2]), "gradient arenas differ"
    assert torch.equal(eager[3], graph[3]), "momentum differs"
    covered = sorted(out["capture_ranges"])
    assert covered[0][0] == 0 and covered[-1][1] == eager[2].numel()
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:])), "buckets must tile the arena"


def test_graphed_step_refuses_a_host_side_backend():
    """gloo runs its collectives on the host: capturing them is impossible, so GraphedTrainStep
    must refuse instead of silently replaying a step without the exchange."""
    port = _free_port()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from unet_implementations_amd import ddp
        ua, model, img, tgt = _setup(0)
        opt = ua.create_optimizer(model)
        sync = ddp.GradBucketAllReduce(model, opt)
        with pytest.raises(RuntimeError, match="cannot be captured"):
            ua.GraphedTrainStep(model, opt, ua.get_loss_function(), img, tgt, grad_sync=sync)
        with pytest.raises(RuntimeError, match="pass its GradBucketAllReduce"):
            ua.GraphedTrainStep(model, opt, ua.get_loss_function(), img, tgt)
    finally:
        dist.destroy_process_group()
