#!/usr/bin/env python3
"""Throughput of the Our_UNet train step on MI355X (BASELINE.json metric).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

One step = zero_grad -> UNet.forward -> SimpleLoss -> backward -> SGD-Nesterov step on one
synthetic batch of 8 images 512x512 per GPU (inputs resident in HBM), train mode (dropout on).
Rank 0 prints ONE JSON line; `value` is images/s over all GPUs (weak scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_IMAGE = 385.188   # fwd + dgrad + wgrad, convolutions only
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (spec)
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E
CONV_GROUPS = ("conv_igemm", "conv_igemm_bf16", "conv_igemm_bf16x3")


def hbm_traffic(args, matmul):
    """HBM bytes per launch of the forward / data-gradient convolution group from the PMC passes
    committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of
    this command, FETCH_SIZE doubled as the MI355X guide prescribes for gfx950; tools/
    pmc_passes.sh, tools/pmc_traffic.py).  A committed measurement of the default workload (and
    of `--matmul bf16`), not of this run."""
    if args.hw != 512 or args.batch != 8 or args.clip:
        return None, None
    names = {"fp32": ("r04_conv_hbm_traffic.json", "r03_conv_hbm_traffic.json",
                      "r02_conv_hbm_traffic.json", "r01_igemm_hbm_traffic.json"),
             "bf16": ("r04_bf16_conv_hbm_traffic.json",)}.get(matmul, ())
    for name in names:
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                return json.load(f)["hbm_bytes_per_launch"], "profiles/" + name
    return None, None


def roofline_of(summ, steps, matmul, args, gsumm=None, gsteps=None):
    """`roofline` object of the dominant kernel group (all 3x3 / 1x1 convolution forward and
    data-gradient launches), from the KernelTimer summary of the TIMED region (`summ`, which
    brackets only that group: an event pair per call costs stream time), and the per-group
    table from the summary of a separate pass with every entry point bracketed (`gsumm`)."""
    pref = {"fp32": "conv_igemm", "bf16": "conv_igemm_bf16", "bf16x3": "conv_igemm_bf16x3"}[matmul]
    k = summ.get(pref) or next((summ[t] for t in CONV_GROUPS if t in summ), None)
    if gsumm is None:
        gsumm, gsteps = summ, steps
    groups = {}
    for t, v in gsumm.items():
        sec = v["ms"] * 1e-3
        g = {"ms_per_step": v["ms"] / gsteps, "launches_per_step": v["launches"] / gsteps}
        if v["flops"]:
            g["tflops"] = v["flops"] / sec * 1e-12
            if v["executed"] != v["flops"]:
                g["executed_tflops"] = v["executed"] / sec * 1e-12
        if v["bytes"]:   # streaming kernels: algorithmic bytes against the HBM roofline
            g["hbm_gbs"] = v["bytes"] / sec * 1e-9
            g["frac_of_hbm_peak"] = g["hbm_gbs"] / PEAK_HBM_GBS
        groups[t] = g
    if k is None:
        return None, groups
    total = sum(v["flops"] for v in gsumm.values()) * steps / gsteps
    sec = k["ms"] * 1e-3
    ach = k["flops"] / sec * 1e-12
    # bf16x3 issues 6 bf16 MFMA flops per algorithmic flop
    peak = {"fp32": PEAK_F32_MFMA_TFLOPS, "bf16": PEAK_BF16_MFMA_TFLOPS,
            "bf16x3": PEAK_BF16_MFMA_TFLOPS / 6.0}[matmul]
    traffic, src = hbm_traffic(args, matmul)
    exe = k["executed"] / sec * 1e-12
    roof = {
        # `achieved` / `frac`: the MFMA FLOPs the kernels actually ISSUE over their measured time
        # (matrix-pipe utilisation against the dense peak).  Two reassociations issue fewer
        # FLOPs than the reference layer's 3x3 convolution: the data gradient of the up-sampled
        # operand runs on the low-resolution grid (1/4) and the Winograd kernels multiply 16
        # instead of 36 times per 2x2 output tile (4/9).  Crediting the reference's (algorithmic)
        # count gives `algorithmic` / `algorithmic_frac` - the figure to compare with a direct
        # convolution, whose ceiling is the peak itself (`algorithmic_frac` > 1 is possible).
        "bound": "mfma", "achieved": exe, "peak": peak, "unit": "TFLOP/s", "frac": exe / peak,
        "algorithmic": ach, "algorithmic_frac": ach / peak,
        "traffic": traffic, "traffic_source": src,
        # HBM rate of the same launches: measured bytes per launch / measured launch time
        "traffic_gbs": (traffic / (k["ms"] * 1e-3 / k["launches"]) * 1e-9) if traffic else None,
        "traffic_frac_of_hbm_peak": (traffic / (k["ms"] * 1e-3 / k["launches"]) * 1e-9 / PEAK_HBM_GBS)
        if traffic else None,
        "kernel": "convolution forward + data-gradient group ("
                  + {"fp32": "conv_wino_kernel, conv_wino32q_kernel, conv_wino_up32_kernel (Winograd F(2x2,3x3): 16/36 of the "
                             "direct MFMA FLOPs), conv_patch_f32_kernel, conv_patch_up_kernel, "
                             "conv_patch_s2_kernel, conv_igemm_kernel, conv_dgrad_s2_patch_kernel",
                     "bf16": "conv_patch_b16_kernel, conv_igemm_bf16_kernel on bf16 tensors",
                     "bf16x3": "conv_patch_split_kernel on the fused pipeline + the fp32 kernels "
                               "for the shapes it does not tile (stride 2, 1/32 resolution)"}[matmul] + "): "
                  f"{k['flops'] / steps * 1e-12:.3f} of the step's "
                  f"{total / steps * 1e-12:.3f} algorithmic conv TFLOP",
        "launches_per_step": k["launches"] / steps,
        "avg_launch_us": 1e3 * k["ms"] / k["launches"],
        "flop_per_step": k["executed"] / steps,
        "flop_per_launch": k["executed"] / k["launches"],
        "algorithmic_flop_per_launch": k["flops"] / k["launches"],
    }
    return roof, groups


def synthetic_batch(seed, n, h, w):
    """The bench workload: images ~ N(0,1) (ImageNet-standardised pixels), int64
    masks with background 0, one blob of class 1 or 2 per image and a ring of 255 around it."""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = torch.from_numpy(rng.standard_normal((n, 3, h, w)).astype(np.float32))
    yy, xx = np.mgrid[0:h, 0:w]
    mask = np.zeros((n, h, w), dtype=np.int64)
    for i in range(n):
        cy, cx = h * (0.4 + 0.2 * rng.random()), w * (0.4 + 0.2 * rng.random())
        ry, rx = h * (0.22 + 0.1 * rng.random()), w * (0.25 + 0.1 * rng.random())
        d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
        mask[i][d < 1.0] = 1 + (i % 2)
        mask[i][(d >= 1.0) & (d < 1.0 + 8.0 / min(ry, rx))] = 255
    return img, torch.from_numpy(mask)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(steps=8, hw=512, n=2):
    """The oracle's train step (stock torch CPU ops, verified equal to the reference) timed on
    this box's host cores: the reported CPU baseline, never the thing shipped."""
    from oracle import unet_ref as O
    # a 1-GPU box owns a 16-core share of the host (oversubscribing all 256 hardware threads
    # makes torch's CPU kernels ~40x slower); use the affinity mask, capped at 16 threads
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    sd = O.leaf_state_dict(O.fill_state_dict(1, trained_like=False))
    bufs = [None] * len(sd)
    img, tgt = O.synthetic_batch(1234, n, hw, hw)
    masks = O.draw_dropout_masks(0, n)
    O.train_step(sd, bufs, img, tgt, masks)  # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(sd, bufs, img, tgt, masks)
    dt = time.perf_counter() - t0
    used = torch.get_num_threads()
    # one thread: one step on ONE image (a bs-2 step would take ~20 s)
    torch.set_num_threads(1)
    m1 = [m[:1] for m in masks]
    t1 = time.perf_counter()
    O.train_step(sd, bufs, img[:1], tgt[:1], m1)
    dt1 = time.perf_counter() - t1
    torch.set_num_threads(used)
    return {"value": n * steps / dt, "unit": "images/s", "cores": used,
            "kind": "port", "cpu": cpu_model(),
            "one_thread": {"value": 1.0 / dt1, "unit": "images/s",
                           "sample": f"1 train step of the oracle at bs=1, {hw}x{hw}, 1 thread"},
            "sample": f"{steps} train steps of the oracle (torch CPU fp32) at bs={n}, {hw}x{hw}, "
                      f"after 1 warm-up step"}


class stdout_to_stderr:
    """fd 1 -> fd 2 for the duration of the block.  RCCL prints a version banner on the C-level
    stdout when its communicator comes up; the contract of this script is ONE JSON line there."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def self_launch(n):
    """Run this command line under `python -m torch.distributed.run --nproc-per-node n` as a
    child process (never exec: the parent stays a plain launcher that has not touched the GPU)."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--hw", type=int, default=512)
    ap.add_argument("--matmul", choices=["fp32", "bf16", "bf16x3"], default="fp32",
                    help="fp32 = fp32 matrix cores; bf16 = BASELINE config 4 (bf16 layer tensors in "
                         "HBM and bf16 MFMA operands; fp32 accumulate, statistics, master weights "
                         "and optimizer); bf16x3 = fp32 operands split into three bf16 terms, six "
                         "products per multiply on the bf16 matrix cores (fp32-class accuracy, see "
                         "DESIGN.md)")
    ap.add_argument("--clip", action="store_true",
                    help="BASELINE config 5: CLIP_UNet variant with synthetic CLIP features")
    ap.add_argument("--loss-sync", choices=["local", "global"], default="local",
                    help="N>1: per-shard loss + averaged gradients (default) or the loss of the "
                         "concatenated batch + summed gradients")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the second timed loop that reports the bf16x3 operand mode beside "
                         "the fp32 headline (same model, same K steps)")
    ap.add_argument("--no-graph", action="store_true",
                    help="skip the extra leg that replays the step from a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process
        # per GPU through torch.distributed.run) BEFORE this process touches the GPU, relay rank
        # 0's JSON line and exit with the launcher's code
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("UNET_SHARE_GPU"):   # rehearsal: every rank on device 0
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's "
                         "--nproc-per-node must equal --gpus")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import torch.distributed as dist
    import unet_implementations_amd as ua
    from unet_implementations_amd import ddp

    backend_used = None
    # UNET_BENCH_RCCL=1 at N=1: a one-rank RCCL group, so the fence's barrier, the MAX all-reduce
    # of the step time and the bucketed gradient all-reduces (always_reduce) really go through
    # RCCL in the bench itself (a rehearsal of the N>1 call path on the one GPU of a test box)
    solo_rccl = world == 1 and bool(os.environ.get("UNET_BENCH_RCCL"))
    use_dist = world > 1 or solo_rccl
    if use_dist:
        # backend "nccl" is RCCL on ROCm; UNET_DIST_BACKEND=gloo lets the N>1 path be rehearsed
        # with several ranks sharing one GPU (RCCL refuses duplicate devices)
        backend = os.environ.get("UNET_DIST_BACKEND", "nccl")
        backend_used = backend if backend != "nccl" else "nccl (RCCL)"
        if solo_rccl:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                sock = socket.socket()
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
                sock.close()
        kw = dict(rank=rank, world_size=world) if solo_rccl else {}
        with stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev, **kw)
            else:
                dist.init_process_group(backend=backend, **kw)
            dist.barrier()          # brings the communicator (and its banner) up here

    # the CPU baseline (the oracle on the host cores) runs BEFORE the GPU legs: the GPU phases
    # then form one contiguous stretch that a coarse utilisation sampler can see
    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline()

    torch.manual_seed(1234)          # same initial replica on every rank
    if args.clip:
        model = ua.CLIPUNet(with_clip_features=True, clip_dim=512).to(dev).train()
    else:
        model = ua.create_model(dev).train()
    model.matmul_precision = args.matmul
    opt = ua.create_optimizer(model)
    lossf = ua.get_loss_function()
    lossf.batch_sync = args.loss_sync
    sync = None
    if use_dist:
        ddp.broadcast_parameters(model)
        sync = ddp.GradBucketAllReduce(model, opt, average=args.loss_sync != "global",
                                       always_reduce=solo_rccl)
    img, tgt = synthetic_batch(1234 + rank, args.batch, args.hw, args.hw)
    img, tgt = img.to(dev), tgt.to(dev)
    torch.manual_seed(99 + rank)     # dropout stream differs per rank

    clip = torch.randn(args.batch, 512, args.hw // 32, args.hw // 32, device=dev) if args.clip else None

    def step():
        if clip is None:
            return ua.train_step(model, opt, lossf, img, tgt,
                                 grad_sync=sync.finish if sync else None)
        opt.zero_grad()
        loss = lossf(model(img, clip), tgt)
        loss.backward()
        if sync:
            sync.finish()
        opt.step()
        return loss.detach()

    for _ in range(args.warmup):
        loss = step()
    # HIP events on the launch stream bracket the roofline group (3x3 forward / data gradient)
    # inside the timed region; the other groups are timed in a separate short pass below
    timer = None
    if not args.no_kernel_timer:
        timer = ua.ops.KernelTimer(only={"conv"})

    def groups_pass(nsteps=3):
        gt = ua.ops.KernelTimer()
        fence()
        ua.ops.set_timer(gt)
        for _ in range(nsteps):
            step()
        fence()
        ua.ops.set_timer(None)
        return gt.summary(), nsteps

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    ua.ops.set_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    ua.ops.set_timer(None)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()
    if not (final_loss == final_loss):
        raise SystemExit("loss is NaN")
    gsumm, gsteps = groups_pass() if timer is not None else (None, None)

    def enqueue_ms(reps=5):
        """Host time to ENQUEUE one step onto an idle stream (median of `reps`): the Python /
        ctypes walk over the ~330 entry points, no HIP events, no waiting on the device."""
        ts = []
        for _ in range(reps):
            fence()
            t = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t)
        fence()
        return 1e3 * sorted(ts)[len(ts) // 2]

    enqueue = enqueue_ms()

    # ---- the line as it stands after the EAGER timed region (rank 0 only) ----
    result = None
    if rank == 0:
        images = args.batch * world * args.steps
        value = images / dt
        result = {
            "metric": "images/sec (512x512, bs/GPU=8) Our_UNet train step",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 (layer tensors and MFMA operands bf16; f32 "
                                            "accumulate, statistics, weights, optimizer)",
                      "bf16x3": "f32 (operands split into 3 bf16 terms, 6 bf16 MFMA products per "
                                "multiply, f32 accumulate + storage)"}[args.matmul],
            "data": "synthetic",
            "config": {"workload": ("CLIP_UNet (synthetic CLIP features [N,512,16,16]) " if args.clip else "") +
                                   f"Our_UNet 6-stage 3-class {args.hw}x{args.hw} bs={args.batch}/GPU "
                                   "fp32 train step (fwd + Dice/wCE loss + bwd + SGD-Nesterov), "
                                   "train mode, HIP conv/IN/upsample kernels",
                       "global_batch": args.batch * world, "image": [args.hw, args.hw],
                       "parallelism": f"dp{world}", "world": world,
                       "dist_backend": backend_used},
            # which of the two timed forms of the SAME step `value` is: "eager" (one host call
            # per kernel) or "hip_graph_replay" (the step captured once, one host call per
            # step); the other one is reported under its own key
            "value_path": "eager",
            "final_loss": final_loss,
            # host time to enqueue one step (Python + ctypes, idle stream); the step is
            # launch-bound when this approaches ms_per_step
            "enqueue_ms_per_step": enqueue,
        }
        if timer is not None:
            roof, groups = roofline_of(timer.summary(), args.steps, args.matmul, args, gsumm, gsteps)
            if roof:
                result["roofline"] = roof
            result["kernel_groups"] = groups
            result["kernel_groups_pass"] = (f"{gsteps} extra steps after the timed region with every "
                                            "entry point bracketed by HIP events (the timed region "
                                            "brackets only the roofline group)")
        if cpu_base is not None:
            result["cpu_baseline"] = cpu_base

    emitted = []

    def emit():
        """rank 0 prints THE line (once)."""
        if rank == 0 and not emitted:
            emitted.append(True)
            v = result["value"]
            result["step_tflops"] = v / world * GFLOP_PER_IMAGE * 1e-3
            result["step_frac_of_f32_mfma_peak"] = result["step_tflops"] / PEAK_F32_MFMA_TFLOPS
            print(json.dumps(result), flush=True)

    class Bailout:
        """N > 1 only: a leg that captures RCCL collectives in a HIP graph has never run on more
        than one rank (no multi-GPU node was ever available to rehearse it).  If it has not
        finished after `seconds`, every rank gives up on it: rank 0 prints the line as it stands
        (the eager measurement, which is complete at this point) and the process exits."""

        def __init__(self, seconds, what):
            import threading
            self.timer = threading.Timer(seconds, self.fire)
            self.timer.daemon = True
            self.what = what

        def fire(self):
            if result is not None:
                result.setdefault("notes", []).append(f"{self.what}: no result after the time "
                                                      "limit; the line carries the eager figures")
            emit()
            os._exit(0)

        def __enter__(self):
            if world > 1:
                self.timer.start()
            return self

        def __exit__(self, *exc):
            self.timer.cancel()
            return False

    # The data-parallel step is capturable too (RCCL's collectives become graph nodes on RCCL's
    # stream).  At N = 1 the leg always runs (with UNET_BENCH_RCCL=1 through the one-rank RCCL
    # group, collectives captured).  At N > 1 it is OPT-IN (UNET_BENCH_DDP_GRAPH=1, under the
    # Bailout time limit above): no node with more than one GPU was ever available to rehearse a
    # multi-rank capture, and a failure there need not be an exception (RCCL's watchdog thread
    # aborts the process), which would cost the run its eager line as well.
    ddp_graph = use_dist and backend == "nccl" and \
        os.environ.get("UNET_BENCH_DDP_GRAPH", "1" if world == 1 else "0") != "0"

    def graph_leg():
        """The same K steps replayed from ONE HIP graph (ua.GraphedTrainStep): what the step
        costs once the host walk is out of the way.  Plain UNet only; with a process group the
        bucketed all-reduces are captured with the step (backend nccl = RCCL)."""
        if (use_dist and not ddp_graph) or args.clip or args.no_graph:
            return None
        gstep = ua.GraphedTrainStep(model, opt, lossf, img, tgt, grad_sync=sync)
        for _ in range(2):
            gstep(img, tgt)
        fence()
        t = time.perf_counter()
        for _ in range(args.steps):
            gstep(img, tgt)
        t_enq = time.perf_counter() - t
        fence()
        dt_g = time.perf_counter() - t
        if use_dist:
            tt = torch.tensor([dt_g], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_g = tt.item()
        opt.use_device_hyper(False)
        return {"value": args.batch * world * args.steps / dt_g, "unit": "images/s",
                "ms_per_step": 1e3 * dt_g / args.steps,
                "enqueue_ms_per_step": 1e3 * t_enq / args.steps,
                "collectives_captured": bool(use_dist),
                "note": "train step captured once in a HIP graph and replayed (same kernels, "
                        "same order; dropout masks drawn inside the graph)"}

    def guarded_graph_leg(what):
        try:
            with Bailout(240.0, what):
                return graph_leg()
        except Exception as e:   # capture refused: keep the eager figures
            if world > 1:        # (a half-captured collective leaves the group unusable: stop here)
                if result is not None:
                    result.setdefault("notes", []).append(f"{what}: {type(e).__name__}: {e}"[:300])
                emit()
                os._exit(0)
            return {"error": f"{type(e).__name__}: {e}"[:300]}

    graph = guarded_graph_leg("graph replay of the data-parallel step")
    if rank == 0 and graph is not None:
        result["graph_replay"] = graph
        if graph.get("value", 0.0) > result["value"]:
            # the same step, replayed: what a training loop runs (train.GraphedTrainStep)
            result["eager"] = {"value": result["value"], "unit": "images/s",
                               "ms_per_step": result["ms_per_step"]}
            result["value"], result["ms_per_step"] = graph["value"], graph["ms_per_step"]
            result["value_path"] = "hip_graph_replay"

    # Beside the fp32-matrix-core headline, the same K steps in the two other operand modes
    # (reported, never `value`): the split-bf16 ("bf16x3") mode, which holds the same reference
    # fixtures and tolerances as fp32 (tests/test_net_gpu.py,
    # profiles/r01_bf16x3_accuracy_vs_fp64.txt), and BASELINE config 4, bf16 mixed precision
    # (bf16 layer tensors + bf16 MFMA, fp32 accumulate / statistics / master weights;
    # tests/test_bf16_gpu.py holds it against the oracle on bf16-rounded operands).
    ALT = {"bf16x3": dict(
               dtype="f32 tensors; conv operands split into 3 bf16 terms, 6 bf16 MFMA products "
                     "per multiply, f32 accumulate",
               parity="same fixtures and tolerances as fp32 (1e-4 logits, bit-exact argmax off "
                      "ties); per-conv error vs fp64 <= 2.5e-6",
               note="peak = dense bf16 MFMA peak / 6 products per multiply; the chip holds "
                    "~1.8 GHz under this load (DESIGN.md section 3d)"),
           "bf16": dict(
               dtype="bf16 layer tensors and MFMA operands; f32 accumulate, statistics, weights, "
                     "optimizer (BASELINE config 4)",
               parity="every entry point and the whole net against the oracle evaluated on "
                      "bf16-rounded operands with bf16 stores (tests/test_bf16_gpu.py)",
               note="peak = dense bf16 MFMA peak; this mode is bound by HBM / the gather-GEMM "
                    "loaders, not by the matrix cores")}
    alt = None
    if args.matmul == "fp32" and not args.no_alt and not args.clip:
        alt = {}
        for mode in ("bf16x3", "bf16"):
            model.matmul_precision = mode
            for _ in range(max(2, args.warmup // 2)):
                step()
            fence()
            alt_timer = None if args.no_kernel_timer else ua.ops.KernelTimer(only={"conv"})
            ua.ops.set_timer(alt_timer)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                loss = step()
            fence()
            dt_alt = time.perf_counter() - t1
            ua.ops.set_timer(None)
            if use_dist:
                t = torch.tensor([dt_alt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_alt = t.item()
            alt[mode] = {"value": args.batch * world * args.steps / dt_alt, "unit": "images/s",
                         "ms_per_step": 1e3 * dt_alt / args.steps,
                         "enqueue_ms_per_step": enqueue_ms(), "dtype": ALT[mode]["dtype"],
                         "parity": ALT[mode]["parity"]}
            g_alt = guarded_graph_leg(f"graph replay ({mode})")
            if g_alt is not None:
                alt[mode]["graph_replay"] = g_alt
            if alt_timer is not None:
                asumm, asteps = groups_pass(2)
            if alt_timer is not None and rank == 0:
                roof, groups = roofline_of(alt_timer.summary(), args.steps, mode, args, asumm, asteps)
                if roof:
                    roof["note"] = ALT[mode]["note"]
                    alt[mode]["roofline"] = roof
                alt[mode]["kernel_groups"] = groups
        model.matmul_precision = args.matmul

    if rank == 0 and alt is not None:
        result["alt_modes"] = alt
    emit()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
