#!/usr/bin/env python3
"""Benchmark of the U-Net training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one full training pass over one synthetic batch: forward, fused Dice loss +
confusion counts, backward, gradient all-reduce (N > 1), clip_grad_norm_(1.0), AdamW.
Workload at every N: UNet(in=1, classes=4, base=32), bfloat16 compute, 16 x 512x512 images
per GPU (BASELINE.json configs[1]; N > 1 is configs[2], weak scaling).  Inputs are resident
in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the kernel family with the largest share of a step, timed with HIP events on
                the launch stream in an instrumented replay of the same steps;
  cpu_baseline  the CPU restatement of the reference step (oracle/) on the host cores, on a
                bounded sample of the same workload.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _early_args(argv):
    """--gpus / --force-dist as argparse reads them, before anything that starts the HIP runtime is imported."""
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--force-dist", action="store_true")
    return ap.parse_known_args(argv)[0]


# Runs with a gradient exchange use three streams side by side (the pass, the weight-gradient stream, the collective
# library's): with 4 hardware queues per process (HIP's default, and what the benchmark hosts export) they share queues
# and serialise, with 8 they do not (4.92 -> 4.61 ms per step with a one-rank group; functional.DP_SIDE_STREAM).  The
# HIP runtime reads the variable when it starts, so a run with an exchange raises it to 8 here, before anything touches
# the GPU — unless a profiler's preloaded library has started the runtime before this line: the process then has the
# queues it was started with whatever is assigned here, so the value is left alone (export GPU_MAX_HW_QUEUES=8 ahead of
# the profiler instead).
_early = _early_args(sys.argv[1:])
_exchange = int(os.environ.get("WORLD_SIZE", "1")) > 1 or (_early.gpus or 1) > 1 or _early.force_dist


def _queue_policy(environ, exchange):
    """(value to export or None, who set what is in effect)."""
    had = environ.get("GPU_MAX_HW_QUEUES")
    if not exchange or "rocprof" in environ.get("LD_PRELOAD", ""):
        return None, ("environment" if had is not None else "HIP default")
    if had is not None and had.isdigit() and int(had) >= 8:
        return None, "environment"
    return "8", "bench.py (the environment had %s)" % (had if had is not None else "no value")


_set, HW_QUEUES_FROM = _queue_policy(os.environ, _exchange)
if _set:
    os.environ["GPU_MAX_HW_QUEUES"] = _set

import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_HBM_GBS = 8000.0        # MI355X HBM3E peak (spec)
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA peak (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node (default: WORLD_SIZE when a launcher set it, else 1)")
    ap.add_argument("--steps", type=int, default=200)      # (the first step after the barrier is issue-bound: ~2 ms once)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--model", default="unet", choices=["unet", "a_unet", "rdam_unet"],
                    help="unet: the headline workload (BASELINE configs[1], 1 input channel); a_unet / rdam_unet: the "
                         "attention-gated variants of configs[3] (3 input channels), same step, same JSON objects")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--h2d", action="store_true",
                    help="copy the batch from pinned host memory inside every step (PCIe-inclusive rate; not the "
                         "headline value, whose inputs are resident in HBM)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal of the multi-GPU code path on one GPU: RCCL group of one rank, gradient exchange, "
                         "barriers and the max-over-ranks timing all run (the value is a single-GPU value)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend: nccl (= RCCL over xGMI, the measured configuration) or gloo (rehearsal)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of --gpus N on a box with fewer GPUs: every rank uses cuda:0 (needs --backend gloo; "
                         "RCCL refuses two ranks on one device); the value is NOT a multi-GPU value")
    ap.add_argument("--global-dice", action="store_true",
                    help="pool the Dice sums over all ranks before the ratio (global-batch loss, one extra 12-float "
                         "all-reduce per step) instead of the per-rank loss of N replicas")
    ap.add_argument("--dist-timeout", type=int, default=180,
                    help="seconds a collective (and the rendezvous) may wait for a missing rank before the run fails")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a HIP graph (single GPU; utils/graph_step.py) instead of issuing it "
                         "from Python: same kernels, one host call per step")
    return ap.parse_args()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(threads):
    """The reference's CPU train step as restated by oracle/ (kind "port": the reference itself
    cannot travel to this box), on BASELINE.json configs[0]: UNet(in=1, 4 classes, base 32), fp32,
    batch 4, 256x256, torch.randn images / randint masks from a seed-0 generator, AdamW(8e-4,
    wd 1e-4, betas (0.95, 0.999)), Dice loss, clip_grad_norm_(1.0).  The metrics phase runs the
    reference's own pass structure (7 argmax + 14 one-hot passes per call), so the step costs what
    the reference's step costs.  3 warm-up + 20 timed steps on all cores this process may use
    (capped at 64), and 3 + 10 on 8 threads for comparison with the figure measured where the
    reference itself runs; `value` is the faster of the two."""
    from oracle import unet_oracle as orc
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    limit = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if limit > 0:
        avail = min(avail, limit)
    avail = max(1, min(avail, 64))
    counts = [threads] if threads > 0 else sorted({avail, min(8, avail)}, reverse=True)
    B, S, names = 4, 256, ["a", "b", "c"]
    runs = []
    orc.ATEN_FUSED = True          # BatchNorm / upsampling through the ATen calls the reference makes
    for n_thr in counts:
        torch.set_num_threads(n_thr)
        g = torch.Generator().manual_seed(0)
        ts = orc.TrainState(orc.make_state(1, 4, base=32, seed=1))
        x = torch.randn(B, 1, S, S, generator=g)
        t = torch.randint(0, 4, (B, S, S), generator=g)
        for _ in range(3):
            orc.train_step(ts, x, t, names, per_call_metrics=True)
        timers, n = {}, (20 if n_thr == counts[0] else 10)     # (the comparison run is shorter)
        t0 = time.perf_counter()
        for _ in range(n):
            orc.train_step(ts, x, t, names, timers=timers, per_call_metrics=True)
        dt = time.perf_counter() - t0
        runs.append({"threads": n_thr, "timed_steps": n, "images_per_s": round(B * n / dt, 3), "ms_per_step": round(1e3 * dt / n, 1),
                     "phase_ms_per_step": {k: round(1e3 * v / n, 1) for k, v in timers.items()}})
    best = max(runs, key=lambda r: r["images_per_s"])
    return {"value": best["images_per_s"], "unit": "images/s", "cores": best["threads"], "kind": "port",
            "cpu": cpu_model(),
            "sample": "20 timed steps (after 3 warm-up) of batch 4 at 256x256, fp32 (BASELINE configs[0]): torch-CPU "
                      "restatement of the reference step, metrics computed pass by pass as the reference does",
            "runs": runs}


def pmc_traffic(roof, B, S, dtype):
    """`traffic` of the roofline object: HBM bytes per kernel launch of the dominant family from
    the L2 memory-side counters (2*FETCH_SIZE + WRITE_SIZE), collected over this same command by
    tools/pmc_bench.sh in separate rocprofv3 --pmc runs and stored under profiles/.  Only valid
    for the workload it was collected on (the default one)."""
    import glob
    if not roof or (B, S, dtype) != (16, 512, "bf16"):
        return
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return
    try:
        fam = json.load(open(files[-1]))["families"].get(roof["kernel"])
    except (OSError, ValueError, KeyError):
        return
    if fam:
        roof["traffic"] = fam["traffic_bytes_per_kernel_launch"]
        roof["traffic_unit"] = "bytes per kernel launch (PMC, %s)" % os.path.basename(files[-1])


def spawn_ranks(n):
    """`python bench.py --gpus N` started on its own (no launcher): this process has not touched
    the GPU; it starts one child per GPU with the launcher's environment (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR/PORT on loopback), passes rank 0's stdout (the one JSON line) through,
    sends the other ranks' stdout to stderr, and exits non-zero if any rank failed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, live = 0, dict(enumerate(procs))
    while live:
        time.sleep(0.1)
        for r, p in list(live.items()):
            c = p.poll()
            if c is None:
                continue
            del live[r]
            if c != 0 and rc == 0:
                # the other ranks would wait for this one in their next collective for ever
                print(f"bench.py: rank {r} exited with status {c}; stopping the other ranks", file=sys.stderr)
                rc = c if c > 0 else 1
                for q in live.values():
                    q.terminate()
    raise SystemExit(rc)


def main():
    args = parse()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} was given but the launcher set WORLD_SIZE={world}")
    if args.share_gpu:
        if args.backend != "gloo":
            raise SystemExit("--share-gpu needs --backend gloo")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_dist
    stdout_fd = None
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # stdout carries the one JSON line and nothing else: RCCL prints its version banner (NCCL_DEBUG=VERSION)
        # straight to file descriptor 1 when the communicator is created, so descriptor 1 points at stderr until
        # the line is printed
        sys.stdout.flush()
        stdout_fd = os.dup(1)
        os.dup2(2, 1)
        # a rank that never arrives ends the run with an error after the timeout instead of hanging it
        import datetime
        limit = datetime.timedelta(seconds=args.dist_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, timeout=limit)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)

    from medical_image_segmentation_amd import _lib, functional as F_
    _lib.load()
    from medical_image_segmentation_amd.model.unet import UNet
    from medical_image_segmentation_amd.model.a_unet import A_UNet
    from medical_image_segmentation_amd.model.rdam_unet import RDAM_UNet
    from medical_image_segmentation_amd.utils.loss_fn import diceloss
    from medical_image_segmentation_amd.utils.metrics import Metrics
    from medical_image_segmentation_amd.parallel import GradSync
    from medical_image_segmentation_amd.utils import optim as _optim
    from medical_image_segmentation_amd.utils.optim import clip_and_step

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    names = ["a", "b", "c"]
    torch.manual_seed(1234)
    net, cin = {"unet": (UNet, 1), "a_unet": (A_UNet, 3), "rdam_unet": (RDAM_UNet, 3)}[args.model]
    model = net(cin, 4, 0, base_channels=32, compute_dtype=cdt).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=8e-4, betas=(0.95, 0.999), weight_decay=1e-4)
    sync = None
    if dist_on:
        sync = GradSync(model, always_reduce=True)
        sync.broadcast_parameters()
    glob = bool(args.global_dice and dist_on)
    loss_fn, metric = diceloss(names, global_batch=glob), Metrics(names, global_batch=glob)
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    B, S = args.batch, args.size
    images = torch.randn(B, cin, S, S, generator=g).to(dev)
    masks = torch.randint(0, 4, (B, S, S), generator=g).to(dev)
    params = [p for p in model.parameters()]
    pending = []
    host = (images.cpu().pin_memory(), masks.cpu().pin_memory()) if args.h2d else None

    def step():
        nonlocal images, masks
        if host is not None:
            images, masks = host[0].to(dev, non_blocking=True), host[1].to(dev, non_blocking=True)
        opt.zero_grad()
        logits = model(images)
        ld = loss_fn(logits, masks)
        pending.append(metric.confusion_counts(logits, masks))
        if sync is not None:
            sync.begin_step()
        ld["total_loss"].backward()
        if sync is not None:
            sync.finish_step()
        clip_and_step(opt, params, max_norm=1.0)      # clip_grad_norm_ + AdamW.step (train_and_eval.py:115-118)
        return ld

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    eager_step = step
    if args.graph:
        if world > 1:
            raise SystemExit("--graph is single-GPU")
        from medical_image_segmentation_amd.utils.graph_step import GraphedTrainStep

        def loss_of(m, x, y):
            logits = m(x)
            ld = loss_fn(logits, y)
            return ld["total_loss"], (ld, metric.confusion_counts(logits, y))

        graphed = GraphedTrainStep(model, opt, loss_of, images, masks, max_norm=1.0, warmup=2)

        def step():
            if host is not None:
                _, (ld, counts), _ = graphed(host[0], host[1])
            else:
                _, (ld, counts), _ = graphed(graphed.images, graphed.masks)
            pending.append(counts)
            return ld

    for _ in range(args.warmup):
        step()
    pending.clear()
    # Python's cyclic garbage collector: a full collection of a freshly started interpreter with torch loaded
    # stops the host for 80-90 ms (measured: tools/cold_steps.py; in the first process on a box it fell on
    # step 47, i.e. inside a short timed region).  Collect now and move the survivors out of the collector's
    # sight, as a long-running training job would be past that point anyway.
    gc.collect()
    gc.freeze()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ld = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    loss_val = float(ld["total_loss"].detach())
    bad = loss_val != loss_val or abs(loss_val) == float("inf")
    if dist_on:
        # every rank leaves together: a rank that went on alone would wait in the next collective for ever
        flag = torch.tensor([1.0 if bad else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        bad = bool(flag.item() > 0)
    if bad:
        # a step that produced non-finite numbers did not do the work being measured
        if dist_on:
            dist.destroy_process_group()
        raise SystemExit(f"bench.py: rank {rank}: the loss after the timed steps is {loss_val} "
                         "(or not finite on another rank); no value reported")
    # host post-math of the metrics for the timed steps (outside the timed region: the counts
    # were produced inside it)
    from medical_image_segmentation_amd.utils.train_and_eval import _finish_metrics
    if args.graph:
        pending = pending[-1:]        # the graph's output tensors are overwritten by every replay
    met = _finish_metrics(metric, pending, 4) / max(len(pending), 1)

    roof = None
    if not args.no_roofline:
        # instrumented replay: every rank runs the same 3 steps (they contain the gradient
        # collectives), rank 0 records the per-kernel HIP events
        if rank == 0:
            F_.PROFILE = {}
        for _ in range(3):
            eager_step()              # (graph mode too: the replayed kernels are these)
        torch.cuda.synchronize()
        if rank == 0:
            roof = F_.profile_summary(PEAK_HBM_GBS, PEAK_BF16_TFLOPS, steps=3)
            F_.PROFILE = None
            if args.model == "unet":
                pmc_traffic(roof["dominant"], B, S, args.dtype)

    if rank == 0:
        out = {
            "metric": "train images/sec U-Net 512^2 fwd+bwd+Dice/CE",
            "value": round(B * world * args.steps / dt, 2),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic" + (" (copied from pinned host memory every step)" if args.h2d else ""),
            "config": {"workload": f"{net.__name__}(in={cin},classes=4,base=32) train step, batch {B}/GPU, {S}x{S}, "
                                   f"Dice loss + confusion counts + clip + AdamW",
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "dice_sums": "global batch (all-reduced)" if glob else "per rank",
                       "backend": (("rccl" if args.backend == "nccl" else "gloo") if dist_on else None),
                       "devices": "all ranks on cuda:0 (rehearsal)" if args.share_gpu else "one per rank",
                       "issue": "hip graph replay" if args.graph else "eager",
                       "ranks_in_group": dist.get_world_size() if dist_on else 1,
                       "collective_library": (("rccl %s" % ".".join(map(str, torch.cuda.nccl.version()))
                                               if args.backend == "nccl" else "gloo") if dist_on else None),
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "4 (HIP default)"),
                       "hw_queues_set_by": HW_QUEUES_FROM,
                       "wgrad_stream_with_exchange": bool(F_.DP_SIDE_STREAM) if sync is not None else None},
            "final_loss": round(loss_val, 6),
            "optimizer_table_builds": _optim.PLAN_BUILDS, "optimizer_table_uploads": _optim.PLAN_UPLOADS,
            "mean_dice": round(float(met[2, -1]), 6),
        }
        if roof is not None:
            out["roofline"] = roof["dominant"]
            out["kernels"] = roof["kernels"]
            out["conv_layers"] = roof["conv_layers"]      # every conv shape against its own roof (HBM or matrix cores)
        if not args.no_cpu_baseline and world == 1 and args.model == "unet":
            out["cpu_baseline"] = cpu_baseline(args.cpu_threads)
        if stdout_fd is not None:
            sys.stdout.flush()
            os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
