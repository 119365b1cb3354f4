#!/usr/bin/env python3
"""Headline benchmark: attributions/sec of 50-step Integrated Gradients on ResNet-50 at
3x224x224 (BASELINE.json metric; config[1] "IG 50 steps, ResNet-50, 32-image batch").

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of 32 synthetic images per GPU:
K1 interpolation -> classifier forward/backward (PyTorch-ROCm, fp32) for 32 x 50 interpolants
-> K2 accumulation with the fused |sum_c| epilogue.  Inputs are resident in HBM before the
timed region.  Images shard over ranks with no data-path collective (weak scaling): every rank
works on its own 32 images; the timed region is bracketed by barrier + synchronize and the
reported time is the MAX over ranks.

The headline `value` is measured in the PARITY configuration (`--mode parity`, the default): MIOpen immediate mode with
deterministic solvers only and ONE image = the reference's 50-interpolant batch per classifier pass (saliencyMethods.py:40-46)
-- the configuration every parity test runs and the 1e-5 claim is made on; its classifier passes are run-to-run bit-identical.
Consecutive passes run side by side on `--streams` HIP streams (default 3), one host thread per stream, each thread replaying its pass
as a hipGraph it captured on its own library handles (xai_engine/streams.py): the same kernels on the same shapes, so the maps are
bit-identical to the one-stream run, while the low-occupancy layers of one pass overlap another pass's work.
`--mode throughput` is the fastest configuration instead (the shipped MIOpen find-db's solvers, which include split-K kernels
that are not run-to-run reproducible, and 2 images per pass); at N = 1 the default run measures it in a child process and
reports it as `throughput_mode`.

`--workload sweep` (opt-in; the driver's contract line is the default `ig` workload) measures north_star's scaling
target instead: the insertion/deletion sweep of BASELINE config 5 over ONE fixed list of `--sweep-images` synthetic images
(strong scaling: image i belongs to rank i % world, the list does not grow with N), every image attributed with each of
`--sweep-methods` and pushed through the ten metrics (224 steps each); one 96-byte all-reduce per method; value = images/s.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline        the IG accumulation kernel (xai_ig_accum_f32): algorithmic bytes per launch ((S+2)*4N per image, SURVEY
                  8(d)) / mean launch duration measured with HIP events inside the timed steps, against the 8 TB/s HBM peak
  parity_mode     says whether the headline IS the parity configuration (it is by default), and repeats its figures
  single_stream   the same step with every pass on one stream (what the stream overlap buys)
  reference_api   the same 32 images as 32 calls of the reference's one-image signature IG(input, model, 50, 50, 1, 0, device, t),
                  issued serially and round-robin on the streams
  unfused_classifier  the same step with the classifier left to PyTorch's own BatchNorm / ReLU kernels (N = 1)
  throughput_mode `--mode throughput` measured by a child process (N = 1)
  sweep_strong    north_star's scaling target measured by the SAME command: a FIXED list of --strong-images synthetic images
                  (it does not grow with N; image i -> rank i % N), IG + the ten insertion/deletion metrics (224 steps each)
                  per image, one 96-byte all-reduce; images/s = list length / max-over-ranks time
  cpu_baseline    the CPU oracle (oracle/ig.py, a port of the reference's IG) on the host cores,
                  a bounded sample (four attributions), rank 0 at N=1 only
`--lean` skips every side leg (single_stream ... cpu_baseline).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "image-classification-xai_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0          # the guide's measured float4 copy (read + write); a read-only stream reaches 6.75 TB/s on this part
STEPS_IG, C, H, W = 50, 3, 224, 224
N_ELEM = C * H * W


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--mode", choices=["parity", "throughput"], default="parity",
                    help="parity (default; the headline): deterministic MIOpen solvers in immediate mode, one image = 50 interpolants per "
                         "classifier pass -- the configuration of the parity tests; throughput: the shipped find-db's solvers, 2 images per pass")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams consecutive classifier passes (sweep: consecutive images) are queued on")
    ap.add_argument("--images-per-pass", type=int, default=None, help="images x 50 interpolants per classifier pass (default: 1 parity / 2 throughput)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lean", action="store_true", help="headline + roofline only: skip single_stream, reference_api, unfused_classifier, "
                    "throughput_mode, sweep_strong and cpu_baseline")
    ap.add_argument("--strong-images", type=int, default=256, help="length of the fixed image list of the sweep_strong leg (IG + ten metrics "
                    "per image, image i -> rank i %% N); 0 = skip the leg")
    ap.add_argument("--workload", choices=["ig", "sweep"], default="ig", help="ig = BASELINE config 2 (the contract line); sweep = config 5, "
                    "strong scaling over a fixed image list")
    ap.add_argument("--sweep-images", type=int, default=1000, help="--workload sweep: length of the (global) image list")
    ap.add_argument("--sweep-methods", default="grad,inp_x_grad,ig,lig,idg,gc", help="--workload sweep: attribution methods per image")
    ap.add_argument("--deterministic", type=int, default=None, help="override the mode: 1 = torch.backends.cudnn.deterministic (MIOpen: "
                    "deterministic solvers only, immediate mode; run-to-run bit-identical classifier passes, see profiles/r02_resnet_determinism_*.json)")
    ap.add_argument("--channels-last", type=int, default=0, help="1 = NHWC classifier weights (slower with MIOpen fp32 on gfx950)")
    ap.add_argument("--fuse-bn-relu", type=int, default=1, help="1 = run the classifier's eval-mode BatchNorm + ReLU (+ residual add) as one HIP "
                    "kernel per direction (xai_engine/prepare.py: fuse_bn_relu; every call site is verified bit-identical to the PyTorch "
                    "kernels before use); the unfused classifier is timed too and reported beside it")
    ap.add_argument("--fold-bn", type=int, default=0, help="1 = fold eval-mode BatchNorm into the convolutions (opt-in, see xai_engine/prepare.py)")
    ap.add_argument("--miopen-find", type=int, default=0, help="1 = torch.backends.cudnn.benchmark (MIOpen exhaustive find, minutes on a fresh box)")
    ap.add_argument("--miopen-db", type=int, default=None, help="override the mode: 1 = reuse the shipped MIOpen find-db "
                    "(image-classification-xai_amd/miopen_db, recorded by one exhaustive find of this workload on an MI355X) so that find mode "
                    "costs no search time; 0 = immediate mode")
    args = ap.parse_args()
    throughput = args.mode == "throughput"
    if args.deterministic is None:
        args.deterministic = 0 if throughput else 1
    if args.miopen_db is None:
        args.miopen_db = 1 if throughput else 0
    if args.images_per_pass is None:
        args.images_per_pass = 2 if throughput else 1
    return args


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline():
    """Time the oracle's IG (reference algorithm, NumPy element-wise + torch CPU classifier) on
    four synthetic images: 50 steps, batch 50 -- the same per-attribution work as the GPU leg."""
    from oracle import ig as oig
    from xai_engine.zoo import resnet50
    torch.set_num_threads(host_cores())
    model = resnet50(seed=0)
    n_attr = 4
    xs = torch.randn(n_attr, C, H, W, generator=torch.Generator().manual_seed(2)).numpy()
    with torch.no_grad():
        targets = model(torch.from_numpy(xs)).argmax(1).tolist()
    times = []
    for i in range(n_attr):
        t0 = time.perf_counter()
        oig.ig(xs[i:i + 1], model, STEPS_IG, 50, 1, 0, targets[i])
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2] if len(times) % 2 else sum(sorted(times)[len(times) // 2 - 1:len(times) // 2 + 1]) / 2
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": 1.0 / med, "unit": "attributions/s", "cores": torch.get_num_threads(), "kind": "port", "cpu_model": cpu,
            "sample": f"{n_attr} attributions (IG {STEPS_IG} steps, batch 50, ResNet-50 fp32, 3x224x224), median of "
                      f"{', '.join(f'{t:.2f}' for t in times)} s"}


class SyntheticImages:
    """Fixed global list of synthetic images (seed 1000 + i), generated on access: every rank sees
    the same list whatever the world size, and only its own images are ever materialised."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.n))]
        if not 0 <= i < self.n:
            raise IndexError(i)
        return torch.randn(1, C, H, W, generator=torch.Generator().manual_seed(1000 + i))


def run_sweep_workload(args, model, dev, rank, world, prep, miopen_mode, fence, max_over_ranks):
    """BASELINE config 5 as a strong-scaling benchmark; prints the JSON line on rank 0."""
    from xai_engine.sweep import sweep_images, get_CNN_attr, KEYS
    methods = [m for m in args.sweep_methods.split(",") if m]
    images = SyntheticImages(args.sweep_images)
    td = {"models": [model], "img_hw": H, "batch_size": 50, "device": str(dev), "device_maps": True}

    def one_pass(imgs):
        out = {}
        for m in methods:
            tdm = dict(td, attr_func=m)
            total, used, _ = sweep_images(imgs, model, dev, lambda x, t, tdm=tdm: get_CNN_attr(x, None, t, tdm), img_hw=H, batch_size=50,
                                          rank=rank, world=world, streams=args.streams, kind=m)
            out[m] = {k: total[k] / max(used, 1) for k in KEYS}
            out[m]["images"] = used
        return out

    n_warm = max(2, args.streams) * world
    log(f"sweep workload: {len(images)} images x {methods}; warmup x{args.warmup} (reduced: {n_warm} images)")
    for _ in range(args.warmup):
        one_pass(SyntheticImages(n_warm))
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        means = one_pass(images)
    fence()
    dt = max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        n = len(images)
        print(json.dumps({
            "metric": "images/sec (insertion/deletion sweep: 10 metrics x 224 steps per image and method, ResNet-50 224^2)",
            "value": n * args.steps / dt, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"insertion/deletion sweep over a fixed list of {n} synthetic 3x224x224 images (seeds 1000..), ResNet-50 "
                                   f"(seeded random weights), methods {methods}, 224 perturbation steps, batch 50; one step = the whole list",
                       "images": n, "methods": methods, "image_method_pairs_per_s": n * len(methods) * args.steps / dt, "streams": args.streams,
                       "mode": args.mode, "warmup_step": f"a {n_warm}-image sweep per method (not the full list)", "classifier_prep": prep,
                       "miopen": miopen_mode, "parallelism": f"image i -> rank i % {world}; one all-reduce(SUM) of 12 fp64 (96 B) per method"},
            "metric_means": means}), flush=True)


def sweep_strong_leg(n_images, model, dev, rank, world, fence, max_over_ranks, streams):
    """Strong-scaling leg of the default line: IG (50 steps) + the ten insertion/deletion numbers of every image of ONE fixed list,
    image i on rank i % world, one all-reduce(SUM) of 12 fp64.  Returns the object for the JSON line (the same on every rank)."""
    from xai_engine.sweep import sweep_images, get_CNN_attr, KEYS, FORWARD_COUNTS
    td = {"models": [model], "img_hw": H, "batch_size": 50, "device": str(dev), "device_maps": True, "attr_func": "ig"}

    def one_pass(imgs):
        return sweep_images(imgs, model, dev, lambda x, t: get_CNN_attr(x, None, t, td), img_hw=H, batch_size=50, rank=rank, world=world,
                            streams=streams, kind="ig")

    one_pass(SyntheticImages(max(2, streams) * world))    # warm-up: one image per stream and rank (solver selection, allocator, workspaces)
    fence()
    t0 = time.perf_counter()
    total, used, attr_s = one_pass(SyntheticImages(n_images))
    fence()
    dt = max_over_ranks(time.perf_counter() - t0)
    return {"value": n_images / dt, "unit": "images/s", "images": n_images, "seconds": dt, "scaling": "strong", "streams": streams,
            "workload": f"insertion/deletion sweep over a fixed list of {n_images} synthetic 3x224x224 images (seeds 1000..): IG 50 steps + ten "
                        "metrics x 224 perturbation steps per image, batch 50 (BASELINE config 5 restricted to one method)",
            "parallelism": f"image i -> rank i % {world}; one all-reduce(SUM) of 12 fp64 (96 B)", "images_used": used,
            "forward_batches_warmup_and_timed": dict(FORWARD_COUNTS),
            "metric_means": {k: total[k] / max(used, 1) for k in KEYS}}


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child torchrun (nothing has touched the
    GPU yet in this process) and pass its exit code on."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def throughput_child(args):
    """`--mode throughput` in a CHILD process (MIOpen reads its user find-db path once per process, and the parity headline must
    not see that db): a child, not an exec -- this process has initialised the GPU.  -> the object for the JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--mode", "throughput", "--lean", "--gpus", "1", "--steps", str(max(2, min(args.steps, 5))),
           "--warmup", "1", "--images", str(args.images), "--streams", str(args.streams), "--fuse-bn-relu", str(args.fuse_bn_relu)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"child exited with {r.returncode}", "stderr_tail": r.stderr[-400:]}
        d = json.loads(lines[-1])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                "images_per_pass": d["config"]["images_per_pass"], "streams": d["config"]["streams"], "miopen": d["config"]["miopen"],
                "classifier_passes_warmup_and_timed": d["config"].get("classifier_passes_warmup_and_timed"),
                "k2_avg_launch_ms": d["roofline"]["avg_launch_ms"], "k2_frac_of_hbm_peak": d["roofline"]["frac"],
                "note": "same step, same work, fp32: MIOpen's find-db solvers (they include split-K kernels that are not run-to-run "
                        "reproducible: IG twice differs by ~5e-4, profiles/r02_resnet_determinism_finddb.json) and 2 images = 100 interpolants "
                        "per classifier pass, a batch the reference's one-image API cannot form; the attribution kernels of this composition "
                        "are held bit-identical to the oracle on the same batches (tests/test_gpu_configs.py::"
                        "test_config2_benched_composition_two_images_per_pass); measured by a child process of this run"}
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as e:
        return {"error": f"{type(e).__name__}: {e}"}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    from xai_engine.prepare import use_tuned_miopen_db
    # --deterministic: no find mode either -- find picks among the deterministic solvers by timing, so two processes may settle on different
    # ones (measured: 7e-8 between a 1-rank and a 2-rank sweep of the same list); without find the solver is a function of the shape
    tuned = bool(args.miopen_db) and not args.channels_last and not args.fold_bn and not args.deterministic and use_tuned_miopen_db(rank)
    # rehearsal knobs (never set by the driver): XAI_DIST_BACKEND=gloo + XAI_FORCE_DEVICE=0 let several ranks share
    # the one GPU of a test box so that the N>1 control flow (barrier, max-over-ranks, rank-0 print) can be exercised
    backend = os.environ.get("XAI_DIST_BACKEND", "nccl")
    dev_index = int(os.environ.get("XAI_FORCE_DEVICE", local))
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    import xai_engine
    xai_engine.load_library()                      # no extension -> no benchmark
    from xai_engine.ig import ig_batch, IG
    from xai_engine.streams import run_on_streams
    from xai_engine.zoo import resnet50

    torch.backends.cudnn.benchmark = bool(args.miopen_find) or tuned
    torch.backends.cudnn.deterministic = bool(args.deterministic)
    model = resnet50(seed=0).to(dev)
    if args.fold_bn:
        from xai_engine.prepare import fold_batchnorm
        model = fold_batchnorm(model)
    if args.channels_last:
        model = model.to(memory_format=torch.channels_last)
    B = args.images
    x = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(2 + rank)).to(dev)
    plain_model, prep = model, ("conv+bn folded" if args.fold_bn else "none")
    if args.fuse_bn_relu and not args.fold_bn and not args.channels_last:
        from xai_engine.prepare import fuse_bn_relu
        try:
            model = fuse_bn_relu(plain_model, verify=x[:2], fork_residual=True)
            prep = ("eval-mode BN+ReLU(+residual add) fused into one HIP kernel per direction; every fused call site verified "
                    "bit-identical (forward and gradients) to the PyTorch kernels on this device before use")
        except ValueError as e:                       # never silently: say so in the line and run the classifier as given
            log(f"classifier fusion refused: {e}")
            prep = f"none (fusion refused: {e})"
    miopen_mode = ("find mode with shipped find-db" if tuned else ("find" if args.miopen_find else "immediate mode")) + \
        (", deterministic solvers only" if args.deterministic else "")

    def fence():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(dt):
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        return dt

    if args.workload == "sweep":
        run_sweep_workload(args, model, dev, rank, world, prep, miopen_mode, fence, max_over_ranks)
        if world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()
        return

    with torch.no_grad():
        targets = plain_model(x).argmax(1)
    grads = torch.empty((B, STEPS_IG, C, H, W), dtype=torch.float32, device=dev)
    events = []

    def step(sink=None, net=None, streams=args.streams):
        return ig_batch(x, net if net is not None else model, targets, steps=STEPS_IG, alpha_star=1, baseline=0,
                        images_per_pass=args.images_per_pass, want_abs=True, grads_buffer=grads, event_sink=sink, streams=streams)

    def timed(fn, n):
        """n calls of fn between two fences -> seconds per call, max over ranks"""
        fence()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        fence()
        return max_over_ranks(time.perf_counter() - t) / n

    if tuned and args.images_per_pass not in (1, 2):
        log("NOTE: the shipped find-db holds the shapes of --images-per-pass 1 and 2 only; for other batch shapes MIOpen's find mode searches "
            "first (minutes per new shape on a fresh box) -- pass --miopen-db 0 for immediate mode")
    log(f"model + inputs ready on {dev}; {args.mode} mode ({miopen_mode}), {args.images_per_pass} image(s) per pass, {args.streams} stream(s); "
        f"warmup x{args.warmup}")
    for _ in range(args.warmup):
        step()
    fence()
    log("warmup done; timing")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    fence()
    dt = max_over_ranks(time.perf_counter() - t0)
    from xai_engine.ig import PASS_COUNTS
    how = dict(PASS_COUNTS)
    log(f"timed {args.steps} steps in {dt:.3f} s ({world * B * args.steps / dt:.1f} attributions/s); classifier passes so far: {how}")

    algo_bytes = B * (STEPS_IG + 2) * 4 * N_ELEM + B * H * W * 4      # read S grads + x, write out (+ |sum_c| map); b is a scalar
    # two HIP-event timings of the accumulation launches of the timed steps: events bracketing each launch (they include the
    # dispatch latency, ~5 us) and events the dispatch itself stamps at the kernel's start and stop (hipExtLaunchKernel);
    # the second is the kernel's duration and is what the roofline uses, unless the runtime hands back nonsense
    bracket_ms = sum(e[0].elapsed_time(e[1]) for e in events) / max(len(events), 1)
    try:
        kernel_ms = sum(e[2].elapsed_time(e[3]) for e in events) / max(len(events), 1)
    except RuntimeError:
        kernel_ms = 0.0
    stamped = 0.0 < kernel_ms <= bracket_ms             # the kernel cannot take longer than the events around its launch
    kern_ms = kernel_ms if stamped else bracket_ms
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0

    single = api = unfused = strong = None
    if not args.lean:
        if args.streams > 1:
            step(streams=1)
            d1 = timed(lambda: step(streams=1), 2)
            single = {"value": world * B / d1, "unit": "attributions/s", "ms_per_step": d1 * 1e3, "steps": 2, "streams": 1,
                      "note": "same run, every classifier pass on one stream: bit-identical maps, no overlap between passes"}
            log(f"single stream: {d1 * 1e3:.1f} ms/step")

        # the reference's one-image signature, 32 calls: serially as the reference's harness calls it, and round-robin on the streams
        def api_serial():
            return [IG(x[i:i + 1], model, STEPS_IG, 50, 1, 0, dev, targets[i]) for i in range(B)]

        def api_streams():                              # one host thread per stream (xai_engine/streams.py); IG's backward passes take turns
            return run_on_streams(dev, args.streams, [lambda i=i: IG(x[i:i + 1], model, STEPS_IG, 50, 1, 0, dev, targets[i]) for i in range(B)],
                                  kind="IG one image")

        api_serial()
        da = timed(api_serial, 1)
        api = {"serial": {"value": world * B / da, "unit": "attributions/s", "ms_per_step": da * 1e3, "steps": 1}}
        if args.streams > 1:
            api_streams()
            ds = timed(api_streams, 1)
            api["on_streams"] = {"value": world * B / ds, "unit": "attributions/s", "ms_per_step": ds * 1e3, "steps": 1, "streams": args.streams}
        api["note"] = (f"the same {B} images as {B} calls of IG(input, model, 50, 50, 1, 0, device, target) -- the reference's one-image signature "
                       "(alpha_star == 1 streams the step gradients into a (C,H,W) accumulator: no gradient buffer, no filing copy); `serial` = one "
                       "call after the other on one stream, `on_streams` = the caller issues the calls from one host thread per HIP stream "
                       "(xai_engine.streams.run_on_streams)")
        log("one-image API: " + ", ".join(f"{k} {v['ms_per_step']:.1f} ms" for k, v in api.items() if isinstance(v, dict)))

        if model is not plain_model and world == 1:       # the same workload on the classifier exactly as given, for the record
            step(net=plain_model)
            du = timed(lambda: step(net=plain_model), 2)
            unfused = {"value": B / du, "unit": "attributions/s", "ms_per_step": du * 1e3, "steps": 2,
                       "note": "same run, classifier left as PyTorch modules (no BN/ReLU fusion)"}
            log(f"unfused classifier: {du * 1e3:.1f} ms/step")

        if args.strong_images > 0:
            strong = sweep_strong_leg(args.strong_images, model, dev, rank, world, fence, max_over_ranks, args.streams)
            log(f"sweep_strong: {strong['images']} images in {strong['seconds']:.2f} s ({strong['value']:.2f} images/s)")

    if rank == 0:
        traffic = traffic_source = None
        pmc = os.path.join(ROOT, "profiles", "ig_accum_pmc.json")     # written from a separate rocprofv3 --pmc run
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            traffic_source = ("profiles/ig_accum_pmc.json: FETCH_SIZE/WRITE_SIZE of this kernel at this shape from a separate "
                              "rocprofv3 --pmc run (PMC passes cannot share a process with the timed run); NOT measured in this process")
        peak_gb = torch.cuda.max_memory_allocated(dev) / 2 ** 30         # everything this rank held at once, graphs' private pools included
        parity_cfg = bool(args.deterministic) and args.images_per_pass == 1 and not tuned and not args.miopen_find
        value = world * B * args.steps / dt
        line = {
            "metric": "attributions/sec (IG 50-step ResNet-50 224^2)",
            "value": value,
            "unit": "attributions/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"IG 50 steps, ResNet-50 (seeded random weights), {B}-image batch of 3x224x224 per GPU, "
                                   "alpha_star=1, baseline=0", "images_per_gpu": B, "ig_steps": STEPS_IG, "mode": args.mode,
                       "images_per_pass": args.images_per_pass, "streams": args.streams, "classifier_passes_warmup_and_timed": how,
                       "classifier_prep": prep, "miopen": miopen_mode, "peak_device_memory_gib_rank0": round(peak_gb, 2),
                       "parallelism": f"image-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "xai_ig_accum_f32", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "measured_copy_peak": HBM_COPY_GBS, "frac_of_measured_copy": achieved / HBM_COPY_GBS, "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": algo_bytes,
                         "avg_launch_ms": kern_ms, "launches_timed": len(events),
                         "timing": ("HIP events stamped by the dispatch at kernel start / stop (hipExtLaunchKernel), mean over the timed steps' launches"
                                    if stamped else "HIP events bracketing each launch (kernel-stamped events unavailable)"),
                         "avg_launch_ms_events_bracketing_the_launch": bracket_ms},
            "parity_mode": {"is_headline": parity_cfg, "value": value if parity_cfg else None, "ms_per_step": dt / args.steps * 1e3 if parity_cfg else None,
                            "k2_avg_launch_ms": kern_ms if parity_cfg else None,
                            "note": ("`value` above IS the parity configuration: deterministic MIOpen solvers in immediate mode, one image = the "
                                     "reference's 50-interpolant batch per classifier pass (saliencyMethods.py:40-46); the maps are bit-identical to "
                                     "the oracle fed the same classifier outputs and within 1e-5 of the oracle / the reference's CPU outputs "
                                     "(tests/test_gpu_configs.py, tests/test_gpu_e2e.py), whatever --streams is (bit-identical to one stream: "
                                     "tests/test_gpu_configs.py::test_classifier_passes_on_several_streams_are_bit_identical_to_one_stream)") if parity_cfg else
                                    "this run is NOT the parity configuration (see config.mode / config.miopen / config.images_per_pass)"},
        }
        if not args.lean:
            line.update({"single_stream": single, "reference_api": api, "unfused_classifier": unfused, "sweep_strong": strong})
            if world == 1 and args.mode == "parity":
                torch.cuda.empty_cache()
                log("throughput_mode: measuring in a child process")
                line["throughput_mode"] = throughput_child(args)
                tm = line["throughput_mode"]
                log("throughput_mode: " + (f"{tm['value']:.1f} attributions/s" if "value" in tm else str(tm)))
            if world == 1 and not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
