"""Integrated-Gradients family on the HIP kernels.

The classifier forward/backward stays PyTorch-ROCm (`getGradientsParallel`); the path
interpolation, the Left-IG cutoff and the Riemann accumulation run in libxai_hip.so.
Mirrors util/attribution_methods/saliencyMethods.py of the reference (signatures, return
shapes, the print-and-return-zeros error convention); `ig_batch` is the multi-image fast
path the reference does not have.
"""
import contextlib
import threading

import torch

from . import kernels as K
from ._lib import XaiHipError
from .streams import CAPTURE_LOCK, backward_turn, on_worker, run_on_streams


def hip_device(device):
    """'cuda:N' / torch.device -> torch.device on a HIP GPU; anything else raises: the
    product has no CPU path."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise XaiHipError(f"device '{device}' is not a HIP GPU: xai_engine runs on 'cuda:N' (ROCm) only")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _logits_of(output):
    return output if isinstance(output, torch.Tensor) else output.logits


def _select_class(output, target_class):
    """output[:, target_class].  PyTorch's indexing turns a 0-d device tensor index into a Python int with .item():
    a host sync after every forward pass (and illegal inside a hipGraph capture); index_select reads the index on the
    device instead.  Same values, same gradient."""
    if torch.is_tensor(target_class) and target_class.is_cuda and target_class.dim() == 0:
        return output.index_select(1, target_class.reshape(1).to(output.device)).squeeze(1)
    return output[:, target_class]


def getGradientsParallel(inputs, model, target_class):
    """d logit[target] / d inputs for a batch; raw logits (reference saliencyMethods.py:209-215)."""
    output = _logits_of(model(inputs))
    scores = _select_class(output, target_class)
    with backward_turn(inputs.device) if inputs.is_cuda else contextlib.nullcontext():      # streams.py
        gradients = torch.autograd.grad(scores, inputs, grad_outputs=torch.ones_like(scores))[0]
    return gradients.detach().squeeze(), scores.detach().squeeze()


def getPredictionParallel(inputs, model, target_class):
    """(reference saliencyMethods.py:218-224)"""
    output = _logits_of(model(inputs))
    return _select_class(output, target_class).detach().squeeze()


def input_grad(input, model, target_class):
    """(reference saliencyMethods.py:7-11)"""
    input.requires_grad = True
    gradient, _ = getGradientsParallel(input, model, target_class)
    input.requires_grad = False
    return gradient


_ALPHAS = {}


def _uniform_alphas(steps, dev):
    """linspace(0, 1, steps) computed on the HOST like the reference (saliencyMethods.py:21; ATen's CPU fill, not the device's, decides
    the last bit) and kept on the device: uploading 50 floats from pageable memory on every call blocks the host until the
    stream has drained, which serialises callers that overlap attributions on several streams."""
    key = (int(steps), str(dev))
    t = _ALPHAS.get(key)
    if t is None:
        t = torch.linspace(0, 1, steps).to(dev)
        torch.cuda.current_stream(dev).synchronize()          # once: every stream may read it from here on
        _ALPHAS[key] = t
    return t


def _prep(input, baseline, device):
    dev = hip_device(device)
    x = input.to(dev, torch.float32).contiguous()
    if torch.is_tensor(baseline):
        base = baseline.to(dev, torch.float32).contiguous()
        if base.shape != x.shape:
            base = base.expand_as(x).contiguous()
    else:
        base = float(baseline)
    return dev, x, base


def _path(x, base, alphas, model, batch_size, target_class, want_grads=True):
    """Walk the path in `batch_size` chunks.  x: (1,C,H,W); alphas: (steps,) on the device.
    Returns grads (1, steps, C,H,W) (or None) and logits (1, steps)."""
    steps = alphas.shape[0]
    dev = x.device
    grads = torch.empty((1, steps) + tuple(x.shape[1:]), dtype=torch.float32, device=dev) if want_grads else None
    logits = torch.empty((1, steps), dtype=torch.float32, device=dev)
    for lo in range(0, steps, batch_size):
        hi = lo + batch_size
        imgs = K.ig_interp(x, base, alphas[lo:hi])[0]                    # (batch, C,H,W), a fresh leaf
        if want_grads:
            imgs.requires_grad_(True)
            g, s = getGradientsParallel(imgs, model, target_class)
            K.store_grads(g.contiguous(), grads[0, lo:hi])
        else:
            with torch.no_grad():
                s = getPredictionParallel(imgs, model, target_class)
        logits[0, lo:hi] = s.reshape(-1)
    return grads, logits


def _worker_pass(x, base, alphas, model, batch_size, target_class):
    """On a stream worker, when ONE classifier pass covers the whole path (batch_size == steps): this thread's hipGraph of that pass
    (`_CapturedPass`, shared with ig_batch) -> (gradients (steps, C,H,W), logits (steps,)), else None.  Same kernels, same bits."""
    steps = alphas.shape[0]
    if not on_worker() or batch_size != steps or x.shape[0] != 1:
        return None
    if torch.is_tensor(target_class):
        if not target_class.is_cuda:
            return None                                    # a host index would need an upload (a host sync on this stream): stay eager
        t = target_class.reshape(1).long()
    else:
        t = torch.full((1,), int(target_class), dtype=torch.int64, device=x.device)
    cp = _thread_pass(model, 1, steps, tuple(x.shape[1:]), x.device, alphas, base)
    if cp is None:
        return None
    PASS_COUNTS["replayed"] += 1
    return cp(x, t, base if torch.is_tensor(base) else None)


def _path_sum(x, base, alphas, model, batch_size, target_class):
    """Walk the path like `_path`, but keep only the running sum of the step gradients: each pass's autograd gradient
    (cache-resident, `batch_size` x N) is added straight into a (1,C,H,W) fp32 accumulator by the streaming form of K2.
    No (steps, N) buffer exists, nothing is filed with a copy and read back.  Per element the sum runs 0 + g_0 + g_1 + ...
    over ascending steps in fp32 -- the order of the buffered K2 launch, so the two forms are bit-identical
    (tests/test_gpu_kernels.py::test_ig_streaming_form_equals_buffered, test_IG_streams_when_alpha_star_is_1)."""
    steps = alphas.shape[0]
    acc = torch.zeros_like(x)
    replay = _worker_pass(x, base, alphas, model, batch_size, target_class)
    if replay is not None:
        K.ig_accum_add(replay[0], acc[0])
        return acc
    for lo in range(0, steps, batch_size):
        imgs = K.ig_interp(x, base, alphas[lo:lo + batch_size])[0].requires_grad_(True)
        g, _ = getGradientsParallel(imgs, model, target_class)
        K.ig_accum_add(g.reshape(imgs.shape).contiguous(), acc[0])
    return acc


def IG(input, model, steps, batch_size, alpha_star, baseline, device, target_class):
    """IG (alpha_star == 1) / Left-IG of one image (1,C,H,W) -> (C,H,W) on the device
    (reference saliencyMethods.py:13-72).
    alpha_star == 1 needs no per-step state (`gradients.mean(dim=0)`, :53): the step gradients are summed as they are
    produced (`_path_sum`) and `xai_ig_finish_f32` applies / steps * (x - b).  Left-IG's cutoff depends on all the logits
    (:48-65), so it keeps the reference's (steps, C,H,W) gradient buffer and reduces its prefix with one K2 launch."""
    if steps % batch_size != 0:
        print("steps must be evenly divisible by batch size: " + str(batch_size) + "!")
        return 0, 0, 0, 0
    dev, x, base = _prep(input, baseline, device)
    alphas = _uniform_alphas(steps, dev)                  # computed on the host, as the reference does
    if alpha_star == 1:
        return K.ig_finish(_path_sum(x, base, alphas, model, batch_size, target_class), steps, x, base)[0]
    grads, logits = _path(x, base, alphas, model, batch_size, target_class)
    return K.ig_accum(grads, x, base, n_use=K.ig_cutoff(logits, alpha_star))[0]


def getSlopes(baseline, baseline_diff, model, steps, batch_size, device, target_class):
    """Finite-difference logit slopes on the uniform path (reference saliencyMethods.py:226-261).
    Takes baseline and (input - baseline) like the reference."""
    if steps % batch_size != 0:
        print("steps must be evenly divisible by batch size: " + str(batch_size) + "!")
        return 0, 0
    dev = hip_device(device)
    base = baseline.to(dev, torch.float32).contiguous()
    x = (base + baseline_diff.to(dev, torch.float32)).contiguous()
    cpu_alphas = torch.linspace(0, 1, steps)
    _, logits = _path(x, base, cpu_alphas.to(dev), model, batch_size, target_class, want_grads=False)
    x_diff = float(cpu_alphas[1] - cpu_alphas[0])
    # 50 numbers: finish on the host so the division is the IEEE one the reference's CPU path
    # performs (a device tensor / python scalar is a multiply by the reciprocal)
    lg = logits[0].cpu()
    slopes = torch.zeros(steps)
    slopes[1:] = (lg[1:] - lg[:-1]) / x_diff
    return slopes.to(dev), x_diff


def getAlphaParameters(slopes, steps, step_size):
    """IDG's slope-proportional sample placement (reference saliencyMethods.py:264-314):
    a 50-element host computation, returned as CPU tensors like the reference."""
    s = slopes.detach().float().cpu()
    unit = (s - s.min()) / (s.max() - s.min())
    unit[0] = 0
    share = unit / unit.sum()
    want = share * steps
    count = want.type(torch.int)
    spare = steps - int(count.sum())
    want[torch.where(count != 0)[0]] = -1
    by_need = torch.flip(torch.sort(want)[1], dims=[0])
    count[by_need[0:spare]] = 1
    alphas = torch.zeros(steps)
    substep = torch.zeros(steps)
    at, a0 = 0, 0
    for n in count:
        n = int(n)
        if n == 0:
            continue
        alphas[at:at + n] = torch.linspace(a0, a0 + step_size, n + 1)[0:n]
        substep[at:at + n] = step_size / n
        at += n
        a0 += step_size
    return alphas, substep


def IDG(input, model, steps, batch_size, baseline, device, target_class):
    """Integrated Decision Gradients (reference saliencyMethods.py:74-136)."""
    if batch_size == 0 or steps % batch_size != 0:
        print("steps must be evenly divisible by batch size!")
        return 0, 0, 0
    dev, x, base = _prep(input, baseline, device)
    base_t = base if torch.is_tensor(base) else torch.full_like(x, base)
    slopes, step_size = getSlopes(base_t, x - base_t, model, steps, batch_size, dev, target_class)
    alphas, substep = getAlphaParameters(slopes, steps, step_size)
    alphas = alphas.to(dev)
    grads, logits = _path(x, base, alphas, model, batch_size, target_class)
    w = torch.zeros(steps, device=dev)
    w[1:] = (logits[0, 1:] - logits[0, :-1]) / (alphas[1:] - alphas[:-1])
    return K.ig_accum(grads, x, base, w1=w.reshape(1, steps).contiguous(), w2=substep.to(dev).reshape(1, steps).contiguous())[0]


def IDGI(input, model, steps, batch_size, baseline, device, target_class):
    """IDGI (reference saliencyMethods.py:139-181)."""
    if steps % batch_size != 0:
        print("steps must be evenly divisible by batch size: " + str(batch_size) + "!")
        return 0, 0, 0, 0
    dev, x, base = _prep(input, baseline, device)
    grads, logits = _path(x, base, _uniform_alphas(steps, dev), model, batch_size, target_class)
    g = grads[0]
    return K.idgi_accum(g, logits[0].contiguous(), K.sumsq(g))


_thread_graphs = threading.local()        # per host thread: {key: _CapturedPass} -- a graph is replayed only by the thread that captured it
PASS_COUNTS = {"replayed": 0, "eager": 0, "captures": 0, "captures_refused": 0}     # how ig_batch's passes ran (diagnostics; bench.py prints them)


class _CapturedPass:
    """K1 + classifier forward + backward of `k` images x `steps` interpolants as ONE hipGraph on static buffers.

    Why: with one host thread per stream (streams.py) the passes of a step are enqueued by three Python threads that share one
    interpreter lock -- ~1000 launches and ~100 autograd nodes per pass; the host, not the GPU, becomes the limit (77-84 attributions/s
    eager against 84-85 replayed, ResNet-50).  A replay is one launch.
    Why it is safe: a graph bakes in the pointers of the library workspaces its kernels were captured with, and those belong to the
    capturing THREAD's MIOpen / rocBLAS handles (streams.py).  Each stream worker captures its own graph, on its own handles, with
    autograd inline, and is the only thread that ever replays it -- graphs of different workers share nothing.  (Graphs captured by ONE
    thread and replayed on several streams corrupt each other: profiles/r03_exp_ig_graph_streams*.jsonl.)
    The capture proves itself: its first replay must reproduce the eager pass on the same buffers -- bit for bit with deterministic
    solvers, to the solvers' own run-to-run noise otherwise -- else the pass stays eager."""

    def __init__(self, model, k, steps, img_shape, dev, alphas, base_tensor, base_scalar):
        self.x = torch.zeros((k,) + img_shape, dtype=torch.float32, device=dev)
        self.t = torch.zeros(k, dtype=torch.int64, device=dev)
        self.base = torch.zeros_like(self.x) if base_tensor else None
        self.base_scalar, self.alphas, self.model, self.steps, self.img_shape = base_scalar, alphas, model, steps, img_shape
        cur = torch.cuda.current_stream(dev)
        for _ in range(2):                                    # MIOpen picks its solvers and loads their kernels here, never inside the capture
            eager_g, eager_s = self._run()
        cur.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        try:
            with CAPTURE_LOCK:
                # thread_local: the other stream workers may keep launching and allocating while this thread captures
                with torch.cuda.graph(self.graph, stream=cur, capture_error_mode="thread_local"):
                    self.g, self.scores = self._run()
            self.graph.replay()
            cur.synchronize()
        except Exception:                                     # a classifier that cannot be captured (host syncs in its forward, ...): eager
            self.g = None
        if self.g is None:
            self.ok = False
        elif torch.backends.cudnn.deterministic:
            self.ok = bool(torch.equal(self.g, eager_g) and torch.equal(self.scores, eager_s))
        else:       # MIOpen's non-deterministic solvers differ run to run by themselves (~1e-3 after ReLU-gate flips); a broken replay is off by tens of per cent
            self.ok = bool((self.g - eager_g).abs().max() <= 2e-2 * eager_g.abs().max() and (self.scores - eager_s).abs().max() <= 1e-3 * eager_s.abs().max())
        PASS_COUNTS["captures" if self.ok else "captures_refused"] += 1
        if not self.ok:
            self.graph = self.g = self.scores = None          # give the graph's memory pool back; the pass stays eager

    def _run(self):
        imgs = K.ig_interp(self.x, self.base if self.base is not None else self.base_scalar, self.alphas)
        flat = imgs.view((-1,) + self.img_shape).requires_grad_(True)
        out = _logits_of(self.model(flat))
        scores = out.gather(1, self.t.repeat_interleave(self.steps).unsqueeze(1)).squeeze(1)
        (g,) = torch.autograd.grad(scores, flat, grad_outputs=torch.ones_like(scores))
        return g.contiguous(), scores.detach()

    def __call__(self, x, targets, base):
        self.x.copy_(x, non_blocking=True)
        self.t.copy_(targets, non_blocking=True)
        if self.base is not None:
            self.base.copy_(base, non_blocking=True)
        self.graph.replay()
        return self.g, self.scores


def _thread_pass(model, k, steps, img_shape, dev, alphas, base):
    """This stream worker's graph of a k-image pass (captured on first use), or None when the pass has to stay eager."""
    cache = getattr(_thread_graphs, "passes", None)
    if cache is None:
        cache = _thread_graphs.passes = {}
    key = (id(model), k, steps, img_shape, str(dev), torch.is_tensor(base), None if torch.is_tensor(base) else float(base),
           bool(torch.backends.cudnn.deterministic), bool(torch.backends.cudnn.benchmark))
    if key not in cache:
        if len(cache) >= 4:                                                       # a handful of (model, shape) combinations per thread
            cache.pop(next(iter(cache)))
        cache[key] = _CapturedPass(model, k, steps, img_shape, dev, alphas, torch.is_tensor(base), None if torch.is_tensor(base) else float(base))
    return cache[key] if cache[key].ok else None


def ig_batch(x, model, targets, steps=50, alpha_star=1, baseline=0, images_per_pass=4, want_abs=False,
             grads_buffer=None, event_sink=None, buffered=None, streams=1, graphs=None):
    """Multi-image IG / Left-IG: x (B,C,H,W) on a HIP device, targets (B,) long.
    `images_per_pass` images x `steps` interpolants go through the classifier at once.
    Returns (B,C,H,W) [and the (B,H,W) |sum_c| map the metrics consume].
    Two data flows, bit-identical per element (fp32 sum over ascending steps, / n, * (x - b)):
      buffered   all step gradients land in one [B][steps][C][H][W] buffer that a single K2 launch reduces (per-image Left-IG
                 cutoffs are computed on the device, no host sync).  Needed by Left-IG; it is also the HBM-sized launch
                 bench.py times for the roofline figure (BASELINE: 32 x 50 x 602 KB = 963 MB).
      streaming  alpha_star == 1 only: each pass's gradient is added into a (B,C,H,W) accumulator while it is still cache-resident
                 (xai_ig_accum_add_f32) and xai_ig_finish_f32 scales it -- no buffer, no filing copy, 1/3 of the buffered flow's
                 HBM traffic.
    `buffered`: None = buffered exactly when it has to be (alpha_star != 1) or the caller asked for it by passing
    `grads_buffer` / `event_sink`.
    `streams` > 1: consecutive classifier passes run side by side on that many HIP streams, each driven by its own host thread with
    autograd inline (xai_engine/streams.py: why one thread must not feed two streams on PyTorch-ROCm) -- the low-occupancy layers
    of one pass overlap another pass's work.  Every pass still launches the same kernels on the same shapes and writes disjoint
    rows, so the result is bit-identical to `streams=1`
    (tests/test_gpu_configs.py::test_classifier_passes_on_several_streams_are_bit_identical_to_one_stream).
    `graphs` (with `streams` > 1; default on): every stream worker replays its full-size passes as ONE hipGraph it captured itself
    (`_CapturedPass`: the host stops being the limit once three threads enqueue); a ragged last pass, or a capture whose first replay
    does not reproduce the eager pass bit for bit, runs eagerly.
    `event_sink`: optional list that receives (start, end, kernel_start, kernel_stop) torch.cuda.Events of the
    accumulation launch: a pair bracketing it and a pair stamped by the dispatch itself (used by bench.py for the roofline figure)."""
    if not x.is_cuda:
        raise XaiHipError("ig_batch needs its input on a HIP device")
    x = x.float().contiguous()
    B = x.shape[0]
    dev = x.device
    base = baseline.to(dev, torch.float32).contiguous() if torch.is_tensor(baseline) else float(baseline)
    alphas = _uniform_alphas(steps, dev)
    if buffered is None:
        buffered = alpha_star != 1 or grads_buffer is not None or event_sink is not None
    if not buffered and alpha_star != 1:
        raise ValueError("Left-IG (alpha_star != 1) needs the per-step gradients: buffered=False is for alpha_star == 1 only")
    targets = targets.to(dev).long().reshape(B)
    shape = (B, steps) + tuple(x.shape[1:])
    if buffered:
        if grads_buffer is None:
            grads_buffer = torch.empty(shape, dtype=torch.float32, device=dev)
        elif tuple(grads_buffer.shape) != shape:
            raise ValueError(f"grads_buffer must have shape {shape}")
        logits = torch.empty((B, steps), dtype=torch.float32, device=dev)
    else:
        acc = torch.zeros_like(x)
    img_shape = tuple(x.shape[1:])
    use_graphs = (graphs if graphs is not None else True)

    def one_pass(lo, hi, on_worker=False):
        b = base[lo:hi] if torch.is_tensor(base) else base
        cp = _thread_pass(model, hi - lo, steps, img_shape, dev, alphas, base) if (on_worker and use_graphs and hi - lo == images_per_pass) else None
        PASS_COUNTS["replayed" if cp is not None else "eager"] += 1
        if cp is not None:
            g, scores = cp(x[lo:hi], targets[lo:hi], b if torch.is_tensor(base) else None)
        else:
            imgs = K.ig_interp(x[lo:hi], b, alphas)                              # (k, steps, C,H,W)
            flat = imgs.view((-1,) + img_shape).requires_grad_(True)
            out = _logits_of(model(flat))
            scores = out.gather(1, targets[lo:hi].repeat_interleave(steps).unsqueeze(1)).squeeze(1)
            with backward_turn(dev):
                (g,) = torch.autograd.grad(scores, flat, grad_outputs=torch.ones_like(scores))
            g = g.contiguous()
        if buffered:
            K.store_grads(g, grads_buffer[lo:hi])
            logits[lo:hi] = scores.detach().view(hi - lo, steps)
        else:
            g = g.view((hi - lo, steps) + tuple(x.shape[1:]))
            for j in range(hi - lo):
                K.ig_accum_add(g[j], acc[lo + j])

    spans = [(lo, min(lo + images_per_pass, B)) for lo in range(0, B, images_per_pass)]
    n_streams = 1 if on_worker() else max(1, min(int(streams), len(spans)))      # (called from a stream worker: that thread is the stream)
    if n_streams == 1:
        for lo, hi in spans:
            one_pass(lo, hi)
    else:
        kind = ("ig_batch", id(model), images_per_pass, steps, img_shape, buffered, use_graphs)
        run_on_streams(dev, n_streams, [lambda lo=lo, hi=hi: one_pass(lo, hi, on_worker=True) for lo, hi in spans], kind=kind)
    if not buffered:
        return K.ig_finish(acc, steps, x, base, want_abs=want_abs)
    n_use = None if alpha_star == 1 else K.ig_cutoff(logits, alpha_star)
    if event_sink is None:
        return K.ig_accum(grads_buffer, x, base, n_use=n_use, want_abs=want_abs)
    # bench.py: HIP events for the accumulation launch, on the stream it is launched on -- a pair bracketing the launch
    # (includes the dispatch latency) and a pair the dispatch itself stamps with the kernel's start / stop
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(torch.cuda.current_stream(dev))
    out = K.ig_accum(grads_buffer, x, base, n_use=n_use, want_abs=want_abs, timing_events=(k0, k1))
    t1.record(torch.cuda.current_stream(dev))
    event_sink.append((t0, t1, k0, k1))
    return out
