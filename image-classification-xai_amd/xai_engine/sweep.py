"""Harness counterpart of XAI_Survey/evaluations/evaluatePerturbation.py for the accelerated
methods: attribution dispatch (get_CNN_attr :82-181), the ten perturbation numbers of one image
(run_perturbation :448-497), and the image sweep with its Counter sum / CSV (:499-620), image-
sharded over ranks with one RCCL all-reduce at the end.  The sums are plain sums by default; `reference_counter=True` reproduces
the reference's `Counter +=` fold and CSV loop exactly (keys with a running sum <= 0 vanish, :594-596,:612-615).

`run_perturbation` drives the eight metric objects exactly like the reference.
`PerturbationSweep` computes the same ten numbers from the THREE distinct image sequences the
eight runs contain (insert-blur-descending, delete-zero-descending, delete-zero-ascending;
compare MASTestFunctions.py:137-185, AICTestFunctions.py:94-123, PosNegPertFunctions.py:73-119,
MonotonicityTest.py:93-120): 3 + 3*n_steps classifier passes instead of ~8*(n_steps+3).
"""
import collections
import csv
import os
import threading
import time
from collections import Counter

import numpy as np
import torch
from scipy.stats import spearmanr

from . import curves
from . import kernels as K
from .blur import GaussianBlur
from .gradcam import gradcam_saliency, CapturedGradCam
from .ig import IG, IDG, getGradientsParallel, hip_device, _logits_of
from .perturb import (AICMetric, MASMetric, MonotonicityMetric, PositiveNegativePerturbation, _Probe, sequence_stats)
from .smooth import smoothGrad
from .streams import CAPTURE_LOCK

KEYS = ("MAS_ins", "MAS_del", "RISE_ins", "RISE_del", "AIC_ins", "AIC_del", "LERF_res", "MORF_res", "MONO_pos", "MONO_neg")
CNN_ATTR_FUNCS = ("grad", "inp_x_grad", "ig", "lig", "idg", "sg", "gc")
VIT_ATTR_FUNCS = ("attn", "grad", "n_rollout", "rollout", "t_attn", "bi_attn", "attn_ig", "VIT_CX", "TIS")


def get_CNN_attr(input_tensor, trans_img, target_class, testing_dict):
    """(H,W) float32 numpy saliency map = |sum over channels| of the attribution
    (reference evaluatePerturbation.py:82-181, the methods on the accelerated path).
    testing_dict["device_maps"] = True (not a reference key): return the same map as a device tensor instead -- no
    device-to-host copy and, above all, no host sync between the attribution and the perturbation sweep of an image, so
    `sweep_images` can keep the GPU queue full (the fused sweep takes device maps as they are)."""
    model = testing_dict["models"][0]
    batch_size = testing_dict["batch_size"]
    img_hw = testing_dict["img_hw"]
    device = testing_dict["device"]
    attr_function = testing_dict["attr_func"]
    steps, baseline = 50, 0
    dev = hip_device(device)
    if attr_function == "grad":
        x = input_tensor.to(dev).detach().requires_grad_(True)
        saliency_map, _ = getGradientsParallel(x, model, target_class)
    elif attr_function == "inp_x_grad":
        x = input_tensor.to(dev).detach().requires_grad_(True)
        grad, _ = getGradientsParallel(x, model, target_class)
        saliency_map = x.detach().squeeze() * grad
    elif attr_function == "ig":
        saliency_map = IG(input_tensor, model, steps, batch_size, 1, baseline, device, target_class)
    elif attr_function == "lig":
        saliency_map = IG(input_tensor, model, steps, batch_size, .9, baseline, device, target_class)
    elif attr_function == "idg":
        saliency_map = IDG(input_tensor, model, steps, batch_size, baseline, device, target_class)
    elif attr_function == "sg":
        saliency_map = smoothGrad("IG", input_tensor, model, 50, baseline, target_class, device)
    elif attr_function == "gc":
        # |cam_up + cam_up + cam_up| fused into the up-sample kernel (scale 3, abs); with testing_dict["capture_gradcam"] the
        # launch-bound one-image pass is one hipGraph replay (captured once per model and input shape)
        x = input_tensor.to(dev)
        if testing_dict.get("capture_gradcam"):
            # one graph per stream: a replay reads and writes the graph's static buffers, so two streams must not share one
            key = (id(model), tuple(x.shape), str(dev), torch.cuda.current_stream(dev).cuda_stream)
            cache = testing_dict.setdefault("_captured_gradcam", {})
            if key not in cache:
                cache[key] = CapturedGradCam(model, model.layer4, x, (img_hw, img_hw))
            sal = cache[key](x, target_class)[0]
        else:
            sal = gradcam_saliency(model, model.layer4, x, target_class, (img_hw, img_hw))[0]
        return sal if testing_dict.get("device_maps") else sal.cpu().numpy()
    else:
        print("Model-attribution mismatch, please use --help.")
        raise SystemExit
    if testing_dict.get("device_maps"):
        # (a + b) + c per pixel, the order NumPy's sum over the leading axis uses: bit-identical to the host expression
        m = saliency_map.detach()
        return ((m[0] + m[1]) + m[2]).abs() if m.shape[0] == 3 else m.sum(0).abs()
    return np.abs(np.sum(saliency_map.detach().cpu().numpy(), axis=0))


def get_VIT_attr(input_tensor, trans_img, target_class, testing_dict):
    """(H,W) float32 numpy saliency map of a hooked ViT for the attention-space methods on the accelerated
    path (reference evaluatePerturbation.py:192-370): patch map (1,p,p) -> bilinear up-sample to the image
    (torchvision Resize(antialias=True) == plain bilinear when up-sampling) -> abs, the last two fused in
    xai_bilinear_up_f32.  "attn_ig" (Baselines.IG) is commented out in the reference's table (:276-278) and
    offered here because it is the config-4 method."""
    from .vit_attr import Baselines
    model = testing_dict["models"][0]
    img_hw = testing_dict["img_hw"]
    device = testing_dict["device"]
    attr_function = testing_dict["attr_func"]
    dev = hip_device(device)
    explainer = Baselines(model)
    x = input_tensor.to(dev)
    if attr_function == "attn":
        sal = explainer.generate_raw_attn(x, dev)
    elif attr_function == "grad":
        sal = explainer.generate_grad(x, target_class, dev)
    elif attr_function == "n_rollout":
        sal, _, _ = explainer.generate_naive_rollout(x)
    elif attr_function == "rollout":
        sal, _, _ = explainer.generate_rollout(x)
    elif attr_function == "t_attn":
        _, _, sal, _, _ = explainer.generate_transition_attention_maps(x, target_class, start_layer=0, device=dev)
    elif attr_function == "bi_attn":
        sal, _ = explainer.bidirectional(x, target_class, device=dev)
    elif attr_function == "attn_ig":
        sal = explainer.IG(x, target_class, device=dev)
    elif attr_function == "VIT_CX":
        # :231-235: ViT-CX map of the top-1 class (the reference does not pass target_class), min-max normalised, times
        # ones(H,W,3), then |sum over the 3 copies| = 3 * normalised map.  The reference passes gpu_batch=1; the batch only
        # groups the 2N masked forwards, 50 is ViT_CX's own default.
        from .vit_cx import ViT_CX
        result, _ = ViT_CX(model, x, model.blocks[-1].norm1, device=device, gpu_batch=testing_dict.get("vit_cx_batch", 50),
                           return_feature_map=False)
        result = result.reshape(img_hw, img_hw)
        result = (result - result.min()) / (result.max() - result.min())
        return (result * 3.0).abs().numpy()
    elif attr_function == "TIS":
        from .tis import TIS
        # :236-239 (n_masks is the class default, 1024, in the reference; the key exists for models with fewer channels)
        sal = TIS(model, n_masks=testing_dict.get("tis_n_masks", 1024), batch_size=64)(x, class_idx=target_class)[None]
    else:
        print("Model-attribution mismatch, please use --help.")
        raise SystemExit
    return K.bilinear_up(sal.detach().float().contiguous(), img_hw, img_hw, scale=1.0, take_abs=True)[0].cpu().numpy()


def run_perturbation(input_tensor, attribution, testing_dict, CLIP_test_info=None, blur=None):
    """Eight single_run calls, ten numbers -- call for call the reference's run_perturbation
    (evaluatePerturbation.py:448-497); the blur substrate runs on the device."""
    img_hw = testing_dict["img_hw"]
    step_size = img_hw
    model = testing_dict["models"][0]
    batch_size = testing_dict["batch_size"]
    device = testing_dict["device"]
    HW = img_hw * img_hw
    blur = blur if blur is not None else GaussianBlur(31, 31, device)
    z = torch.zeros_like
    kw = dict(max_batch_size=batch_size, CLIP_test_info=CLIP_test_info)
    _, MAS_ins, _, _, RISE_ins = MASMetric(model, HW, 'ins', step_size, blur).single_run(input_tensor, attribution, device, **kw)
    _, MAS_del, _, _, RISE_del = MASMetric(model, HW, 'del', step_size, z).single_run(input_tensor, attribution, device, **kw)
    _, AIC_ins = AICMetric(model, HW, 'ins', step_size, blur).single_run(input_tensor, attribution, device, **kw)
    _, AIC_del = AICMetric(model, HW, 'del', step_size, z).single_run(input_tensor, attribution, device, **kw)
    _, LERF_res = PositiveNegativePerturbation(model, HW, 'lerf', step_size, z).single_run(input_tensor, attribution, device, **kw)
    _, MORF_res = PositiveNegativePerturbation(model, HW, 'morf', step_size, z).single_run(input_tensor, attribution, device, **kw)
    _, MONO_pos = MonotonicityMetric(model, HW, 'positive', step_size, blur).single_run(input_tensor, attribution, device, **kw)
    _, MONO_neg = MonotonicityMetric(model, HW, 'negative', step_size, z).single_run(input_tensor, attribution, device, **kw)
    auc = curves.auc
    return Counter({"MAS_ins": auc(MAS_ins), "MAS_del": auc(MAS_del), "RISE_ins": auc(RISE_ins), "RISE_del": auc(RISE_del),
                    "AIC_ins": auc(AIC_ins), "AIC_del": auc(AIC_del), "LERF_res": auc(LERF_res), "MORF_res": auc(MORF_res),
                    "MONO_pos": MONO_pos, "MONO_neg": MONO_neg})


_thread_forwards = threading.local()     # per host thread: {key: _CapturedForward}
FORWARD_COUNTS = {"replayed": 0, "eager": 0, "captures": 0, "captures_refused": 0}


class _CapturedForward:
    """The classifier's forward pass for a batch of b step images as ONE hipGraph on a static input buffer, captured and replayed by
    one stream worker only (same reasoning as ig._CapturedPass: with one host thread per stream the ~250 launches of every forward
    batch are enqueued under one interpreter lock; a replay is one launch, and a graph captured on the worker's own library
    handles shares nothing with the other workers' graphs).  Its first replay must reproduce the eager forward (bit for bit with
    deterministic solvers)."""

    def __init__(self, model, b, img_shape, dev):
        self.x = torch.zeros((b,) + tuple(img_shape), dtype=torch.float32, device=dev)
        cur = torch.cuda.current_stream(dev)
        with torch.no_grad():
            for _ in range(2):
                eager = _logits_of(model(self.x)).detach().clone()
            cur.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            self.logits = None
            try:
                with CAPTURE_LOCK:
                    with torch.cuda.graph(self.graph, stream=cur, capture_error_mode="thread_local"):
                        self.logits = _logits_of(model(self.x))
                self.graph.replay()
                cur.synchronize()
            except Exception:                                  # a classifier whose forward cannot be captured: eager
                self.logits = None
        if self.logits is None:
            self.ok = False
        elif torch.backends.cudnn.deterministic:
            self.ok = bool(torch.equal(self.logits, eager))
        else:                                                  # non-deterministic solvers: to their own run-to-run noise
            self.ok = bool((self.logits - eager).abs().max() <= 1e-3 * eager.abs().max())
        FORWARD_COUNTS["captures" if self.ok else "captures_refused"] += 1
        if not self.ok:
            self.graph = self.logits = None


class PerturbationSweep:
    """The same ten numbers from three device-resident sequences (see module docstring)."""

    def __init__(self, model, img_hw, device, step_size=None, batch_size=50, klen=31, ksig=31):
        self.model = model
        self.dev = hip_device(device)
        self.img_hw = img_hw
        self.HW = img_hw * img_hw
        self.step_size = step_size or img_hw
        self.batch_size = batch_size
        self.blur = GaussianBlur(klen, ksig, self.dev)

    def _stats(self, images, target, out=None, offset=0):
        with torch.no_grad():
            return _Probe(_logits_of(self.model(images)).detach(), target, out, offset)

    def _captured(self, b, img_shape):
        """this thread's graph of a b-image forward, or None; only on stream workers (`graphs` is set by sweep_images)"""
        if not getattr(_thread_forwards, "enabled", False):
            return None
        cache = getattr(_thread_forwards, "graphs", None)
        if cache is None:
            cache = _thread_forwards.graphs = {}
        key = (id(self.model), b, tuple(img_shape), str(self.dev), bool(torch.backends.cudnn.deterministic), bool(torch.backends.cudnn.benchmark))
        if key not in cache:
            if len(cache) >= 6:
                cache.pop(next(iter(cache)))
            cache[key] = _CapturedForward(self.model, b, img_shape, self.dev)
        return cache[key] if cache[key].ok else None

    def launch(self, input_tensor, attribution):
        """Queue the whole device part of one image (probes, ranking, three sequences) and an asynchronous
        copy of the curves into pinned host memory; returns a handle for `finish`.  Nothing here waits for the
        GPU, so the next image can be queued (or the previous one finished) while this one runs.
        `attribution`: (H,W) float32 NumPy array or device tensor (a device tensor avoids the upload)."""
        dev = self.dev
        n_steps, step, batches = curves.step_plan(self.HW, self.step_size, self.batch_size)
        if input_tensor.is_cuda:
            img = input_tensor.to(dev, torch.float32).contiguous()
        else:
            img = input_tensor.to(torch.float32).contiguous().pin_memory().to(dev, non_blocking=True)
        blurred = self.blur(img)
        zeros = torch.zeros_like(img)
        orig = self._stats(img, None)
        target = orig.argmax
        pb = self._stats(blurred, target)
        pz = self._stats(zeros, target)
        if torch.is_tensor(attribution) and attribution.is_cuda:
            sal = attribution.to(dev, torch.float32).reshape(1, self.HW).contiguous()
        else:
            host_sal = torch.as_tensor(np.ascontiguousarray(attribution, dtype=np.float32)).reshape(1, self.HW)
            sal = host_sal.pin_memory().to(dev, non_blocking=True)
        order, rk = K.rank(sal)
        f_desc = K.flip_steps(rk[0], True, step)
        f_asc = K.flip_steps(rk[0], False, step)
        seg_d, total = K.segment_sums(sal[0], order[0], True, step, n_steps)
        img_shape = tuple(img.shape[1:])

        def slot_for(b):
            cf = self._captured(b, img_shape)
            return None if cf is None else cf.x

        def stats(images, tgt, out=None, offset=0):
            cf = self._captured(images.shape[0], img_shape)
            if cf is None or images.data_ptr() != cf.x.data_ptr():
                FORWARD_COUNTS["eager"] += 1
                return self._stats(images, tgt, out, offset)
            FORWARD_COUNTS["replayed"] += 1
            cf.graph.replay()                                  # the step images are in cf.x already (K6 wrote them there)
            return _Probe(cf.logits, tgt, out, offset)

        ins = sequence_stats(stats, blurred[0], img[0], f_desc, n_steps, batches, target, pb, slot_for)
        dele = sequence_stats(stats, img[0], zeros[0], f_desc, n_steps, batches, target, orig, slot_for)
        lerf = sequence_stats(stats, img[0], zeros[0], f_asc, n_steps, batches, target, orig, slot_for)
        packed = torch.cat([ins[0], ins[2].float(), dele[0], dele[2].float(), lerf[0], seg_d, total,
                            orig.p, pb.p, pz.p, pb.argmax.float(), pz.argmax.float(), target.float()])
        host = torch.empty(packed.shape, dtype=packed.dtype, pin_memory=True)
        host.copy_(packed, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(dev))
        return (n_steps, host, done, packed)          # `packed` is kept alive until the copy has happened

    def finish(self, handle, return_curves=False):
        """Wait for `launch`'s copy and do the 225-point curve arithmetic on the host."""
        n_steps, host_t, done, _ = handle
        done.synchronize()
        host = host_t.numpy().astype(np.float64)
        n1 = n_steps + 1
        r_ins, a_ins, r_del, a_del, r_lerf = (host[i * n1:(i + 1) * n1] for i in range(5))
        seg = host[5 * n1:5 * n1 + n_steps].astype(np.float32)
        tot = np.float32(host[5 * n1 + n_steps])
        o_p, b_p, z_p, b_cls, z_cls, tgt = host[5 * n1 + n_steps + 1:]

        norm_ins = curves.monotone_normalise(r_ins, b_p, o_p, falling=False)
        norm_del = curves.monotone_normalise(r_del, z_p, o_p, falling=True)
        mas_ins = curves.mas_correct(norm_ins, curves.density_curve(seg, tot, True), 'ins')
        mas_del = curves.mas_correct(norm_del, curves.density_curve(seg, tot, False), 'del')
        aic_i = (a_ins == tgt).astype(np.float64)
        aic_d = (a_del == tgt).astype(np.float64)
        aic_i[0], aic_d[0] = float(b_cls == tgt), 1.0
        with np.errstate(divide="ignore", invalid="ignore"):
            aic_ins = curves.monotone_normalise(aic_i, float(b_cls == tgt), 1, falling=False)
            aic_del = curves.monotone_normalise(aic_d, float(z_cls == tgt), 1, falling=True)
        auc = curves.auc
        out = Counter({"MAS_ins": auc(mas_ins), "MAS_del": auc(mas_del), "RISE_ins": auc(norm_ins), "RISE_del": auc(norm_del),
                       "AIC_ins": auc(aic_ins), "AIC_del": auc(aic_del), "LERF_res": auc(r_lerf), "MORF_res": auc(r_del),
                       "MONO_pos": spearmanr(np.linspace(0, 1, n1), r_ins).correlation,
                       "MONO_neg": spearmanr(np.linspace(1, 0, n1), r_del).correlation})
        if return_curves:
            return out, dict(ins=r_ins, dele=r_del, lerf=r_lerf, mas_ins=mas_ins, mas_del=mas_del)
        return out

    def run(self, input_tensor, attribution, return_curves=False):
        return self.finish(self.launch(input_tensor, attribution), return_curves)


# ------------------------------------------------------------------------------ image sweep, sharded over ranks
def shard_indices(n_items, rank, world):
    """Round-robin ownership: item i belongs to rank i % world (deterministic, order-preserving)."""
    return list(range(rank, n_items, world))


def reduce_counters(local_sum, n_local, device=None, group=None, world=None, attr_seconds=None):
    """One all-reduce(SUM) of [10 metric sums, images used, seconds in attribution] in fp64 (96 bytes) -> global Counter, count
    (and the summed seconds when `attr_seconds` is given).  `world=1` (a caller that did not shard) skips the collective even
    inside an initialised job."""
    import torch.distributed as dist
    vec = torch.tensor([float(local_sum.get(k, 0.0)) for k in KEYS] + [float(n_local), float(attr_seconds or 0.0)], dtype=torch.float64)
    if world != 1 and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "nccl":
            vec = vec.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
        vec = vec.cpu()
    out = (Counter({k: float(vec[i]) for i, k in enumerate(KEYS)}), int(round(float(vec[len(KEYS)]))))
    return out if attr_seconds is None else out + (float(vec[len(KEYS) + 1]),)


def replay_reference_counter(rows):
    """The reference's running Counter over per-image results in FILE ORDER (evaluatePerturbation.py:593-596): the first image's
    Counter is taken as it is, every later one is folded in with `+=` -- collections.Counter.__iadd__ adds and then deletes every
    key whose running sum is not > 0 (a zero AIC sum, a negative Spearman sum, a NaN), and a deleted key that comes back starts from
    nothing and sits at the end.  rows: iterable of 10-vectors in KEYS order.  Uses the standard library's Counter itself, so the
    semantics are the reference's by construction (pinned by tests/golden/sweep_counter.npz)."""
    total = None
    for r in rows:
        c = Counter({k: float(v) for k, v in zip(KEYS, r)})
        if total is None:
            total = c
        else:
            total += c
    return total if total is not None else Counter()


def gather_rows(local_rows, n_items, device=None, group=None, world=None, attr_seconds=0.0):
    """All ranks' per-image 10-vectors as one (n_items, 10) fp64 matrix plus a used-flag column, by ONE all-reduce(SUM) of the
    zero-padded (n_items + 1, 11) matrix (every image is owned by exactly one rank, so the rows are disjoint; 88 B x images --
    88 KB for the 1000-image sweep; the extra row carries the seconds in attribution).  local_rows: {global index: 10-vector}.
    -> (rows (n_items, 10) float64, used (n_items,) bool, summed attribution seconds)."""
    import torch.distributed as dist
    mat = torch.zeros((n_items + 1, len(KEYS) + 1), dtype=torch.float64)
    for i, r in local_rows.items():
        mat[int(i), :len(KEYS)] = torch.as_tensor(np.asarray(r, dtype=np.float64))
        mat[int(i), len(KEYS)] = 1.0
    mat[n_items, 0] = float(attr_seconds)
    if world != 1 and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "nccl":
            mat = mat.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        dist.all_reduce(mat, op=dist.ReduceOp.SUM, group=group)
        mat = mat.cpu()
    m = mat.numpy()
    return m[:n_items, :len(KEYS)].copy(), m[:n_items, len(KEYS)] > 0.5, float(m[n_items, 0])


def sweep_identity(**facts):
    """Stable string describing WHAT a checkpointed sweep computes (attribution method, model, weights, flow, image
    list ...): the facts as sorted `key=value` pairs, long values hashed.  A checkpoint resumes only under the same string."""
    import hashlib

    def short(v):
        v = str(v)
        return v if len(v) <= 40 else "sha256:" + hashlib.sha256(v.encode()).hexdigest()[:24]
    return ";".join(f"{k}={short(facts[k])}" for k in sorted(facts))


def model_fingerprint(model, max_tensors=8):
    """Cheap fingerprint of a classifier's weights: class name, parameter count and the sums of its first and last
    few tensors (enough to tell two checkpoints of one architecture apart without hashing 100 MB)."""
    tensors = list(model.state_dict().items())
    picked = tensors[:max_tensors // 2] + tensors[-(max_tensors // 2):]
    parts = [type(model).__name__, str(sum(int(t.numel()) for _, t in tensors))]
    parts += [f"{k}:{float(t.double().sum()):.10e}" for k, t in picked]
    return "|".join(parts)


class CheckpointMismatch(RuntimeError):
    pass


class SweepState:
    """Per-rank running sums of a sharded sweep, checkpointed so that a crash does not lose the run (the
    reference writes its CSV once at the very end, evaluatePerturbation.py:612).  One small JSON file per
    rank, replaced atomically.  It is only valid for the same (n_items, rank, world) split AND the same
    `identity` (see sweep_identity): resuming somebody else's sums into this run's CSV is refused loudly."""

    def __init__(self, n_items, rank, world, identity=""):
        self.n_items, self.rank, self.world, self.identity = n_items, rank, world, identity
        self.sums = {k: 0.0 for k in KEYS}
        self.used = 0            # images folded into `sums`
        self.next_pos = 0        # position in this rank's shard list
        self.attr_time = 0.0
        self.rows = {}           # reference_counter mode: {global image index: that image's ten numbers}

    @staticmethod
    def path_for(prefix, rank, world):
        return f"{prefix}.rank{rank}of{world}.json"

    def save(self, prefix):
        import json
        path = self.path_for(prefix, self.rank, self.world)
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(dict(n_items=self.n_items, rank=self.rank, world=self.world, identity=self.identity, sums=self.sums,
                           used=self.used, next_pos=self.next_pos, attr_time=self.attr_time,
                           rows={str(i): [float(v) for v in r] for i, r in self.rows.items()}), f)
        os.replace(tmp, path)

    @classmethod
    def load_or_new(cls, prefix, n_items, rank, world, identity=""):
        import json
        st = cls(n_items, rank, world, identity)
        path = cls.path_for(prefix, rank, world) if prefix else None
        if path and os.path.exists(path):
            d = json.load(open(path))
            split = (d.get("n_items"), d.get("rank"), d.get("world"))
            if split != (n_items, rank, world) or d.get("identity", "") != identity or set(d.get("sums", {})) != set(KEYS):
                raise CheckpointMismatch(
                    f"{path} belongs to a different sweep and is not resumed: it holds (n_items, rank, world) = {split}, "
                    f"identity '{d.get('identity', '')}'; this run is {(n_items, rank, world)}, identity '{identity}'. "
                    "Use another --checkpoint prefix or delete the file.")
            st.sums = {k: float(d["sums"][k]) for k in KEYS}
            st.used, st.next_pos, st.attr_time = int(d["used"]), int(d["next_pos"]), float(d["attr_time"])
            st.rows = {int(i): [float(v) for v in r] for i, r in d.get("rows", {}).items()}
        return st


def sweep_images(images, model, device, attr_fn, img_hw=224, batch_size=50, fused=True, rank=0, world=1, testing_dict=None,
                 checkpoint=None, checkpoint_every=25, identity=None, streams=1, reference_counter=False, kind=None, graphs=True):
    """Attribution + ten perturbation numbers for every image this rank owns; returns the
    globally reduced (Counter of sums, images used, seconds in attribution).
    images: sequence of (1,C,H,W) CPU/device tensors (already selected -- the order-dependent
    filters of evaluatePerturbation.py:520-576 must run as a deterministic pre-pass so that the
    1-GPU and N-GPU runs see the same list).  attr_fn(x, target) -> (H,W) float32 numpy map.
    `checkpoint`: path prefix; every `checkpoint_every` images the rank's running sums are saved and an
    interrupted sweep with the same split and the same `identity` resumes after the last saved image; a checkpoint
    of another split or identity raises CheckpointMismatch.  `identity`: string from `sweep_identity(...)` naming
    what the caller's attr_fn / image list are (the harness passes attr_func, model name, file-name hash, ...); the
    flow (fused / eight runs), geometry and a fingerprint of the classifier's weights are always added here.
    `reference_counter`: fold the per-image results the way the reference does (`replay_reference_counter`: Counter `+=` in file
    order, keys with a running sum <= 0 dropped) instead of plain sums.  The fold is order-dependent, so every rank keeps its
    images' ten numbers, ONE all-reduce of the zero-padded (images, 11) matrix hands every rank all of them (`gather_rows`), and
    the replay runs over them in file order -- the same Counter for every world size.  The Counter returned then holds only
    the surviving keys, in the reference's order; `write_csv(..., reference_counter=True)` writes exactly those rows.
    `streams` > 1 (fused flow): consecutive images are queued round-robin on that many HIP streams, each driven by its own host
    thread (xai_engine/streams.py; `attr_fn` and the image access then run on those threads, in no particular order) --
    attribution, ranking and the three step sequences of one image form a serial chain of mostly small or low-occupancy launches,
    so the chains of `streams` images overlap on the chip.  Only for classifiers whose forward keeps no per-pass state ON THE MODEL:
    the hooked ViT saves attention maps / gradients / block outputs on its modules and TIS / ViT-CX hang hooks on it, so concurrent
    passes through one such model would read each other's tensors -- use `streams=1` there (the harness does).
    `graphs` (with `streams` > 1): every stream worker replays the forward passes of its step batches as hipGraphs it captured itself
    (`_CapturedForward`; three threads enqueueing ~250 launches per batch under one interpreter lock are otherwise the limit).
    Every image still runs the same kernels on the same shapes and the per-image Counters are folded in image order, so the sums are
    bit-identical to `streams=1` (tests/test_gpu_configs.py::test_classifier_passes_on_several_streams_are_bit_identical_to_one_stream).
    The third return value, seconds in attribution, is measured with HIP events on the image's stream when the map stays on the
    device (the reference times a finished attribution, evaluatePerturbation.py:581-590; the host clock around an asynchronous
    launch would only see the enqueue)."""
    dev = hip_device(device)
    sweep = PerturbationSweep(model, img_hw, dev, batch_size=batch_size) if fused else None
    td = testing_dict or {"models": [model], "img_hw": img_hw, "batch_size": batch_size, "device": str(dev)}
    blur = GaussianBlur(31, 31, dev)
    mine = shard_indices(len(images), rank, world)
    ident = ""
    if checkpoint:
        ident = sweep_identity(caller=identity or "", fused=fused, img_hw=img_hw, batch_size=batch_size, model=model_fingerprint(model))
    st = SweepState.load_or_new(checkpoint, len(images), rank, world, ident)

    def fold(c, pos):
        for k in KEYS:                                   # plain sums: see DESIGN.md on the reference's Counter `+=`
            st.sums[k] += float(c[k])
        if reference_counter:
            st.rows[mine[pos]] = [float(c[k]) for k in KEYS]
        st.used += 1
        st.next_pos = pos + 1
        if checkpoint and (st.next_pos % checkpoint_every == 0 or st.next_pos == len(mine)):
            st.save(checkpoint)

    from .streams import on_worker
    n_streams = max(1, int(streams)) if (fused and not on_worker()) else 1      # (called from a stream worker: this thread is the stream)
    ws = None
    if n_streams > 1:
        from .streams import workers, join, first_alone
        ws = workers(dev, n_streams)                     # one host thread per stream (streams.py)
        # the first image of a kind of sweep on each worker runs alone (streams.first_alone); `kind` names what attr_fn does (the
        # harness passes the method name), by default the function object itself
        kind = ("sweep", id(model), kind if kind is not None else id(attr_fn), img_hw, batch_size, bool(graphs))
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))     # weights, blur taps: whatever the caller's stream has queued so far
    pending = collections.deque()                        # (future or result, pos) of the images whose device work is in flight
    failed = None

    def device_part(pos):
        """Everything of one image that runs on the device, queued on the current stream of the calling thread:
        -> (handle for `finish` or the finished Counter, attribution timer or None)."""
        x = images[mine[pos]]
        if ws is not None:
            torch.cuda.current_stream(dev).wait_event(ready)
            _thread_forwards.enabled = graphs
        if fused and not x.is_cuda:
            # one upload through pinned memory, queued behind the previous image's work: a pageable .to(dev) blocks the
            # host until the stream has drained, which would undo the pipelining
            x = x.to(torch.float32).contiguous().pin_memory().to(dev, non_blocking=True)
        with torch.no_grad():
            target = _logits_of(model(x.to(dev))).argmax(1)[0]
        timer = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        t0 = time.time()
        timer[0].record(torch.cuda.current_stream(dev))
        sal = attr_fn(x, target)
        timer[1].record(torch.cuda.current_stream(dev))
        host_seconds = None
        if not (torch.is_tensor(sal) and sal.is_cuda):
            host_seconds = time.time() - t0              # a host map: attr_fn has waited for the device itself
            timer = None
        if fused:
            return sweep.launch(x, sal), timer, host_seconds
        if timer is not None:
            timer[1].synchronize()
            host_seconds, timer = timer[0].elapsed_time(timer[1]) * 1e-3, None
        return run_perturbation(x.cpu(), sal, td, blur=blur), timer, host_seconds

    def finish_oldest():
        job, pos = pending.popleft()
        out, timer, host_seconds = job.result() if ws is not None else job
        c = sweep.finish(out) if fused else out
        if timer is not None:                            # the events lie before `done` on the same stream: complete by now
            st.attr_time += timer[0].elapsed_time(timer[1]) * 1e-3
        if host_seconds is not None:
            st.attr_time += host_seconds
        fold(c, pos)

    try:
        for pos in range(st.next_pos, len(mine)):
            # `n_streams` images deep: queue this image's device work (on its stream's worker thread), then do the oldest image's
            # host arithmetic while the device runs
            pending.append((first_alone(ws[pos % n_streams], kind, lambda pos=pos: device_part(pos)) if ws is not None else device_part(pos), pos))
            while len(pending) > (n_streams if fused else 0):
                finish_oldest()
    except BaseException as e:
        failed = e
        raise
    finally:
        while pending:                                   # also on an exception: the queued images are complete work, keep them
            try:
                finish_oldest()
            except Exception:
                if failed is None:                       # nothing else went wrong: this IS the error
                    raise
                break                                    # a device error already in flight makes the fold fail too: report the original
        if ws is not None:
            join(dev, ws)
    if reference_counter:
        if len(st.rows) != st.used:
            raise CheckpointMismatch("this checkpoint was written without reference_counter: it holds sums only, not the per-image "
                                     "numbers the reference's order-dependent fold needs; start again with another --checkpoint prefix")
        rows, flags, attr_seconds = gather_rows(st.rows, len(images), dev, world=world, attr_seconds=st.attr_time)
        return replay_reference_counter(rows[flags]), int(flags.sum()), attr_seconds
    return reduce_counters(st.sums, st.used, dev, world=world, attr_seconds=st.attr_time)


def write_csv(path, counter_sum, images_used, attr_time, total_time, reference_counter=False):
    """rows `key,mean` for the metrics + the two runtime rows (reference :606-618).
    Default: always the ten keys in KEYS order, plain sums / images, and a last row `Fold,plain sums` that says so.
    `reference_counter=True`: `counter_sum` is the Counter of `sweep_images(..., reference_counter=True)` and the file is the reference's,
    row for row (:612-618) -- only the keys that survived its `+=`, in the Counter's own order, then the two runtime rows."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    with open(path, "w") as f:
        w = csv.writer(f)
        for k in (counter_sum if reference_counter else KEYS):
            w.writerow([k, str(counter_sum[k] / images_used)])
        w.writerow(["Attr Avg Runtime", str(attr_time / images_used)])
        w.writerow(["Total Runtime", str(total_time)])
        if not reference_counter:                        # (nothing in the reference reads these files back; the row tells the two folds apart)
            w.writerow(["Fold", "plain sums"])
