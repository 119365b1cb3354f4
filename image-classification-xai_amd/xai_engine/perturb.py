"""Device-resident insertion/deletion loop and the five metric classes built on it.

Where the reference edits a NumPy view pixel by pixel on the host and ships every batch over
PCIe (MASTestFunctions.py:245-281 and the same loop in RISE/AIC/PosNegPert/Monotonicity),
this keeps start, finish and the per-pixel flip step on the GPU: K8 ranks the saliency map
once, K6 materialises each batch of step images straight into the classifier's input buffer,
K9 reduces the logits to (p[target], entropy, argmax) and only 3 x (n_steps+1) floats come
back.  The 225-point curve arithmetic stays NumPy float64 on the host (curves.py).

Class names, constructor/single_run signatures, mode strings, return tuples and the
assert / print-and-return-zeros conventions are the reference's.
"""
import numpy as np
import torch
from scipy.stats import spearmanr

from . import curves
from . import kernels as K
from .ig import hip_device, _logits_of


class _Probe:
    """softmax statistics of one forward pass, kept on the device."""

    def __init__(self, logits, target=None, out=None, offset=0):
        self.p, self.entropy, self.argmax = K.softmax_stats(logits.float().contiguous(), target, out=out, offset=offset)


def sequence_stats(stats, start, finish, flip, n_steps, batches, target, first, slot_for=None):
    """Run one insertion/deletion sequence on the device.  `stats(images, target, out, offset) -> _Probe`
    writes its rows straight into the curves; start/finish (C,H,W); flip (H*W,) int32; `first` = the _Probe
    of curve point 0.  Returns device tensors p[target], entropy, argmax of length n_steps + 1.
    `slot_for(b)`: optional; the static input buffer (b,C,H,W) of a captured forward pass for batches of b images, or None --
    K6 then writes the step images straight into it (sweep.PerturbationSweep replays the classifier forward as a hipGraph)."""
    dev = start.device
    p = torch.empty(n_steps + 1, dtype=torch.float32, device=dev)
    ent = torch.empty(n_steps + 1, dtype=torch.float32, device=dev)
    am = torch.empty(n_steps + 1, dtype=torch.int32, device=dev)
    p[0:1], ent[0:1], am[0:1] = first.p, first.entropy, first.argmax
    buf = torch.empty((max(batches) if batches else 0,) + tuple(start.shape), dtype=torch.float32, device=dev)
    done = 0
    for b in batches:
        if b == 0:                                      # MonotonicityTest's empty remainder batch: nothing to add
            continue
        slot = slot_for(b) if slot_for is not None else None
        images = K.perturb_batch(start, finish, flip, done, b, out=slot if slot is not None else buf[:b])
        stats(images, target, (p, ent, am), 1 + done)
        done += b
    return p, ent, am


class _PerturbationMetric:
    MODES = ()
    ALWAYS_LEFTOVER = False

    def __init__(self, model, HW, mode, step_size, substrate_fn):
        assert mode in self.MODES
        self.model = model
        self.HW = HW
        self.mode = mode
        self.step_size = step_size
        self.substrate_fn = substrate_fn

    # ---- what differs between metrics -------------------------------------------------
    def _inserting(self):
        return self.mode in ("ins", "positive")

    def _descending(self):
        return self.mode != "lerf"

    # ---- classifier access --------------------------------------------------------------
    def _logits(self, images, clip_info):
        with torch.no_grad():
            if clip_info is None:
                return _logits_of(self.model(images)).detach()
            emb = clip_info["embeddings"]
            return (self.model.encode_image(images) @ emb.squeeze().T).detach()

    # ---- the shared device pipeline ------------------------------------------------------
    def _run(self, img_tensor, saliency_map, device, patch_mask, max_batch_size, clip_info=None, want_density=False,
             want_embeddings=False, given_order=None):
        dev = hip_device(device)
        n_steps, step_size, batches = curves.step_plan(self.HW, self.step_size, max_batch_size, patch_mask, self.ALWAYS_LEFTOVER)
        if patch_mask is not None:
            self.step_size = step_size                      # the reference overwrites it too (:92)
        temp = 0.1 if clip_info is not None else None       # CLIP similarities are softmaxed at T = 0.1

        # return_embeddings (MASTestFunctions.py:121-133,283-296): after the pass over the original image and after every
        # step batch, the block outputs the hooked ViT retained (`block.get_block_out()`) and the arg-max classes are kept --
        # on the device, one (num_blocks, batch, tokens, dim) tensor per pass; the substrate probe in between is not recorded
        kept = {"on": False, "emb": [], "cls": []}

        def stats(images, target, out=None, offset=0):
            lg = self._logits(images, clip_info)
            if kept["on"]:
                kept["emb"].append(torch.stack([blk.get_block_out().detach() for blk in self.model.blocks]))
                kept["cls"].append(lg.argmax(1))
            return _Probe(lg / temp if temp else lg, target, out, offset)

        img = img_tensor.to(dev, torch.float32).contiguous()
        substrate = self.substrate_fn(img_tensor).to(dev, torch.float32).contiguous()
        kept["on"] = want_embeddings
        if clip_info is None:
            orig = stats(img, None)
        else:
            orig = stats(clip_info["input"].to(dev), None)
        target = orig.argmax                                 # int32 (1,) on the device, never synced
        kept["on"] = False
        sub = stats(substrate, target)
        kept["on"] = want_embeddings
        start, finish = (substrate, img) if self._inserting() else (img, substrate)

        # pixel order -> flip step per pixel
        seg = total = None
        salient_order = None
        if given_order is not None:
            # caller-supplied flip order (keyword-only `salient_order=` of single_run; not a reference argument): the pixel --
            # or, with patch_mask, patch -- indices in the order they flip.  The reference ranks with NumPy's default, unstable
            # argsort, so on tied maps (ReLU'd Grad-CAM) its order is machine-dependent; handing that order in reproduces its
            # curves exactly, where this build's own rule is the stable sort.
            go = np.ascontiguousarray(np.asarray(given_order).reshape(-1), dtype=np.int64)
            n_units = self.HW if patch_mask is None else n_steps
            if go.shape[0] != n_units or not np.array_equal(np.sort(go), np.arange(n_units)):
                raise ValueError(f"salient_order must be a permutation of range({n_units})")
            salient_order = go.reshape(1, -1) if patch_mask is None else go
            if patch_mask is None:
                flip_np = np.empty(self.HW, dtype=np.int32)
                flip_np[go] = (np.arange(self.HW) // step_size).astype(np.int32)
                flip = torch.from_numpy(flip_np).to(dev)
                if want_density:
                    sal = torch.as_tensor(np.ascontiguousarray(saliency_map, dtype=np.float32)).reshape(1, self.HW).to(dev)
                    seg, total = K.segment_sums(sal[0], torch.from_numpy(go.astype(np.int32)).to(dev), False, step_size, n_steps)
            else:
                pm = np.asarray(patch_mask.cpu() if hasattr(patch_mask, "cpu") else patch_mask).reshape(-1)
                step_of_patch = np.empty(n_steps, dtype=np.int32)
                step_of_patch[go] = np.arange(n_steps, dtype=np.int32)
                flip_np = step_of_patch[pm].astype(np.int32)
                flip = torch.from_numpy(flip_np).to(dev)
                if want_density:
                    seg, total = curves.patch_density_sums(saliency_map, flip_np, self.HW, n_steps)
        elif patch_mask is None:
            sal = torch.as_tensor(np.ascontiguousarray(saliency_map, dtype=np.float32)).reshape(1, self.HW).to(dev)
            order, rk = K.rank(sal)
            flip = K.flip_steps(rk[0], self._descending(), step_size)
            if want_embeddings:                              # (1, HW) like np.flip(np.argsort(...)) / np.argsort(...) (:209-212)
                asc = order.cpu().numpy().astype(np.int64)
                salient_order = asc[:, ::-1].copy() if self._descending() else asc
            if want_density:
                seg, total = K.segment_sums(sal[0], order[0], self._descending(), step_size, n_steps)
        else:
            flip_np, salient_order = curves.patch_flip_steps(saliency_map, patch_mask, self.HW, n_steps, self._descending())
            flip = torch.from_numpy(flip_np).to(dev)
            if want_density:
                seg, total = curves.patch_density_sums(saliency_map, flip_np, self.HW, n_steps)

        # every step image through the classifier, in the reference's batch sizes
        first = sub if self._inserting() else orig
        p, ent, am = sequence_stats(stats, start[0], finish[0], flip, n_steps, batches, target, first)

        # one device->host transfer for everything the host arithmetic needs
        host = torch.cat([p, ent, am.float(), orig.p, sub.p, sub.argmax.float(), target.float()]).cpu().numpy()
        n1 = n_steps + 1
        out = dict(n_steps=n_steps, response=host[:n1].astype(np.float64), entropy=host[n1:2 * n1].astype(np.float64),
                   argmax=host[2 * n1:3 * n1].astype(np.int64), original_pred=float(host[3 * n1]),
                   baseline_pred=float(host[3 * n1 + 1]), baseline_class=int(host[3 * n1 + 2]), target=int(host[3 * n1 + 3]))
        if want_density:
            if torch.is_tensor(seg):
                seg, total = seg.cpu().numpy(), total.cpu().numpy()[0]
            out["density"] = curves.density_curve(seg, total, self._inserting())
        if want_embeddings:
            emb, cls = kept["emb"], kept["cls"]              # [original, batch 1, batch 2, ...]
            if self.mode == "ins":                           # the reference appends the original image's entry LAST for 'ins' (:375-377)
                emb, cls = emb[1:] + emb[:1], cls[1:] + cls[:1]
            out["embeddings"] = torch.cat(emb, dim=1).cpu().numpy()
            out["classes"] = torch.cat(cls, dim=0).cpu().numpy()
            out["salient_order"] = salient_order
        return out

    def _embeddings_tuple(self, r):
        """(embeddings (num_blocks, n_steps + 1, tokens, dim), classes (n_steps + 1,), raw model response, salient order)
        -- the return_embeddings=True result of MASMetric / RISEMetric (MASTestFunctions.py:370-381, RISETestFunctions.py:223-234)"""
        return r["embeddings"], r["classes"], r["response"], r["salient_order"]


class MASMetric(_PerturbationMetric):
    """reference util/test_methods/MASTestFunctions.py:55-385"""
    MODES = ('del', 'ins', 'lerf', 'morf')

    def single_run(self, img_tensor, saliency_map, device, patch_mask=None, max_batch_size=50, special_version=False,
                   return_embeddings=False, CLIP_test_info=None, *, salient_order=None):
        if return_embeddings and CLIP_test_info is not None:
            raise NotImplementedError("return_embeddings reads model.blocks[i].get_block_out(): hooked ViT classifiers only, as in the reference")
        r = self._run(img_tensor, saliency_map, device, patch_mask, max_batch_size, CLIP_test_info, want_density=True,
                      want_embeddings=return_embeddings, given_order=salient_order)
        if return_embeddings:
            return self._embeddings_tuple(r)
        if CLIP_test_info is not None:
            r["entropy"] = np.ones(r["n_steps"] + 1)        # the reference's CLIP branch never fills it (:143-159,:277-281)
        norm = curves.monotone_normalise(r["response"], r["baseline_pred"], r["original_pred"], falling=(self.mode != 'ins'))
        if special_version:
            # the convex ('del') / concave ('ins') least-squares smoothing of the normalised response (:311-350), solved exactly on the
            # host instead of by cvxopt's interior-point iteration (curves.shape_constrained_fit; parity unpinned: cvxopt is absent)
            norm = curves.shape_constrained_fit(norm, self.mode)
        corrected = curves.mas_correct(norm, r["density"], self.mode)
        return r["n_steps"] + 1, corrected, r["entropy"], r["density"], norm


class RISEMetric(_PerturbationMetric):
    """reference util/test_methods/RISETestFunctions.py:34-237"""
    MODES = ('del', 'ins', 'morf', 'lerf')

    def single_run(self, img_tensor, saliency_map, device, patch_mask=None, max_batch_size=50, return_embeddings=False, *,
                   salient_order=None):
        r = self._run(img_tensor, saliency_map, device, patch_mask, max_batch_size, want_embeddings=return_embeddings,
                      given_order=salient_order)
        if return_embeddings:
            return self._embeddings_tuple(r)
        norm = curves.monotone_normalise(r["response"], r["baseline_pred"], r["original_pred"], falling=(self.mode != 'ins'))
        return r["n_steps"] + 1, r["entropy"], norm


class AICMetric(_PerturbationMetric):
    """reference util/test_methods/AICTestFunctions.py:34-225: the statistic is argmax == target."""
    MODES = ('del', 'ins')

    def single_run(self, img_tensor, saliency_map, device, patch_mask=None, max_batch_size=50, decision_flip=False,
                   CLIP_test_info=None, *, salient_order=None):
        r = self._run(img_tensor, saliency_map, device, patch_mask, max_batch_size, CLIP_test_info, given_order=salient_order)
        response = (r["argmax"] == r["target"]).astype(np.float64)
        original_pred = 1
        baseline_pred = int(r["baseline_class"] == r["target"])
        response[0] = baseline_pred if self.mode == 'ins' else original_pred
        if decision_flip:
            hit = np.where(response == (0 if self.mode == 'del' else 1))[0][0]
            return hit / len(response), response
        with np.errstate(divide="ignore", invalid="ignore"):
            norm = curves.monotone_normalise(response, baseline_pred, original_pred, falling=(self.mode == 'del'))
        return r["n_steps"] + 1, norm


class PositiveNegativePerturbation(_PerturbationMetric):
    """reference util/test_methods/PosNegPertFunctions.py:14-175: returns the RAW response."""
    MODES = ('lerf', 'morf')

    def single_run(self, img_tensor, saliency_map, device, patch_mask=None, max_batch_size=50, CLIP_test_info=None, *,
                   salient_order=None):
        r = self._run(img_tensor, saliency_map, device, patch_mask, max_batch_size, CLIP_test_info, given_order=salient_order)
        return r["n_steps"] + 1, r["response"]


class MonotonicityMetric(_PerturbationMetric):
    """reference util/test_methods/MonotonicityTest.py:34-213"""
    MODES = ('positive', 'negative')
    ALWAYS_LEFTOVER = True

    def _descending(self):
        return True

    def single_run(self, img_tensor, saliency_map, device, patch_mask=None, max_batch_size=50, CLIP_test_info=None, *,
                   salient_order=None):
        r = self._run(img_tensor, saliency_map, device, patch_mask, max_batch_size, CLIP_test_info, given_order=salient_order)
        n1 = r["n_steps"] + 1
        ramp = np.linspace(1, 0, n1) if self.mode == "negative" else np.linspace(0, 1, n1)
        return r["response"], spearmanr(ramp, r["response"]).correlation
