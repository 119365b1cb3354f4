"""Grad-CAM on the HIP kernels, behind captum's `LayerGradCam` call shape.

The reference calls captum 0.7.0 (evaluatePerturbation.py:147-153):
    LayerGradCam(model, model.layer4).attribute(x, target, relu_attributions=True)   # (1,1,7,7)
then torchvision Resize -> x ones(3,H,W) -> |sum over channels| (:153,:181).  The layer forward
and the backward to the layer stay PyTorch-ROCm; the channel-weighted reduction and the
up-sample are xai_gradcam_f32 / xai_bilinear_up_f32.
"""
import threading

import torch

from . import kernels as K
from ._lib import XaiHipError
from .streams import CAPTURE_LOCK, backward_turn


def _grad_of_activation(score, act):
    """d score / d act.  A classifier prepared with prepare.fuse_bn_relu(fork_residual=True) hands a block output on as two
    tensors on one storage (act and act._xai_alias); the gradient of the activation is the sum over both handles."""
    alias = getattr(act, "_xai_alias", None)
    with backward_turn(act.device):                       # streams.py: backward passes on autograd's shared device thread take turns
        if alias is None:
            (grad,) = torch.autograd.grad(score, act)
            return grad
        ga, gb = torch.autograd.grad(score, [act, alias], allow_unused=True)
    if ga is None or gb is None:
        return ga if gb is None else gb
    return ga + gb


class LayerGradCam:
    def __init__(self, forward_func, layer, device_ids=None):
        self.forward_func = forward_func
        self.layer = layer

    def _act_and_grad(self, inputs, target, additional_forward_args=None, layer_input=False):
        kept = {}
        me = threading.get_ident()       # module hooks are shared by every thread that runs this model: keep only OUR pass's tensor
        if layer_input:                                 # captum's attribute_to_layer_input: the layer's (first) input instead of its output
            handle = self.layer.register_forward_pre_hook(lambda mod, inp: kept.__setitem__("act", inp[0]) if threading.get_ident() == me else None)
        else:
            handle = self.layer.register_forward_hook(lambda mod, inp, out: kept.__setitem__("act", out) if threading.get_ident() == me else None)
        extra = () if additional_forward_args is None else \
            (tuple(additional_forward_args) if isinstance(additional_forward_args, (tuple, list)) else (additional_forward_args,))
        try:
            with torch.enable_grad():
                if not inputs.requires_grad:        # captum's apply_gradient_requirements: the layer output
                    inputs = inputs.detach().requires_grad_(True)   # needs a graph even with frozen weights
                out = self.forward_func(inputs, *extra)
                out = out if isinstance(out, torch.Tensor) else out.logits
                if target is None:
                    score = out.sum()
                elif torch.is_tensor(target) and target.dim() > 0 and target.numel() == out.shape[0] and out.shape[0] > 1:
                    score = out.gather(1, target.reshape(-1, 1).to(out.device)).sum()
                else:
                    score = out[:, int(target)].sum()
                if not torch.is_tensor(kept.get("act")):
                    raise XaiHipError("LayerGradCam: the layer's " + ("input" if layer_input else "output") + " is not a single tensor")
                grad = _grad_of_activation(score, kept["act"])
        finally:
            handle.remove()
        return kept["act"].detach(), grad.detach()

    def attribute(self, inputs, target=None, additional_forward_args=None, attribute_to_layer_input=False,
                  relu_attributions=False, attr_dim_summation=True):
        """captum 0.7.0's LayerGradCam.attribute for one input tensor and one layer tensor: the gradient of the target score with
        respect to the layer, averaged over every axis after the channel axis, weighs the layer's activation; with
        `attr_dim_summation` the weighted channels are summed (-> (B,1,*spatial)); optional ReLU.
        The harness's call shape -- a (B,C,h,w) layer output, summed -- is one xai_gradcam_f32 launch; so is any layer of rank
        >= 3 with at most 1024 positions per channel (a ViT block's (B,tokens,dim): "channels" = tokens).  What that kernel does not
        cover (no channel sum, rank-2 layers, larger maps) is the same three-line expression in device torch ops -- never the CPU."""
        if not inputs.is_cuda:
            raise XaiHipError("LayerGradCam.attribute needs its input on a HIP device ('cuda:N')")
        act, grad = self._act_and_grad(inputs, target, additional_forward_args, attribute_to_layer_input)
        act, grad = act.float().contiguous(), grad.float().contiguous()
        spatial = tuple(act.shape[2:])
        n_pos = 1
        for d in spatial:
            n_pos *= int(d)
        if attr_dim_summation and act.dim() >= 3 and n_pos <= 1024:
            B, Cc = act.shape[0], act.shape[1]
            cam = K.gradcam(act.reshape(B, Cc, n_pos, 1), grad.reshape(B, Cc, n_pos, 1), relu=relu_attributions)
            return cam.reshape((B, 1) + spatial)
        weights = grad.mean(dim=tuple(range(2, grad.dim())), keepdim=True) if grad.dim() > 2 else grad
        scaled = weights * act
        if attr_dim_summation:
            scaled = scaled.sum(dim=1, keepdim=True)
        return torch.relu(scaled) if relu_attributions else scaled


def patch_captum():
    """Opt-in ("captum itself is not replaced or monkey-patched unless asked"): make
    `from captum.attr import LayerGradCam` -- evaluatePerturbation.py:43, the one harness import of the Grad-CAM path that does
    not go through `util.*` -- resolve to the HIP engine's class.  Only that one name of an installed captum is rebound; every
    other captum class stays captum's.  Returns the class that was replaced (None if captum is not importable: nothing to do).
    Asked for either by calling this function before the harness's imports, or by XAI_PATCH_CAPTUM=1 in the environment
    (honoured when the `util` mirror is imported, i.e. at the harness's first `util` import, :17)."""
    try:
        import captum.attr as cattr
    except ImportError:
        return None
    old = getattr(cattr, "LayerGradCam", None)
    if old is LayerGradCam:
        return old
    cattr.LayerGradCam = LayerGradCam
    try:                                            # captum re-exports the class from its defining module as well
        import importlib
        mod = importlib.import_module("captum.attr._core.layer.grad_cam")
        mod.LayerGradCam = LayerGradCam
    except ImportError:
        pass
    return old


def gradcam_saliency(model, layer, inputs, target, out_hw, channels=3):
    """The (B,H,W) map get_CNN_attr produces for "gc": |sum of `channels` copies of the
    up-sampled, ReLU'd cam| (reference evaluatePerturbation.py:147-153,181), fused into the
    up-sample kernel as scale = channels, take_abs."""
    cam = LayerGradCam(model, layer).attribute(inputs, target, relu_attributions=True)
    return K.bilinear_up(cam[:, 0].contiguous(), out_hw[0], out_hw[1], scale=float(channels), take_abs=True)


def _n_classes(model, x):
    with torch.no_grad():
        out = model(x)
    return (out if isinstance(out, torch.Tensor) else out.logits).shape[1]


class CapturedGradCam:
    """`gradcam_saliency` for a fixed input shape as ONE hipGraph replay.

    A one-image Grad-CAM is launch-bound: ~500 small classifier kernels (forward, backward to the layer) plus K3 take
    ~3 ms of host launches for well under 1 ms of GPU work.  The whole sequence -- classifier forward, autograd to the
    layer, xai_gradcam_f32, xai_bilinear_up_f32 -- is captured once (torch.cuda.CUDAGraph == hipGraph on ROCm) on
    static input / target buffers and replayed per image (3.2 ms -> 1.0 ms per image on ResNet-50).  Same kernels, same
    arithmetic as the eager path; MIOpen's kernels are not run-to-run deterministic, so the two agree to rounding (<= 1e-6).

        cam = CapturedGradCam(model, model.layer4, example_input, (224, 224))
        sal = cam(x, target)            # (B,H,W) on the device, same values as gradcam_saliency(model, layer, x, target, ...)
    """

    def __init__(self, model, layer, example_input, out_hw, channels=3, warmup=3, verify=True):
        if not example_input.is_cuda:
            raise XaiHipError("CapturedGradCam needs its input on a HIP device ('cuda:N')")
        self.dev = example_input.device
        self.out_hw = (int(out_hw[0]), int(out_hw[1]))
        self.x = example_input.detach().float().clone().requires_grad_(True)
        self.target = torch.zeros(self.x.shape[0], dtype=torch.int64, device=self.dev)
        self._cam = LayerGradCam(model, layer)
        self._channels = float(channels)
        with CAPTURE_LOCK:
            side = torch.cuda.Stream(self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(side):
                for _ in range(warmup):                       # MIOpen picks its algorithms here, never inside the capture
                    self._run()
            torch.cuda.current_stream(self.dev).wait_stream(side)
            torch.cuda.synchronize(self.dev)
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: other stream workers keep launching and allocating while this thread captures; the default ("global") lets
            # any other thread's hipMalloc invalidate the capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.sal = self._run()
        if verify:
            self._verify(model, layer, channels)

    def _verify(self, model, layer, channels):
        """Replay against the eager path on the example input for two classes.  A graph replays raw pointers: every tensor
        its kernels read (inputs, targets, parameters, anything allocated outside the capture) has to outlive it, and most
        of its nodes are library kernels we do not control -- so a captured Grad-CAM proves itself on this model before it is
        handed out."""
        x = self.x.detach().clone()
        n_cls = int(_n_classes(model, x))
        for t in {0, n_cls - 1}:
            want = gradcam_saliency(model, layer, x, t, self.out_hw, channels)
            got = self(x, t)
            scale = float(want.abs().max())
            if not float((got - want).abs().max()) <= 1e-4 * max(scale, 1e-30):
                raise XaiHipError("CapturedGradCam: the hipGraph replay does not reproduce the eager Grad-CAM on this model "
                                  "(a captured node is not replay-safe here); use gradcam_saliency instead")

    def _run(self):
        act, grad = self._act_grad()
        cam = K.gradcam(act.float().contiguous(), grad.float().contiguous(), relu=True)
        return K.bilinear_up(cam, self.out_hw[0], self.out_hw[1], scale=self._channels, take_abs=True)

    def _act_grad(self):
        kept = {}
        me = threading.get_ident()       # (see LayerGradCam._act_and_grad: hooks of other stream workers fire on our forward too)
        handle = self._cam.layer.register_forward_hook(lambda mod, inp, out: kept.__setitem__("act", out) if threading.get_ident() == me else None)
        try:
            with torch.enable_grad():
                out = self._cam.forward_func(self.x)
                out = out if isinstance(out, torch.Tensor) else out.logits
                score = out.gather(1, self.target.view(-1, 1)).sum()        # target read from the static device buffer
                grad = _grad_of_activation(score, kept["act"])
        finally:
            handle.remove()
        return kept["act"].detach(), grad.detach()

    def __call__(self, inputs, target):
        if tuple(inputs.shape) != tuple(self.x.shape):
            raise ValueError(f"captured for inputs of shape {tuple(self.x.shape)}, got {tuple(inputs.shape)}")
        with torch.no_grad():
            self.x.copy_(inputs, non_blocking=True)
            t = target if torch.is_tensor(target) else torch.tensor(target)
            self.target.copy_(t.to(self.dev, torch.int64).reshape(-1).expand(self.x.shape[0]), non_blocking=True)
        with backward_turn(self.dev):                     # the graph holds backward kernels (streams.py)
            self.graph.replay()
        return self.sal.clone()
