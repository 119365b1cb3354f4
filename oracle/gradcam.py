"""Oracle: Grad-CAM reduce + bilinear upsample (test infrastructure only).

PARITY: pinned to the reference-owned CAM arithmetic, unpinned against captum itself.  The reference
does not implement Grad-CAM: it calls captum==0.7.0 `LayerGradCam(model, model.layer4).attribute(x,
target, relu_attributions=True)` (XAI_Survey/evaluations/evaluatePerturbation.py:147-153;
requirements.txt:1).  captum is neither vendored in the reference nor installed here, and no reference
file stores a captum output, so this restates captum's published algorithm:
    w[c]   = mean over (h,w) of d logit_t / d A[c]
    cam    = relu( sum_c w[c] * A[c] )            (keepdim -> (B,1,h,w))
The same arithmetic exists in code the reference does own -- ViT_CX/get_feature_map.py:17-23 (channel
weights = spatial mean of the gradients) and ViT_CX/base_cam.py:55-61,129 (weighted channel sum,
negatives clamped) -- and tests/golden/cam.npz holds vectors produced by running exactly those two
methods (tests/golden/make_golden.py: cam_fixture), which `cam_reduce` must reproduce.
The upsample is torchvision `Resize((H,W), antialias=True)` on a float tensor, i.e.
`F.interpolate(mode="bilinear", align_corners=False, antialias=True)`; for up-sampling the
anti-alias filter degenerates to plain bilinear, which is what `bilinear_up` restates.
"""
import numpy as np
import torch

F32 = np.float32


def layer_act_and_grad(model, layer, x, target):
    """Forward hook on `layer`; gradient of logit[target] w.r.t. the layer output."""
    keep = {}
    h = layer.register_forward_hook(lambda m, i, o: keep.__setitem__("a", o))
    try:
        xt = torch.as_tensor(x).detach().requires_grad_(True)     # frozen weights: the graph must start at the input
        out = model(xt)
        out = out if isinstance(out, torch.Tensor) else out.logits
        score = out[:, int(target)].sum()
        (g,) = torch.autograd.grad(score, keep["a"])
    finally:
        h.remove()
    return keep["a"].detach().cpu().numpy(), g.detach().cpu().numpy()


def cam_reduce(act, grad, relu=True):
    """(B,C,h,w),(B,C,h,w) -> (B,h,w), float32."""
    w = grad.mean(axis=(2, 3), dtype=F32, keepdims=True)
    cam = (w * act).sum(axis=1, dtype=F32)
    return np.maximum(cam, F32(0)) if relu else cam


def bilinear_up(src, H, W):
    """align_corners=False bilinear (B,h,w) -> (B,H,W): source coordinate
    max((d + 0.5) * (in/out) - 0.5, 0), neighbour clamped to the last row/column."""
    B, h, w = src.shape

    def taps(n_in, n_out):
        scale = F32(n_in) / F32(n_out)
        d = np.arange(n_out, dtype=F32)
        s = np.maximum((d + F32(0.5)) * scale - F32(0.5), F32(0))
        i0 = np.floor(s).astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (s - i0.astype(F32)).astype(F32)
        return i0, i1, F32(1) - l1, l1

    y0, y1, wy0, wy1 = taps(h, H)
    x0, x1, wx0, wx1 = taps(w, W)
    top = src[:, y0][:, :, x0] * wx0 + src[:, y0][:, :, x1] * wx1
    bot = src[:, y1][:, :, x0] * wx0 + src[:, y1][:, :, x1] * wx1
    return (top * wy0[None, :, None] + bot * wy1[None, :, None]).astype(F32)


def gradcam_saliency(act, grad, H, W, channels=3):
    """What get_CNN_attr hands to the metrics for "gc": |sum over `channels` copies of the
    up-sampled relu'd cam| [evaluatePerturbation.py:151-153,181]."""
    up = bilinear_up(cam_reduce(act, grad, relu=True), H, W)
    rep = np.repeat(up[:, None], channels, axis=1)
    return np.abs(rep.sum(axis=1, dtype=F32))
