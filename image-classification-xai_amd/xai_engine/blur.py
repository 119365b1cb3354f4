"""Gaussian-blur substrate: the reference's `gkern` (host, scipy) and a device-side separable
application of it (K7) to use as `substrate_fn`."""
import numpy as np
import torch
from scipy.ndimage import gaussian_filter

from . import kernels as K
from .ig import hip_device


def gkern(klen, nsig):
    """(3,3,klen,klen) float32 CPU tensor: scipy-smoothed dirac on the channel diagonal
    (reference MASTestFunctions.py:11-28).  A klen x klen host computation done once."""
    d = np.zeros((klen, klen))
    d[klen // 2, klen // 2] = 1
    k = gaussian_filter(d, nsig)
    kern = np.zeros((3, 3, klen, klen))
    kern[0, 0] = kern[1, 1] = kern[2, 2] = k
    return torch.from_numpy(kern.astype('float32'))


def gkern1d(klen, nsig):
    """1-D factor v of gkern: gkern[c,c] = outer(v, v) (scipy filters axis by axis)."""
    d = np.zeros(klen)
    d[klen // 2] = 1
    return torch.from_numpy(gaussian_filter(d, nsig).astype('float32'))


class GaussianBlur:
    """substrate_fn for the insertion metrics: zero-padded blur with gkern(klen, nsig), computed
    on `device` by the separable HIP kernel.  Accepts a CPU or device (B,C,H,W) tensor and
    returns a device tensor.  Equivalent of `lambda x: conv2d(x, gkern(klen, nsig),
    padding=klen//2)` (reference evaluatePerturbation.py:456-459)."""

    def __init__(self, klen, nsig, device):
        self.device = hip_device(device)
        self.k1d = gkern1d(klen, nsig).to(self.device)

    def __call__(self, x):
        return K.blur_sep(x.to(self.device, torch.float32).contiguous(), self.k1d)


def blur_until_unconfident(model, input_tensor, target_class, device, klen=31, ksig=31, threshold_pct=1.0, max_klen=101):
    """Growing-kernel blur search of the reference's MDA branch (evaluatePerturbation.py:241-257): start at
    gkern(31,31), add 4 taps / 4 sigma until the blurred image's softmax confidence in `target_class` is at most
    `threshold_pct` percent or the kernel exceeds `max_klen` taps.  Returns (GaussianBlur, klen, ksig, confidence %)."""
    dev = hip_device(device)
    x = input_tensor.to(dev, torch.float32)

    def confidence(b):
        with torch.no_grad():
            out = model(b(x))
            out = out if isinstance(out, torch.Tensor) else out.logits
            return float(torch.softmax(out, 1)[0, int(target_class)]) * 100

    blur = GaussianBlur(klen, ksig, dev)
    pct = confidence(blur)
    while pct > threshold_pct:
        klen += 4
        ksig += 4
        blur = GaussianBlur(klen, ksig, dev)
        pct = confidence(blur)
        if klen > max_klen:
            break
    return blur, klen, ksig, pct
