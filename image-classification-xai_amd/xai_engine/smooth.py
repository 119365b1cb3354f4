"""SmoothGrad over IG (reference saliencyMethods.py:184-205), quirks included.

Observable behaviour kept from the reference:
  * noise is drawn with the global torch CPU generator, one `torch.normal` per sample;
  * `total_gradients[i], _, _ = IG(...)` unpacks the (3,H,W) result along its first axis, so
    only channel 0 of every sample survives and is broadcast over the 3 channels (:196);
  * "LIG" / "IDG" hand IG a fractional batch size (`int(steps/2)/2`) and therefore fail
    exactly as they do in the reference (a float cannot index the alpha schedule).
"""
import torch

from . import kernels as K
from .ig import IG, IDG, hip_device, ig_batch


def smoothGrad(attribution, input, model, steps, baseline, target_class, device, sigma_spread=.15, samples=25, vis=False):
    dev = hip_device(device)
    x = input.to(dev, torch.float32)
    stdev = float(sigma_spread * (torch.max(x) - torch.min(x)))
    C = x.shape[1]
    first = torch.empty((1, samples, 1) + tuple(x.shape[2:]), dtype=torch.float32, device=dev)   # channel 0 of every sample
    noisy = torch.empty((samples,) + tuple(x.shape[1:]), dtype=torch.float32, device=dev)
    for i in range(samples):
        noise = torch.normal(mean=0, std=stdev, size=input.shape)           # host RNG stream, as in the reference
        noisy[i] = (x + noise.to(dev))[0]
    if attribution == "IG" and steps % int(steps / 2) == 0:
        # all samples through the multi-image engine: same per-sample arithmetic as `samples` IG calls with
        # batch_size = steps/2 (the classifier sees 2 x steps interpolants per pass instead of steps/2)
        targets = torch.as_tensor(target_class).reshape(-1)[:1].repeat(samples)
        base = baseline.to(dev).expand_as(noisy).contiguous() if torch.is_tensor(baseline) else baseline
        first[0, :, 0] = ig_batch(noisy, model, targets, steps=steps, alpha_star=1, baseline=base, images_per_pass=2, streams=3)[:, 0]   # passes overlap on 3 streams (bit-identical to 1)
    else:
        for i in range(samples):
            if attribution == "IG":
                a, _, _ = IG(noisy[i].unsqueeze(0), model, steps, int(steps / 2), 1, baseline, dev, target_class)
            elif attribution == "LIG":
                a, _, _ = IG(noisy[i].unsqueeze(0), model, steps, int(steps / 2) / 2, .9, baseline, dev, target_class)
            elif attribution == "IDG":
                a, _, _ = IDG(noisy[i].unsqueeze(0), model, steps, int(steps / 2) / 2, baseline, dev, target_class)
            else:
                a = torch.zeros(tuple(x.shape[2:]), device=dev)             # unknown name: rows stay zero
            first[0, i, 0] = a
    ones = torch.ones((1, 1) + tuple(x.shape[2:]), dtype=torch.float32, device=dev)
    mean0 = K.ig_accum(first, ones, 0.0)[0, 0]                              # mean over samples (x - 0 == 1)
    mean = mean0.unsqueeze(0).expand(C, -1, -1).contiguous()
    if not vis:
        return mean
    total = first[0].expand(-1, C, -1, -1).contiguous()
    return mean, total, noisy
