"""Oracle: Integrated-Gradients family (test infrastructure only -- see oracle/__init__.py).

Restates util/attribution_methods/saliencyMethods.py of the reference with NumPy float32
arithmetic.  The classifier is an opaque torch module: `grads_and_logits` /
`logits_only` are the only places torch is touched (reference: getGradientsParallel
:209-215, getPredictionParallel :218-224).
"""
import numpy as np
import torch

F32 = np.float32


def _device_of(model):
    for p in model.parameters():
        return p.device
    return torch.device("cpu")


def grads_and_logits(model, batch, target):
    """d logit_target / d input for every image of `batch` (B,C,H,W float32 ndarray).
    [saliencyMethods.py:209-215] -- raw logits, not softmax."""
    x = torch.from_numpy(np.ascontiguousarray(batch)).to(_device_of(model)).requires_grad_(True)
    out = model(x)
    out = out if isinstance(out, torch.Tensor) else out.logits
    score = out[:, int(target)]
    (g,) = torch.autograd.grad(score, x, grad_outputs=torch.ones_like(score))
    return g.detach().cpu().numpy(), score.detach().cpu().numpy()


def logits_only(model, batch, target):
    """[saliencyMethods.py:218-224]"""
    with torch.no_grad():
        out = model(torch.from_numpy(np.ascontiguousarray(batch)).to(_device_of(model)))
    out = out if isinstance(out, torch.Tensor) else out.logits
    return out[:, int(target)].cpu().numpy()


def linspace01(steps):
    """float32 linspace(0, 1, steps) [saliencyMethods.py:21].  Delegated to torch.linspace on
    the CPU -- the very call the reference makes -- because its float32 result is not a
    pure formula: ATen's vectorised CPU fill rounds `start + step*block` and then adds
    `step*lane`, so the last bit depends on the host's SIMD width."""
    return torch.linspace(0, 1, steps).numpy()


def linspace(a, b, n):
    """torch.linspace(a, b, n) for python-float bounds [saliencyMethods.py:302]."""
    return torch.linspace(a, b, n).numpy()


def as_baseline(x, baseline):
    """float -> full image, tensor/array -> as is [saliencyMethods.py:30-33]."""
    if np.isscalar(baseline):
        return np.full(x.shape, baseline, dtype=F32)
    return np.asarray(baseline, dtype=F32).reshape(x.shape)


def interpolate(x, base, alphas):
    """K1: base + alpha * (x - base), two roundings (no fused multiply-add)
    [saliencyMethods.py:38,44]."""
    diff = x - base
    return base[None] + alphas.reshape(-1, 1, 1, 1) * diff[None]


def path_gradients(model, x, base, alphas, batch_size, target, want_grads=True):
    steps = alphas.shape[0]
    grads = np.zeros((steps,) + x.shape, dtype=F32) if want_grads else None
    logits = np.zeros(steps, dtype=F32)
    for lo in range(0, steps, batch_size):
        imgs = interpolate(x, base, alphas[lo:lo + batch_size])
        if want_grads:
            grads[lo:lo + batch_size], logits[lo:lo + batch_size] = grads_and_logits(model, imgs, target)
        else:
            logits[lo:lo + batch_size] = logits_only(model, imgs, target)
    return grads, logits


def left_cutoff(logits, alpha_star):
    """Number of leading steps Left-IG averages [saliencyMethods.py:48-67]."""
    thr = F32(logits.max()) * F32(alpha_star)
    hit = np.nonzero(logits > thr)[0]
    cut = int(hit[0]) if hit.size else 1
    return max(cut, 1)


def accumulate(grads, n_use, x, base):
    """K2: mean of the first n_use step-gradients times (x - base)
    [saliencyMethods.py:53,67,70]."""
    mean = grads[:n_use].sum(axis=0, dtype=F32) / F32(n_use)
    return mean * (x - base)


def ig(x, model, steps, batch_size, alpha_star, baseline, target, return_path=False):
    """IG (alpha_star == 1) / Left-IG.  x: (1,C,H,W) float32.  Returns (C,H,W)
    [saliencyMethods.py:13-72]; (0,0,0,0) when steps % batch_size != 0 (:14-16)."""
    if steps % batch_size != 0:
        return 0, 0, 0, 0
    x = np.asarray(x, dtype=F32)[0]
    base = as_baseline(x, baseline if np.isscalar(baseline) else np.asarray(baseline)[0])
    grads, logits = path_gradients(model, x, base, linspace01(steps), batch_size, target)
    n_use = steps if alpha_star == 1 else left_cutoff(logits, alpha_star)
    out = accumulate(grads, n_use, x, base)
    return (out, grads, logits, n_use) if return_path else out


def grads_and_logits_rows(model, batch, targets):
    """`grads_and_logits` with one target class per row: row r is scored by logit[targets[r]].  Rows are independent in
    the classifier (eval mode), so this is getGradientsParallel [saliencyMethods.py:209-215] applied to several images'
    interpolants stacked into one classifier batch."""
    x = torch.from_numpy(np.ascontiguousarray(batch)).to(_device_of(model)).requires_grad_(True)
    out = model(x)
    out = out if isinstance(out, torch.Tensor) else out.logits
    idx = torch.as_tensor(np.asarray(targets, dtype=np.int64)).to(out.device)
    score = out.gather(1, idx.unsqueeze(1)).squeeze(1)
    (g,) = torch.autograd.grad(score, x, grad_outputs=torch.ones_like(score))
    return g.detach().cpu().numpy(), score.detach().cpu().numpy()


def ig_stacked(xs, model, steps, images_per_pass, baseline, targets):
    """The reference's IG (alpha_star == 1) [saliencyMethods.py:13-72] for several images whose interpolants share classifier
    passes: `images_per_pass` images x `steps` interpolants ([image][step] order) form one batch -- a batch the reference's
    one-image signature cannot form (:14-16) but bench.py's headline configuration runs.  Everything outside the classifier
    call is the one-image oracle (`interpolate`, `accumulate`).  xs: (B,C,H,W) -> (B,C,H,W)."""
    xs = np.asarray(xs, dtype=F32)
    al = linspace01(steps)
    out = np.zeros_like(xs)
    for lo in range(0, xs.shape[0], images_per_pass):
        part = xs[lo:lo + images_per_pass]
        bases = [as_baseline(x, baseline) for x in part]
        batch = np.concatenate([interpolate(x, b, al) for x, b in zip(part, bases)])
        rows = np.repeat(np.asarray(targets[lo:lo + images_per_pass], dtype=np.int64), steps)
        grads, _ = grads_and_logits_rows(model, batch, rows)
        for j, (x, b) in enumerate(zip(part, bases)):
            out[lo + j] = accumulate(grads[j * steps:(j + 1) * steps], steps, x, b)
    return out


def slopes(x, base, model, steps, batch_size, target):
    """Finite-difference logit slopes on the uniform path [saliencyMethods.py:226-261]."""
    al = linspace01(steps)
    _, logits = path_gradients(model, x, base, al, batch_size, target, want_grads=False)
    dx = float(al[1] - al[0])
    sl = np.zeros(steps, dtype=F32)
    sl[1:] = (logits[1:] - logits[:-1]) / F32(dx)
    return sl, dx


def alpha_parameters(sl, steps, step_size):
    """IDG's slope-proportional alpha schedule [saliencyMethods.py:264-314]."""
    sl = np.asarray(sl, dtype=F32)
    norm01 = (sl - sl.min()) / (sl.max() - sl.min())
    norm01[0] = 0
    share = norm01 / norm01.sum(dtype=F32)
    want = share * F32(steps)
    count = want.astype(np.int32)                       # truncation
    spare = steps - int(count.sum())
    want = want.copy()
    want[count != 0] = -1
    # highest fractional demand first; torch.sort is stable only by accident here, ties are
    # between exact zeros and do not matter unless `spare` reaches into them
    by_need = np.argsort(want, kind="stable")[::-1]
    count[by_need[:spare]] = 1
    alphas = np.zeros(steps, dtype=F32)
    sub = np.zeros(steps, dtype=F32)
    at, a0 = 0, 0.0
    for n in count:
        n = int(n)
        if n == 0:
            continue
        seg = linspace(a0, a0 + step_size, n + 1)[:n]
        alphas[at:at + n] = seg
        sub[at:at + n] = step_size / n
        at += n
        a0 += step_size
    return alphas, sub


def idg(x, model, steps, batch_size, baseline, target):
    """Integrated Decision Gradients [saliencyMethods.py:74-136]."""
    if batch_size == 0 or steps % batch_size != 0:
        return 0, 0, 0
    x = np.asarray(x, dtype=F32)[0]
    base = as_baseline(x, baseline if np.isscalar(baseline) else np.asarray(baseline)[0])
    sl, dx = slopes(x, base, model, steps, batch_size, target)
    alphas, sub = alpha_parameters(sl, steps, dx)
    grads, logits = path_gradients(model, x, base, alphas, batch_size, target)
    w = np.zeros(steps, dtype=F32)
    w[1:] = (logits[1:] - logits[:-1]) / (alphas[1:] - alphas[:-1])
    weighted = grads * w.reshape(-1, 1, 1, 1)
    weighted = weighted * sub.reshape(-1, 1, 1, 1)
    return (weighted.sum(axis=0, dtype=F32) / F32(steps)) * (x - base)


def idgi(x, model, steps, batch_size, baseline, target):
    """IDGI: sum_i g_i^2 * (logit_{i+1} - logit_i) / sum(g_i^2), last step dropped
    [saliencyMethods.py:139-181]."""
    if steps % batch_size != 0:
        return 0, 0, 0, 0
    x = np.asarray(x, dtype=F32)[0]
    base = as_baseline(x, baseline if np.isscalar(baseline) else np.asarray(baseline)[0])
    grads, logits = path_gradients(model, x, base, linspace01(steps), batch_size, target)
    acc = np.zeros_like(x)
    for i in range(steps - 1):
        sq = grads[i] * grads[i]
        acc += sq * (logits[i + 1] - logits[i]) / sq.sum(dtype=F32)
    return acc


def smoothgrad_ig(x, model, steps, baseline, target, sigma_spread=.15, samples=25):
    """smoothGrad("IG", ..., vis=True) [saliencyMethods.py:184-205] -> (mean, total_gradients, noisy_imgs).
    Noise: one torch.normal per sample from the GLOBAL CPU generator (the very call the reference makes, :191 -- seed it
    with torch.manual_seed before calling), std = sigma_spread * (max - min) as a float32 0-d tensor.  IG runs with
    batch_size = int(steps / 2) (:196).  The tuple-unpacking of :196 keeps only channel 0 of every sample and
    broadcasts it over the channels."""
    xt = torch.from_numpy(np.asarray(x, dtype=F32))
    stdev = sigma_spread * (torch.max(xt) - torch.min(xt))
    total = np.zeros((samples,) + xt.shape[1:], dtype=F32)
    noisy = np.zeros((samples,) + xt.shape[1:], dtype=F32)
    for i in range(samples):
        noise = torch.normal(mean=0, std=stdev, size=xt.shape)
        noisy[i] = (xt + noise).numpy()[0]
        a = ig(noisy[i][None], model, steps, int(steps / 2), 1, baseline, target)
        total[i] = a[0][None]                       # first of the three unpacked (H,W) slices, broadcast
    mean = total.sum(axis=0, dtype=F32) / F32(samples)
    return mean, total, noisy
