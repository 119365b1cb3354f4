"""Host-side curve arithmetic of the insertion/deletion metrics: 225-point float64 vectors,
NumPy on the CPU exactly as in the reference (these are not kernels; nothing here touches
image-sized data)."""
import numpy as np


def auc(arr):
    """Normalised trapezoid area (reference MASTestFunctions.py:30-32)."""
    return (arr.sum() - arr[0] / 2 - arr[-1] / 2) / (arr.shape[0] - 1)


def step_plan(HW, step_size, max_batch_size, patch_mask=None, always_leftover=False):
    """-> (n_steps, step_size, [batch sizes]) (reference MASTestFunctions.py:88-98,232-242;
    MonotonicityTest.py:160-161 appends the remainder batch even when it is empty)."""
    if patch_mask is None:
        n_steps = (HW + step_size - 1) // step_size
    else:
        n_steps = len(np.unique(np.asarray(patch_mask.cpu() if hasattr(patch_mask, "cpu") else patch_mask)))
        step_size = int(HW / n_steps)
    bs = n_steps if n_steps < max_batch_size else max_batch_size
    full, left = divmod(n_steps, bs)
    batches = [bs] * full
    if left != 0 or always_leftover:
        batches.append(left)
    return n_steps, step_size, batches


def monotone_normalise(response, baseline_pred, original_pred, falling):
    """clip((r - base)/|orig - base|, 0, 1) then running min (falling) / running max
    (reference MASTestFunctions.py:297-309)."""
    out = response.copy()
    lo, hi = 1.0, 0.0
    for i in range(len(out)):
        v = np.clip((out[i] - baseline_pred) / abs(original_pred - baseline_pred), 0.0, 1.0)
        if falling:
            lo = min(lo, v)
            out[i] = lo
        else:
            hi = max(hi, v)
            out[i] = hi
    return out


def density_curve(seg_f32, total_f32, inserting):
    """Cumulative attribution share: float32 ratio added into a float64 curve
    (reference MASTestFunctions.py:225-230,259-263)."""
    n = len(seg_f32)
    dens = np.zeros(n + 1)
    dens[0] = 0 if inserting else 1
    total = np.float32(total_f32)
    for i in range(n):
        share = np.float32(seg_f32[i]) / total
        dens[i + 1] = dens[i] + share if inserting else dens[i] - share
    return dens


def mas_correct(normalised, density, mode):
    """Alignment penalty, clip, min-max rescale, NaN guard (reference MASTestFunctions.py:352-368)."""
    n = len(normalised)
    penalty = np.abs(normalised - density)
    corrected = normalised - penalty if mode == "ins" else normalised + penalty
    corrected = corrected.clip(0, 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        corrected = (corrected - np.min(corrected)) / (np.max(corrected) - np.min(corrected))
    if np.isnan(corrected).any():
        corrected = np.linspace(1, 0, n) if mode in ("del", "morf") else np.linspace(0, 1, n)
    return corrected


def patch_flip_steps(saliency_map, patch_mask, HW, n_steps, descending):
    """patch_mask branch (reference MASTestFunctions.py:213-223,253): rank the patches by
    their mean saliency (stable sort, see DESIGN.md 'pixel order') and give every pixel the
    step at which its patch flips.  Returns (flip_step int32 (HW,), patch order)."""
    pm = np.asarray(patch_mask.cpu() if hasattr(patch_mask, "cpu") else patch_mask).reshape(-1)
    flat = np.asarray(saliency_map).reshape(HW)
    seg = np.zeros(n_steps)
    for i in range(n_steps):
        seg[i] = np.mean(flat[pm == i])
    asc = np.argsort(seg, kind="stable")
    order = asc[::-1].copy() if descending else asc
    step_of_patch = np.empty(n_steps, dtype=np.int32)
    step_of_patch[order] = np.arange(n_steps, dtype=np.int32)
    return step_of_patch[pm].astype(np.int32), order


def patch_density_sums(saliency_map, flip_step, HW, n_steps):
    """float32 attribution mass of every step + total, NumPy order of summation."""
    flat = np.asarray(saliency_map).reshape(HW)
    seg = np.array([np.sum(flat[flip_step == t]) for t in range(n_steps)], dtype=np.float32)
    return seg, np.sum(flat.reshape(1, 1, HW))
