"""Oracle: insertion/deletion perturbation loop and its five metrics (test infrastructure
only -- see oracle/__init__.py).

Restates, with NumPy on the host,
  util/test_methods/MASTestFunctions.py      gkern :11-28, auc :30-32, MASMetric.single_run :72-385
  util/test_methods/RISETestFunctions.py     RISEMetric.single_run :51-237
  util/test_methods/AICTestFunctions.py      AICMetric.single_run :51-225
  util/test_methods/PosNegPertFunctions.py   PositiveNegativePerturbation.single_run :31-175
  util/test_methods/MonotonicityTest.py      MonotonicityMetric.single_run :51-213
  XAI_Survey/evaluations/evaluatePerturbation.py  run_perturbation :448-497
The classifier is an opaque callable `logits_fn(batch ndarray (B,C,H,W) f32) -> (B,K) f32`.

Tie rule (DESIGN.md "pixel order"): the reference sorts with NumPy's default, unstable
argsort, so equal saliency values have no defined order there.  Oracle and HIP path both use
the STABLE ascending sort and take the descending order as its exact reverse.
"""
import numpy as np
from scipy.ndimage import gaussian_filter
from scipy.stats import spearmanr

F32 = np.float32


# ------------------------------------------------------------------ small pieces
def gkern(klen, nsig):
    """(3,3,klen,klen) float32 depthwise Gaussian: smooth a centred dirac with scipy's
    reflect-mode gaussian_filter, put it on the channel diagonal [MASTestFunctions.py:11-28]."""
    d = np.zeros((klen, klen))
    d[klen // 2, klen // 2] = 1
    k = gaussian_filter(d, nsig)
    out = np.zeros((3, 3, klen, klen))
    for c in range(3):
        out[c, c] = k
    return out.astype(F32)


def gkern1d(klen, nsig):
    """The 1-D factor v with gkern[c,c] == outer(v, v) up to float64 rounding: scipy's
    gaussian_filter is a sequence of 1-D passes, so smoothing a 1-D dirac gives v."""
    d = np.zeros(klen)
    d[klen // 2] = 1
    return gaussian_filter(d, nsig)


def blur_dense(x, kern):
    """Zero-padded cross-correlation of (B,3,H,W) with the (3,3,k,k) kernel, float32
    accumulation in tap order -- what conv2d(x, kern, padding=k//2) means
    [evaluatePerturbation.py:459]."""
    B, C, H, W = x.shape
    k = kern.shape[-1]
    r = k // 2
    pad = np.zeros((B, C, H + 2 * r, W + 2 * r), dtype=F32)
    pad[:, :, r:r + H, r:r + W] = x
    out = np.zeros((B, C, H, W), dtype=F32)
    for co in range(C):
        for ci in range(C):
            if not kern[co, ci].any():
                continue
            for i in range(k):
                for j in range(k):
                    out[:, co] += kern[co, ci, i, j] * pad[:, ci, i:i + H, j:j + W]
    return out


def auc(arr):
    """Trapezoid area with unit x-range [MASTestFunctions.py:30-32]."""
    return (arr.sum() - arr[0] / 2 - arr[-1] / 2) / (arr.shape[0] - 1)


def softmax_rows(logits):
    z = logits.astype(F32)
    e = np.exp(z - z.max(axis=1, keepdims=True))
    return (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)


def entropy_bits(p):
    with np.errstate(divide="ignore", invalid="ignore"):
        return -(p * np.log2(p)).sum(axis=-1, dtype=F32)       # NaN if some p == 0, as in the reference


def pixel_order(sal, HW, descending):
    asc = np.argsort(np.asarray(sal).reshape(HW), kind="stable")
    return asc[::-1].copy() if descending else asc


# ------------------------------------------------------------------ the sequence of images
class Plan:
    """Step bookkeeping shared by all five metrics [MASTestFunctions.py:88-98,232-242]."""

    def __init__(self, HW, step_size, max_batch_size, patch_mask, always_leftover=False):
        if patch_mask is None:
            self.n_steps = (HW + step_size - 1) // step_size
            self.step_size = step_size
        else:
            self.n_steps = len(np.unique(patch_mask))
            self.step_size = int(HW / self.n_steps)
        bs = self.n_steps if self.n_steps < max_batch_size else max_batch_size
        full, left = divmod(self.n_steps, bs)
        self.batches = [bs] * full
        if left != 0 or always_leftover:          # MonotonicityTest.py:160-161 appends even a 0 batch
            self.batches.append(left)


def flip_groups(sal, HW, plan, patch_mask, descending, order=None):
    """List of pixel-index arrays, one per step, in the order they switch from `start`
    to `finish` [MASTestFunctions.py:207-223,251-253].  `order`: the flip order to use instead of the stable sort -- the
    pixel (or patch) indices, first flipped first -- e.g. the order the reference's unstable argsort gave on a tied map."""
    if order is not None:
        order = np.asarray(order).reshape(-1)
        if patch_mask is None:
            s = plan.step_size
            return [order[i * s:(i + 1) * s] for i in range(plan.n_steps)], order
        pm = np.asarray(patch_mask).reshape(-1)
        return [np.nonzero(pm == order[i])[0] for i in range(plan.n_steps)], order
    if patch_mask is None:
        order = pixel_order(sal, HW, descending)
        s = plan.step_size
        return [order[i * s:(i + 1) * s] for i in range(plan.n_steps)], order
    pm = np.asarray(patch_mask).reshape(-1)
    flat = np.asarray(sal).reshape(HW)
    seg = np.zeros(plan.n_steps)
    for i in range(plan.n_steps):
        seg[i] = np.mean(flat[pm == i])
    order = pixel_order(seg, plan.n_steps, descending)
    return [np.nonzero(pm == order[i])[0] for i in range(plan.n_steps)], order


def sequence(start, finish, groups):
    """Yield the image after each step: cumulative copy finish -> start on the group's
    pixels, all channels [MASTestFunctions.py:255-257]."""
    C = start.shape[1]
    cur = start.reshape(C, -1).copy()
    fin = finish.reshape(C, -1)
    for g in groups:
        cur[:, g] = fin[:, g]
        yield cur.reshape(start.shape[1:]).copy()


def run_steps(logits_fn, start, finish, groups, plan):
    """All step images through the classifier in the reference's batch sizes; returns the
    (n_steps, K) logits and, for the tests, the images themselves."""
    imgs = np.stack(list(sequence(start, finish, groups))) if groups else np.zeros((0,) + start.shape[1:], F32)
    outs, at = [], 0
    for b in plan.batches:
        outs.append(logits_fn(imgs[at:at + b]))
        at += b
    return np.concatenate(outs), imgs


def monotone(curve, base, orig, falling):
    """Clip-normalise then running min (deletion) / running max (insertion)
    [MASTestFunctions.py:297-309]."""
    out = curve.copy()
    lo, hi = 1.0, 0.0
    for i in range(len(out)):
        v = np.clip((out[i] - base) / abs(orig - base), 0.0, 1.0)
        if falling:
            lo = min(lo, v)
            out[i] = lo
        else:
            hi = max(hi, v)
            out[i] = hi
    return out


def _probe(logits_fn, img):
    p = softmax_rows(logits_fn(img))[0]
    return p


# ------------------------------------------------------------------ the five metrics
def special_version_problem(y, mode):
    """The dense QP of special_version=True exactly as the reference assembles it [MASTestFunctions.py:311-345]:
    minimise 1/2 x^T Q x + c^T x  s.t.  G x <= h,  A x = b,  with Q = 2 I, c = -2 y, G = [-I; I; shape rows], h = [0; 1; 0],
    A picking the two end points.  Shape rows: (-1, 2, -1) for 'del', (1, -2, 1) for 'ins', all-zero for every other mode."""
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    Q = 2 * np.eye(n)
    c = -2 * y
    A_ineq = np.zeros((n - 2, n))
    rows = np.arange(n - 2)
    if mode == "del":
        A_ineq[rows, rows], A_ineq[rows, rows + 1], A_ineq[rows, rows + 2] = -1, 2, -1
    elif mode == "ins":
        A_ineq[rows, rows], A_ineq[rows, rows + 1], A_ineq[rows, rows + 2] = 1, -2, 1
    G = np.vstack([-np.eye(n), np.eye(n), A_ineq])
    h = np.hstack([np.zeros(n), np.ones(n), np.zeros(n - 2)])
    A = np.zeros((2, n))
    A[0, 0] = 1
    A[1, -1] = 1
    b = np.array([y[0], y[-1]])
    return Q, c, G, h, A, b


def special_version_qp(y, mode):
    """Solve `special_version_problem` with a general-purpose solver (SciPy's SLSQP with exact derivatives) -- independent of the
    product's active-set solution.  The reference calls cvxopt.solvers.qp [MASTestFunctions.py:348-349], which is not
    importable here (parity unpinned); both aim at the unique optimum of the same strictly convex problem."""
    from scipy.optimize import minimize
    Q, c, G, h, A, b = special_version_problem(y, mode)
    y = np.asarray(y, dtype=np.float64)
    res = minimize(lambda x: 0.5 * x @ Q @ x + c @ x, np.clip(y, 0, 1), jac=lambda x: Q @ x + c, method="SLSQP",
                   constraints=[{"type": "ineq", "fun": lambda x: h - G @ x, "jac": lambda x: -G},
                                {"type": "eq", "fun": lambda x: A @ x - b, "jac": lambda x: A}],
                   options={"maxiter": 1000, "ftol": 1e-15})
    return res.x


def kkt_residual(x, y, mode):
    """How far x is from satisfying the Karush-Kuhn-Tucker conditions of `special_version_problem` (necessary and sufficient: the
    problem is convex): max of the primal violations and of the stationarity residual |Qx + c + G^T lam + A^T nu| minimised over
    lam >= 0 supported on the constraints active at x (within 1e-9) and nu free -- itself a non-negative least-squares problem."""
    from scipy.optimize import nnls
    Q, c, G, h, A, b = special_version_problem(y, mode)
    primal = max(float(np.max(G @ x - h)), float(np.abs(A @ x - b).max()), 0.0)
    act = np.nonzero(G @ x - h >= -1e-9)[0]
    M = np.hstack([G[act].T, A.T, -A.T])
    _, stat = nnls(M, -(Q @ x + c), maxiter=50 * M.shape[1])
    return max(primal, float(stat))


def mas(logits_fn, img, sal, mode, step_size, substrate_fn, patch_mask=None, max_batch_size=50, order=None, special_version=False):
    """MASMetric.single_run (no CLIP) [MASTestFunctions.py:72-385].
    Returns (n_steps+1, corrected_scores, entropy, density_response, normalized_response)."""
    assert mode in ("del", "ins", "lerf", "morf")
    HW = img.shape[-1] * img.shape[-2]
    plan = Plan(HW, step_size, max_batch_size, patch_mask)
    n = plan.n_steps
    response = np.zeros(n + 1)
    ent = np.ones(n + 1)

    p_orig = _probe(logits_fn, img)
    target = int(np.argmax(p_orig))
    orig = float(p_orig[target])
    sub = np.asarray(substrate_fn(img), dtype=F32)
    p_sub = _probe(logits_fn, sub)
    base = float(p_sub[target])
    if mode == "ins":
        start, finish = sub, img
        response[0], ent[0] = base, entropy_bits(p_sub)
    else:
        start, finish = img, sub
        response[0], ent[0] = orig, entropy_bits(p_orig)

    groups, _ = flip_groups(sal, HW, plan, patch_mask, descending=(mode != "lerf"), order=order)
    logits, _ = run_steps(logits_fn, start, finish, groups, plan)
    p = softmax_rows(logits)
    response[1:] = p[:, target]
    ent[1:] = entropy_bits(p)

    # density: share of total attribution moved so far; float32 ratio added into float64
    flat = np.asarray(sal).reshape(HW)
    total = np.sum(flat.reshape(1, 1, HW))
    dens = np.zeros(n + 1)
    dens[0] = 0 if mode == "ins" else 1
    sign = 1 if mode == "ins" else -1
    for i, g in enumerate(groups):
        dens[i + 1] = dens[i] + sign * (np.sum(flat[g]) / total)

    norm = monotone(response, base, orig, falling=(mode != "ins"))
    if special_version:
        norm = special_version_qp(norm, mode)
    pen = np.abs(norm - dens)
    corr = norm - pen if mode == "ins" else norm + pen
    corr = corr.clip(0, 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        corr = (corr - corr.min()) / (corr.max() - corr.min())
    if np.isnan(corr).any():
        corr = np.linspace(1, 0, n + 1) if mode in ("del", "morf") else np.linspace(0, 1, n + 1)
    return n + 1, corr, ent, dens, norm


def rise_metric(logits_fn, img, sal, mode, step_size, substrate_fn, patch_mask=None, max_batch_size=50, order=None):
    """RISEMetric.single_run [RISETestFunctions.py:51-237] -> (n_steps+1, entropy, normalized)."""
    assert mode in ("del", "ins", "morf", "lerf")
    HW = img.shape[-1] * img.shape[-2]
    plan = Plan(HW, step_size, max_batch_size, patch_mask)
    n = plan.n_steps
    response, ent = np.zeros(n + 1), np.ones(n + 1)
    p_orig = _probe(logits_fn, img)
    target = int(np.argmax(p_orig))
    orig = float(p_orig[target])
    sub = np.asarray(substrate_fn(img), dtype=F32)
    p_sub = _probe(logits_fn, sub)
    base = float(p_sub[target])
    if mode == "ins":
        start, finish = sub, img
        response[0], ent[0] = base, entropy_bits(p_sub)
    else:
        start, finish = img, sub
        response[0], ent[0] = orig, entropy_bits(p_orig)
    groups, _ = flip_groups(sal, HW, plan, patch_mask, descending=(mode != "lerf"), order=order)
    logits, _ = run_steps(logits_fn, start, finish, groups, plan)
    p = softmax_rows(logits)
    response[1:] = p[:, target]
    ent[1:] = entropy_bits(p)
    return n + 1, ent, monotone(response, base, orig, falling=(mode != "ins"))


def aic(logits_fn, img, sal, mode, step_size, substrate_fn, patch_mask=None, max_batch_size=50, order=None,
        decision_flip=False):
    """AICMetric.single_run [AICTestFunctions.py:51-225]: the statistic is argmax == target."""
    assert mode in ("del", "ins")
    HW = img.shape[-1] * img.shape[-2]
    plan = Plan(HW, step_size, max_batch_size, patch_mask)
    n = plan.n_steps
    response = np.zeros(n + 1)
    target = int(np.argmax(logits_fn(img)[0]))
    orig = 1
    sub = np.asarray(substrate_fn(img), dtype=F32)
    base = int(int(np.argmax(logits_fn(sub)[0])) == target)
    if mode == "ins":
        start, finish = sub, img
        response[0] = base
    else:
        start, finish = img, sub
        response[0] = orig
    groups, _ = flip_groups(sal, HW, plan, patch_mask, descending=True, order=order)
    logits, _ = run_steps(logits_fn, start, finish, groups, plan)
    response[1:] = (np.argmax(logits, axis=1) == target) * 1
    if decision_flip:
        hit = np.nonzero(response == (0 if mode == "del" else 1))[0]
        return hit[0] / len(response), response
    with np.errstate(divide="ignore", invalid="ignore"):
        return n + 1, monotone(response, base, orig, falling=(mode == "del"))


def pnp(logits_fn, img, sal, mode, step_size, substrate_fn, patch_mask=None, max_batch_size=50, order=None):
    """PositiveNegativePerturbation.single_run [PosNegPertFunctions.py:31-175]: RAW response."""
    assert mode in ("lerf", "morf")
    HW = img.shape[-1] * img.shape[-2]
    plan = Plan(HW, step_size, max_batch_size, patch_mask)
    n = plan.n_steps
    response = np.zeros(n + 1)
    p_orig = _probe(logits_fn, img)
    target = int(np.argmax(p_orig))
    response[0] = float(p_orig[target])
    sub = np.asarray(substrate_fn(img), dtype=F32)
    logits_fn(sub)                                          # baseline probe, value unused in the return
    # lerf = the descending order flipped once more [PosNegPertFunctions.py:115-119]
    groups, _ = flip_groups(sal, HW, plan, patch_mask, descending=(mode == "morf"), order=order)
    logits, _ = run_steps(logits_fn, img, sub, groups, plan)
    response[1:] = softmax_rows(logits)[:, target]
    return n + 1, response


def mono(logits_fn, img, sal, mode, step_size, substrate_fn, patch_mask=None, max_batch_size=50, order=None):
    """MonotonicityMetric.single_run [MonotonicityTest.py:51-213] -> (raw response, spearman)."""
    assert mode in ("positive", "negative")
    HW = img.shape[-1] * img.shape[-2]
    plan = Plan(HW, step_size, max_batch_size, patch_mask, always_leftover=True)
    n = plan.n_steps
    response = np.zeros(n + 1)
    p_orig = _probe(logits_fn, img)
    target = int(np.argmax(p_orig))
    sub = np.asarray(substrate_fn(img), dtype=F32)
    p_sub = _probe(logits_fn, sub)
    if mode == "negative":
        start, finish = img, sub
        response[0] = float(p_orig[target])
    else:
        start, finish = sub, img
        response[0] = float(p_sub[target])
    groups, _ = flip_groups(sal, HW, plan, patch_mask, descending=True, order=order)
    logits, _ = run_steps(logits_fn, start, finish, groups, plan)
    response[1:] = softmax_rows(logits)[:, target]
    ramp = np.linspace(1, 0, n + 1) if mode == "negative" else np.linspace(0, 1, n + 1)
    return response, spearmanr(ramp, response).correlation


SWEEP_KEYS = ("MAS_ins", "MAS_del", "RISE_ins", "RISE_del", "AIC_ins", "AIC_del",
              "LERF_res", "MORF_res", "MONO_pos", "MONO_neg")


def run_perturbation(logits_fn, img, sal, step_size, blur_fn, max_batch_size=50):
    """The ten numbers of one image [evaluatePerturbation.py:448-497]; zeros_fn = zeros_like."""
    zeros = np.zeros_like
    _, mas_i, _, _, rise_i = mas(logits_fn, img, sal, "ins", step_size, blur_fn, None, max_batch_size)
    _, mas_d, _, _, rise_d = mas(logits_fn, img, sal, "del", step_size, zeros, None, max_batch_size)
    _, aic_i = aic(logits_fn, img, sal, "ins", step_size, blur_fn, None, max_batch_size)
    _, aic_d = aic(logits_fn, img, sal, "del", step_size, zeros, None, max_batch_size)
    _, lerf = pnp(logits_fn, img, sal, "lerf", step_size, zeros, None, max_batch_size)
    _, morf = pnp(logits_fn, img, sal, "morf", step_size, zeros, None, max_batch_size)
    _, mpos = mono(logits_fn, img, sal, "positive", step_size, blur_fn, None, max_batch_size)
    _, mneg = mono(logits_fn, img, sal, "negative", step_size, zeros, None, max_batch_size)
    return dict(zip(SWEEP_KEYS, (auc(mas_i), auc(mas_d), auc(rise_i), auc(rise_d), auc(aic_i), auc(aic_d),
                                 auc(lerf), auc(morf), mpos, mneg)))


def reference_fold(rows):
    """The reference's running result over the images of a sweep, restated with a plain dict
    [evaluatePerturbation.py:593-596: `pert_result_counter = run_perturbation(...)` for the first image,
    `pert_result_counter += run_perturbation(...)` afterwards].  `+=` on a collections.Counter adds the new
    counts and then deletes every key whose total is not > 0 (zero, negative, NaN); a key deleted earlier starts
    again from 0 and, being newly inserted, moves to the END of the iteration order the CSV loop (:612-615) follows.
    rows: per-image 10-vectors in SWEEP_KEYS order, in file order -> (surviving keys in CSV order, their sums)."""
    total = None
    for r in rows:
        c = {k: float(v) for k, v in zip(SWEEP_KEYS, r)}
        if total is None:
            total = c                               # first image: taken as is, non-positive entries included
            continue
        for k, v in c.items():
            total[k] = total.get(k, 0) + v
        for k in [k for k, v in total.items() if not v > 0]:
            del total[k]
    total = total or {}
    return list(total), list(total.values())
