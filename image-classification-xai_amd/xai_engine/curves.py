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


def shape_constrained_fit(y, mode, return_multipliers=False):
    """`special_version=True` of MASMetric.single_run (reference MASTestFunctions.py:311-350): the least-squares curve x closest to
    the normalised response y with x[0] = y[0], x[-1] = y[-1], 0 <= x <= 1 and non-negative second differences for 'del' (convex),
    non-positive ones for 'ins' (concave); for 'morf' / 'lerf' the reference's shape rows stay zero, leaving the box and the end points.

    The reference hands the dense QP (225 variables, 673 inequality rows) to cvxopt's interior-point solver (not importable here: parity
    unpinned; its answer is the optimum to its default tolerances, ~1e-7).  The optimum is unique -- the objective is strictly convex --
    so it is computed exactly instead: end points eliminated, the least-distance problem  min |z - y_mid|  s.t.  G z <= h  solved through
    its non-negative least-squares dual (Lawson & Hanson's LDP: an active-set method, finite, exact up to rounding), then the active set it
    finds is re-solved as an equality-constrained problem (one Cholesky-sized solve) and the Karush-Kuhn-Tucker conditions of the ORIGINAL
    problem are checked; tests/test_cpu_host.py holds the KKT residual to 1e-9.  Host-only float64 arithmetic on <= 225 numbers.
    A response containing NaN (the blurred / black image scored exactly like the input: 0/0 in the normalisation) is returned
    unchanged, so that the reference's NaN guard downstream (:363-368) takes over as it does without special_version."""
    from scipy.optimize import nnls
    y = np.asarray(y, dtype=np.float64)
    n = y.shape[0]
    x = y.copy()
    if not np.isfinite(y).all():
        return (x, None) if return_multipliers else x
    if n <= 2 or mode not in ("del", "ins"):
        x[1:-1] = np.clip(y[1:-1], 0.0, 1.0)                # box + fixed end points: separable
        return (x, None) if return_multipliers else x
    m = n - 2
    sgn = 1.0 if mode == "del" else -1.0
    # shape rows over x = (y0, z, y_end):  sgn * (-x[i] + 2 x[i+1] - x[i+2]) <= 0,  i = 0 .. n-3, written over z with the end points on the right
    S = np.zeros((m, m))
    rhs = np.zeros(m)
    i = np.arange(m)
    S[i, i] = 2.0 * sgn                                     # x[i+1] = z[i]
    S[i[1:], i[1:] - 1] = -sgn                              # x[i]   = z[i-1]   (i >= 1)
    S[i[:-1], i[:-1] + 1] = -sgn                            # x[i+2] = z[i+1]   (i <= m-2)
    rhs[0] += sgn * y[0]
    rhs[-1] += sgn * y[-1]
    G = np.vstack([-np.eye(m), np.eye(m), S])               # -z <= 0,  z <= 1,  shape
    h = np.concatenate([np.zeros(m), np.ones(m), rhs])
    ym = y[1:-1]
    d = h - G @ ym                                          # G u <= d  for  u = z - ym
    if (d >= 0).all():                                      # y itself is feasible: it is the optimum
        return (x, np.zeros(len(h))) if return_multipliers else x
    # LDP (Lawson & Hanson, ch. 23): (-G) u >= -d;  E = [(-G)^T ; (-d)^T],  f = e_{m+1};  w = argmin_{w >= 0} |E w - f|;  u = -r[:m] / r[m]
    E = np.vstack([-G.T, -d[None, :]])
    f = np.zeros(m + 1)
    f[m] = 1.0
    w, _ = nnls(E, f, maxiter=20 * E.shape[1])
    r = E @ w - f
    if not abs(r[m]) > 1e-14:
        raise ValueError("special_version: the shape constraints are infeasible for this response (cannot happen for end points in [0, 1])")
    z = ym - r[:m] / r[m]
    lam = 2.0 * w / (-r[m])                                  # multipliers of G z <= h for the objective |z - ym|^2
    # polish on the active set the dual found: z = ym - G_A^T mu with (G_A G_A^T) mu = G_A ym - h_A
    act = np.nonzero(w > 0)[0]
    if act.size:
        GA, hA = G[act], h[act]
        mu, *_ = np.linalg.lstsq(GA @ GA.T, GA @ ym - hA, rcond=None)
        zp = ym - GA.T @ mu
        if (mu >= -1e-12).all() and (G @ zp - h <= 1e-12).all():
            z, lam = zp, np.zeros(len(h))
            lam[act] = 2.0 * np.maximum(mu, 0.0)
    x[1:-1] = z
    return (x, lam) if return_multipliers else x


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
