"""Oracle: RISE masks + score-weighted accumulation (test infrastructure only).

Restates util/attribution_methods/CLIP/generate_emap.py generate_masks :65-81, rise :85-101.

PARITY UNPINNED at one boundary: the reference up-samples the binary grid with
`skimage.transform.resize(grid, up, order=1, mode='reflect', anti_aliasing=False)`;
skimage is absent from this image and the reference's requirements.txt pins the
non-existent `skimage==0.0`.  Released skimage (>= 0.19) implements that call as
`scipy.ndimage.zoom(grid, up/in, order=1, mode='mirror', grid_mode=True)` on float32 data,
which is what `upsample_grid` calls; `upsample_grid_formula` spells the same arithmetic
out (it is the formula the HIP kernel mirrors) and the tests hold the two together.
"""
import numpy as np
from scipy import ndimage

F32 = np.float32


def draw_grid_and_shifts(input_size, N, s, p1, rng=np.random):
    """The reference's RNG stream, in its order: one rand(N,s,s) block, then for every mask
    randint(0,cell_h) and randint(0,cell_w) [generate_emap.py:66-76]."""
    cell = np.ceil(np.array(input_size) / s)
    grid = (rng.rand(N, s, s) < p1).astype(F32)
    shifts = np.empty((N, 2), dtype=np.int32)
    for i in range(N):
        shifts[i, 0] = rng.randint(0, cell[0])
        shifts[i, 1] = rng.randint(0, cell[1])
    return grid, shifts, cell.astype(np.int64)


def upsample_grid(g, up):
    """skimage.transform.resize(g, up, order=1, mode='reflect', anti_aliasing=False): ndimage.zoom with the
    'mirror' boundary on the half-pixel grid, then clipped to the input's value range (resize's clip=True)."""
    out = ndimage.zoom(g.astype(F32), (up[0] / g.shape[0], up[1] / g.shape[1]), order=1, mode="mirror", grid_mode=True)
    return np.clip(out, g.min(), g.max())


def taps_1d(n_in, n_out):
    """Per output index: (i0, i1, t) with value = (1-t)*g[i0] + t*g[i1]; coordinate
    (j + 0.5) * n_in/n_out - 0.5, mirrored about 0, neighbour mirrored about n_in-1."""
    j = np.arange(n_out, dtype=np.float64)
    c = (j + 0.5) * (n_in / n_out) - 0.5
    c = np.where(c < 0, -c, c)
    i0 = np.floor(c).astype(np.int64)
    t = c - i0
    i1 = i0 + 1
    i1 = np.where(i1 >= n_in, 2 * n_in - 2 - i1, i1)
    return i0, i1, t


def upsample_grid_formula(g, up):
    r0, r1, tr = taps_1d(g.shape[0], int(up[0]))
    c0, c1, tc = taps_1d(g.shape[1], int(up[1]))
    g = g.astype(np.float64)
    wr0, wr1 = (1 - tr)[:, None], tr[:, None]
    wc0, wc1 = (1 - tc)[None, :], tc[None, :]
    out = (g[r0][:, c0] * (wr0 * wc0) + g[r0][:, c1] * (wr0 * wc1)
           + g[r1][:, c0] * (wr1 * wc0) + g[r1][:, c1] * (wr1 * wc1))
    return np.clip(out, g.min(), g.max()).astype(F32)


def masks_from(grid, shifts, input_size, cell):
    """(N,1,H,W) float64 holding float32-valued masks [generate_emap.py:72-81]."""
    N, s, _ = grid.shape
    up = (s + 1) * cell
    H, W = input_size
    out = np.empty((N, H, W))
    for i in range(N):
        x, y = shifts[i]
        out[i] = upsample_grid(grid[i], up)[x:x + H, y:y + W]
    return out.reshape(N, 1, H, W)


def rise(score_fn, image, N, s, p1, grid, shifts, cell, batch=50):
    """score_fn(batch (B,C,H,W) f32) -> (B,) scores.  sal = sum_i score_i * mask_i / N / p1
    in float64 [generate_emap.py:85-101]."""
    H, W = image.shape[-2:]
    masks = masks_from(grid, shifts, (H, W), cell)
    masked = image.astype(np.float64) * masks                       # (N,C,H,W) float64
    preds = []
    for i in range(0, N, batch):
        preds.append(np.asarray(score_fn(masked[i:i + batch].astype(F32)), dtype=F32).reshape(-1, 1))
    preds = np.concatenate(preds)
    sal = (preds * masks.reshape(N, -1)).sum(0).reshape(H, W)
    return sal / N / p1
