"""RISE on the HIP kernels (reference util/attribution_methods/CLIP/generate_emap.py:65-101).

Bit-parity mode: the host draws the reference's NumPy RNG stream (one rand(N,s,s) block, then
two randint per mask) -- 64 B + 8 B per mask -- and the GPU does the rest: K4 builds each
masked batch in place (no N x H x W float64 mask tensor, no N x 3 x H x W masked tensor on the
host), K5 regenerates the masks while accumulating score-weighted sums in fp64.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import kernels as K
from .ig import hip_device


def draw_masks(input_size, N, s, p1, rng=np.random):
    """grid (N,s,s) uint8, shifts (N,2) int32 [row, col], cell (2,) -- generate_emap.py:66-76."""
    cell = np.ceil(np.array(input_size) / s)
    grid = (rng.rand(N, s, s) < p1).astype(np.uint8)
    shifts = np.empty((N, 2), dtype=np.int32)
    for i in range(N):
        shifts[i, 0] = rng.randint(0, cell[0])
        shifts[i, 1] = rng.randint(0, cell[1])
    return grid, shifts, cell.astype(np.int64)


def draw_masks_on_device(input_size, N, s, p1, device, generator=None):
    """The same draw from the DEVICE generator (torch's Philox): grid (N,s,s) uint8 and shifts (N,2) int32 as device tensors,
    cell (2,) on the host.  Not the reference's NumPy stream -- statistically equivalent masks, no host RNG loop and no
    upload ("RISE RNG": bit-parity mode = `draw_masks`, performance mode = this)."""
    dev = hip_device(device)
    cell = np.ceil(np.array(input_size) / s).astype(np.int64)
    grid = (torch.rand((N, s, s), device=dev, generator=generator) < p1).to(torch.uint8)
    shifts = torch.stack([torch.randint(0, int(cell[0]), (N,), device=dev, generator=generator),
                          torch.randint(0, int(cell[1]), (N,), device=dev, generator=generator)], dim=1).to(torch.int32)
    return grid, shifts, cell


def generate_masks(input_size, N, s, p1, device=None):
    """(N,1,H,W) masks.  Same RNG stream and values as the reference; returned as float32 on the
    HIP device instead of a float64 host tensor (documented divergence)."""
    dev = hip_device(device if device is not None else "cuda")
    grid, shifts, cell = draw_masks(tuple(int(v) for v in input_size), N, s, p1)
    image = torch.zeros((1, int(input_size[0]), int(input_size[1])), dtype=torch.float32, device=dev)
    masks = K.rise_apply(torch.from_numpy(grid).to(dev), torch.from_numpy(shifts).to(dev), cell, image,
                         want_masked=False, want_masks=True)
    return masks.unsqueeze(1)


def rise(model, image, txt_embedding, device, N=2000, s=8, p1=0.5, *, score_fn=None, batch_size=50,
         masks=None, mask_range=None, return_partial=False, streams=1):
    """Saliency (H,W) float32 on the device = sum_i score_i * mask_i / N / p1.

    Reference call shape: rise(model, image, txt_embedding, device, N, s, p1) with the CLIP cosine
    score.  Extensions (keyword-only): `score_fn(batch) -> (B,)` for a plain classifier,
    `masks=(grid, shifts, cell)` to reuse a draw, `mask_range=(lo, hi)` to process a shard of the
    masks (multi-GPU), `return_partial` to get the un-rounded fp64 partial sum for an all-reduce,
    `streams` > 1 to queue consecutive mask batches round-robin on that many HIP streams, each driven by its own host thread
    (xai_engine/streams.py) with its own batch buffer; the batches write disjoint score slices, so the map is bit-identical to
    `streams=1`.
    """
    dev = hip_device(device)
    H, W = int(image.shape[-2]), int(image.shape[-1])
    grid, shifts, cell = masks if masks is not None else draw_masks((H, W), N, s, p1)
    lo, hi = mask_range if mask_range is not None else (0, N)
    img = image.to(dev, torch.float32).reshape(-1, H, W).contiguous()
    if torch.is_tensor(grid):                                   # a device draw (draw_masks_on_device)
        g_all, sh_all = grid[lo:hi].to(dev).contiguous(), shifts[lo:hi].to(dev).contiguous()
    else:
        g_all = torch.from_numpy(np.ascontiguousarray(grid[lo:hi])).to(dev)
        sh_all = torch.from_numpy(np.ascontiguousarray(shifts[lo:hi])).to(dev)
    n = hi - lo
    scores = torch.empty(n, dtype=torch.float32, device=dev)
    spans = [(i, min(i + batch_size, n)) for i in range(0, n, batch_size)]
    from .streams import on_worker
    n_streams = 1 if on_worker() else max(1, min(int(streams), len(spans)))
    bufs = [torch.empty((min(batch_size, max(n, 1)),) + tuple(img.shape), dtype=torch.float32, device=dev) for _ in range(n_streams)]

    def one_batch(i, j, buf):
        masked = K.rise_apply(g_all[i:j], sh_all[i:j], cell, img, out=buf[:j - i])
        if score_fn is not None:
            scores[i:j] = score_fn(masked).reshape(-1).float()
        else:
            feats = F.normalize(model.encode_image(masked), dim=-1)
            scores[i:j] = (feats @ txt_embedding.T).reshape(-1).float()

    def job(k, i, j):
        with torch.no_grad():                                   # grad mode is thread-local: a stream worker starts with it enabled
            one_batch(i, j, bufs[k % n_streams])

    if n_streams == 1:
        for k, (i, j) in enumerate(spans):
            job(k, i, j)
    else:
        from .streams import run_on_streams                     # one host thread per stream (streams.py); forward only
        run_on_streams(dev, n_streams, [lambda k=k, i=i, j=j: job(k, i, j) for k, (i, j) in enumerate(spans)],
                       kind=("rise", id(model), batch_size, tuple(img.shape)))
    acc = torch.zeros((H, W), dtype=torch.float64, device=dev)
    if n > 0:
        K.rise_accum(g_all, sh_all, scores, cell, H, W, 1.0 / N / p1, acc=acc)
    return acc if return_partial else acc.float()
