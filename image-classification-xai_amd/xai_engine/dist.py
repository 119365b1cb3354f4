"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm, xGMI underneath;
"gloo" for the CPU rehearsal tests).  The hot path shards embarrassingly:

  images      -> sweep.sweep_images: round-robin image ownership, ONE all-reduce(SUM) of an
                 12-element fp64 vector (96 B) at the end            [evaluatePerturbation.py:594-618]
  RISE masks  -> rise_sharded: contiguous mask ranges, ONE all-reduce(SUM) of the (H,W) fp64
                 partial map (401 KB at 224x224)                      [generate_emap.py:93-100]

  IG steps    -> ig_step_sharded (latency, not throughput): contiguous step ranges of ONE image, an
                 all-gather of the per-step logits (Left-IG's cutoff needs all of them) and ONE
                 all-reduce(SUM) of the (C,H,W) partial sum (602 KB)   [saliencyMethods.py:40-70]

All messages are latency-bound (<= 0.6 MB); with 7 direct xGMI links per GPU RCCL's default
algorithm is already one hop per peer, so nothing is tuned beyond "one collective".
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, world, device).  Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* set by torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    # rehearsal knobs (same as bench.py's): XAI_DIST_BACKEND=gloo + XAI_FORCE_DEVICE=0 let several ranks share one GPU
    backend = backend or os.environ.get("XAI_DIST_BACKEND")
    local = int(os.environ.get("XAI_FORCE_DEVICE", local))
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if use_gpu else "gloo")
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def mask_range(n_masks, rank, world):
    """Contiguous, balanced [lo, hi) of masks for `rank`."""
    base, extra = divmod(n_masks, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_masks(masks, device, src=0):
    """Make every rank use rank `src`'s RNG draw (grid uint8 (N,s,s), shifts int32 (N,2), cell (2,)): ONE broadcast of
    the three arrays packed into a byte buffer (N = 8000, s = 8: 576 016 B).  Every rank passes a draw of the same
    (N, s) -- its own is simply overwritten -- so the packed size is known everywhere without a size exchange.
    A NumPy draw (`rise.draw_masks`, the reference's RNG stream) comes back as NumPy arrays; a device draw
    (`rise.draw_masks_on_device`: torch tensors) is packed, broadcast and unpacked on the device and comes back as tensors
    there -- no host round trip under RCCL (gloo stages through the host by itself)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return masks
    on = device if dist.get_backend() == "nccl" else torch.device("cpu")
    grid, shifts, cell = masks
    cell = np.asarray(cell, dtype=np.int64).reshape(2)
    if torch.is_tensor(grid) or torch.is_tensor(shifts):
        home = grid.device if torch.is_tensor(grid) else shifts.device
        g = torch.as_tensor(grid).to(home, torch.uint8).contiguous()
        sh = torch.as_tensor(shifts).to(home, torch.int32).contiguous()
        c = torch.from_numpy(cell.copy()).to(home)
        packed = torch.cat([g.reshape(-1), sh.reshape(-1).view(torch.uint8), c.view(torch.uint8)]).to(on)
        dist.broadcast(packed, src=src)
        got = packed.to(home)
        ng, ns = g.numel(), sh.numel() * 4
        return (got[:ng].reshape(g.shape).clone(), got[ng:ng + ns].clone().view(torch.int32).reshape(sh.shape),
                got[ng + ns:].clone().view(torch.int64).cpu().numpy())
    grid = np.ascontiguousarray(grid, dtype=np.uint8)
    shifts = np.ascontiguousarray(shifts, dtype=np.int32)
    packed = np.concatenate([grid.reshape(-1), shifts.reshape(-1).view(np.uint8), cell.view(np.uint8)])
    t = torch.from_numpy(packed).to(on)
    dist.broadcast(t, src=src)
    got = t.cpu().numpy()
    ng, ns = grid.size, shifts.size * 4
    return (got[:ng].reshape(grid.shape).copy(), got[ng:ng + ns].copy().view(np.int32).reshape(shifts.shape),
            got[ng + ns:].copy().view(np.int64))


def all_reduce_sum(t):
    """In-place SUM across ranks (no-op for a single process); gloo needs a CPU tensor."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and t.is_cuda:
            c = t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def rise_sharded(model, image, txt_embedding, device, N=2000, s=8, p1=0.5, *, score_fn=None, batch_size=50, masks=None, streams=1):
    """RISE with the N masks split over the ranks: every rank scores its contiguous range and
    accumulates an fp64 partial; one all-reduce merges them.  Returns the (H,W) fp32 map on
    every rank.  All ranks use rank 0's mask draw."""
    from .rise import draw_masks, rise
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    H, W = int(image.shape[-2]), int(image.shape[-1])
    if masks is None:
        masks = draw_masks((H, W), N, s, p1)
    masks = broadcast_masks(masks, torch.device(device))
    part = rise(model, image, txt_embedding, device, N=N, s=s, p1=p1, score_fn=score_fn, batch_size=batch_size, masks=masks,
                mask_range=mask_range(N, rank, world), return_partial=True, streams=streams)
    return all_reduce_sum(part).float()


def step_range(n_steps, rank, world):
    """Contiguous, balanced [lo, hi) of path steps for `rank` (same split as mask_range)."""
    return mask_range(n_steps, rank, world)


def ig_step_sharded(input, model, steps, alpha_star, baseline, device, target_class):
    """IG / Left-IG of ONE image with the path steps split over the ranks: every rank
    interpolates and back-propagates its contiguous step range, the per-step logits are all-gathered
    (Left-IG's cutoff is global), each rank sums the gradients of its steps below the cutoff with the
    streaming kernel, one all-reduce merges the partial sums and xai_ig_finish_f32 applies
    / n_use * (x - baseline).  Returns (C,H,W) on every rank.  The sum is associated per rank, so the
    result equals the single-process IG to rounding (<= 1e-6 relative), not bitwise."""
    from . import kernels as K
    from .ig import _prep, _uniform_alphas, getGradientsParallel
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    dev, x, base = _prep(input, baseline, device)
    alphas = _uniform_alphas(steps, dev)
    lo, hi = step_range(steps, rank, world)
    n_mine = hi - lo
    logits = torch.zeros(steps, dtype=torch.float32, device=dev)
    grads = None
    if n_mine > 0:
        imgs = K.ig_interp(x, base, alphas[lo:hi])[0].requires_grad_(True)
        g, s = getGradientsParallel(imgs, model, target_class)
        grads = g.reshape((n_mine,) + tuple(x.shape[1:])).contiguous()
        logits[lo:hi] = s.reshape(-1)
    all_reduce_sum(logits)                                   # disjoint supports: a sum is an all-gather
    n_use = steps if alpha_star == 1 else int(K.ig_cutoff(logits.reshape(1, steps), alpha_star)[0])
    acc = torch.zeros_like(x)
    cnt = max(0, min(hi, n_use) - lo)
    if cnt > 0:
        K.ig_accum_add(grads[:cnt], acc[0])
    all_reduce_sum(acc)
    return K.ig_finish(acc, n_use, x, base)[0]
