"""Torch-tensor front end of the C ABI (include/xai_hip.h): shape/device checks, output
allocation, stream plumbing.  Every function launches asynchronously on torch's current
stream of the tensors' device and returns device tensors; nothing here computes on the CPU.
"""
import torch

from . import _lib

F32, I32 = torch.float32, torch.int32


def _need(t, dtype, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise _lib.XaiHipError(f"{name} lives on '{t.device}': the xai_engine kernels run on a HIP device only "
                               "(pass device='cuda:N'); there is no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _call(name, dev, *args):
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(getattr(lib, name)(*args, _stream(dev)), name)


def _base_args(baseline, like, name="baseline"):
    """tensor baseline -> (ptr, 0.0); python scalar -> (None, value)."""
    if isinstance(baseline, torch.Tensor):
        _need(baseline, F32, name)
        if baseline.numel() != like.numel():
            raise ValueError(f"{name} has {baseline.numel()} elements, expected {like.numel()}")
        return baseline, 0.0
    return None, float(baseline)


# ------------------------------------------------------------------------------ IG
def ig_interp(x, baseline, alphas, out=None):
    """x: (n_img, *img) ; alphas: (n_alpha,) shared or (n_img, n_alpha) -> (n_img, n_alpha, *img)."""
    _need(x, F32, "x"); _need(alphas, F32, "alphas")
    n_img = x.shape[0]
    n_elem = x[0].numel()
    if alphas.dim() == 1:
        n_alpha, stride = alphas.shape[0], 0
    else:
        if alphas.shape[0] != n_img:
            raise ValueError("alphas must be (n_alpha,) or (n_img, n_alpha)")
        n_alpha, stride = alphas.shape[1], alphas.shape[1]
    b, bs = _base_args(baseline, x)
    if out is None:
        out = torch.empty((n_img, n_alpha) + tuple(x.shape[1:]), dtype=F32, device=x.device)
    else:
        _need(out, F32, "out")
        if out.numel() != n_img * n_alpha * n_elem:
            raise ValueError("out has the wrong size")
    _call("xai_ig_interp_f32", x.device, _ptr(x), _ptr(b), bs, _ptr(alphas), stride, n_img, n_alpha, n_elem, _ptr(out))
    return out


def ig_cutoff(logits, alpha_star):
    """logits (n_img, n_steps) -> int32 (n_img,) number of leading steps Left-IG averages."""
    _need(logits, F32, "logits")
    n_img, n_steps = logits.shape
    n_use = torch.empty(n_img, dtype=I32, device=logits.device)
    _call("xai_ig_cutoff_f32", logits.device, _ptr(logits), n_img, n_steps, float(alpha_star), _ptr(n_use))
    return n_use


def ig_accum(grads, x, baseline, n_use=None, w1=None, w2=None, want_abs=False, timing_events=None):
    """grads (n_img, n_steps, C, H, W); x (n_img, C, H, W) -> out (n_img, C, H, W)[, abs (n_img, H, W)].
    n_use: None (all steps), int, or int32 device tensor (n_img,).
    timing_events: optional (start, stop) pair of torch.cuda.Event(enable_timing=True): the kernel's own start / stop
    timestamps are recorded into them by the dispatch (xai_ig_accum_timed_f32), start.elapsed_time(stop) is the kernel time."""
    _need(grads, F32, "grads"); _need(x, F32, "x")
    n_img, n_steps, Cc = grads.shape[0], grads.shape[1], grads.shape[2]
    hw = grads[0, 0, 0].numel()
    if x.numel() != n_img * Cc * hw:
        raise ValueError("x does not match grads")
    b, bs = _base_args(baseline, x)
    n_dev, n_host = None, n_steps
    if isinstance(n_use, torch.Tensor):
        n_dev = _need(n_use, I32, "n_use")
        if n_dev.numel() != n_img:
            raise ValueError("n_use tensor must have one entry per image")
    elif n_use is not None:
        n_host = int(n_use)
    for w, nm in ((w1, "w1"), (w2, "w2")):
        if w is not None:
            _need(w, F32, nm)
            if w.numel() != n_img * n_steps:
                raise ValueError(f"{nm} must be (n_img, n_steps)")
    out = torch.empty((n_img,) + tuple(grads.shape[2:]), dtype=F32, device=x.device)
    out_abs = torch.empty((n_img,) + tuple(grads.shape[3:]), dtype=F32, device=x.device) if want_abs else None
    if timing_events is None:
        _call("xai_ig_accum_f32", x.device, _ptr(grads), n_img, n_steps, _ptr(n_dev), n_host, _ptr(w1), _ptr(w2), _ptr(x), _ptr(b), bs,
              Cc, hw, _ptr(out), _ptr(out_abs))
    else:
        e0, e1 = timing_events
        stream = torch.cuda.current_stream(x.device)
        for e in (e0, e1):                         # torch creates the hipEvent_t on the first record; the dispatch re-records it
            e.record(stream)
        _call("xai_ig_accum_timed_f32", x.device, _ptr(grads), n_img, n_steps, _ptr(n_dev), n_host, _ptr(w1), _ptr(w2), _ptr(x), _ptr(b),
              bs, Cc, hw, _ptr(out), _ptr(out_abs), e0.cuda_event, e1.cuda_event)
    return (out, out_abs) if want_abs else out


def store_grads(src, dst):
    """dst <- src (same element count, both contiguous) with streaming non-temporal stores."""
    _need(src, F32, "src"); _need(dst, F32, "dst")
    if src.numel() != dst.numel():
        raise ValueError("src and dst differ in size")
    _call("xai_ig_store_grads_f32", dst.device, _ptr(src), _ptr(dst), src.numel())
    return dst


def ig_accum_add(grads, acc):
    """acc (N elems) += sum over rows of grads (n_batch, N elems)."""
    _need(grads, F32, "grads"); _need(acc, F32, "acc")
    n_batch = grads.shape[0]
    if grads[0].numel() != acc.numel():
        raise ValueError("acc does not match one row of grads")
    _call("xai_ig_accum_add_f32", acc.device, _ptr(grads), n_batch, _ptr(acc), acc.numel())
    return acc


def ig_finish(acc, n_steps, x, baseline, want_abs=False):
    """acc, x: (n_img, C, H, W) -> acc / n_steps * (x - baseline)."""
    _need(acc, F32, "acc"); _need(x, F32, "x")
    n_img, Cc = x.shape[0], x.shape[1]
    hw = x[0, 0].numel()
    b, bs = _base_args(baseline, x)
    out = torch.empty_like(x)
    out_abs = torch.empty((n_img,) + tuple(x.shape[2:]), dtype=F32, device=x.device) if want_abs else None
    _call("xai_ig_finish_f32", x.device, _ptr(acc), n_img, int(n_steps), _ptr(x), _ptr(b), bs, Cc, hw, _ptr(out), _ptr(out_abs))
    return (out, out_abs) if want_abs else out


def sumsq(rows):
    """(n_rows, ...) -> (n_rows,) sum of squares per row."""
    _need(rows, F32, "rows")
    out = torch.empty(rows.shape[0], dtype=F32, device=rows.device)
    _call("xai_sumsq_f32", rows.device, _ptr(rows), rows.shape[0], rows[0].numel(), _ptr(out))
    return out


def idgi_accum(grads, logits, sq):
    """grads (n_steps, *img), logits (n_steps,), sq (n_steps,) -> (*img)."""
    _need(grads, F32, "grads"); _need(logits, F32, "logits"); _need(sq, F32, "sumsq")
    out = torch.empty(tuple(grads.shape[1:]), dtype=F32, device=grads.device)
    _call("xai_idgi_accum_f32", grads.device, _ptr(grads), grads.shape[0], _ptr(logits), _ptr(sq), grads[0].numel(), _ptr(out))
    return out


# ------------------------------------------------------------------------------ Grad-CAM
def gradcam(act, grad, relu=True):
    """(B,C,h,w) x2 -> (B,h,w)."""
    _need(act, F32, "act"); _need(grad, F32, "grad")
    if act.shape != grad.shape or act.dim() != 4:
        raise ValueError("act and grad must both be (B,C,h,w)")
    B, Cc, h, w = act.shape
    cam = torch.empty((B, h, w), dtype=F32, device=act.device)
    nbytes = _lib.load().xai_gradcam_workspace_bytes(B, Cc, h, w)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=act.device) if nbytes else None
    _call("xai_gradcam_f32", act.device, _ptr(act), _ptr(grad), B, Cc, h, w, int(bool(relu)), _ptr(cam), _ptr(ws), nbytes)
    return cam


def bilinear_up(src, H, W, scale=1.0, take_abs=False):
    """(B,h,w) -> (B,H,W), align_corners=False."""
    _need(src, F32, "src")
    B, h, w = src.shape
    dst = torch.empty((B, H, W), dtype=F32, device=src.device)
    _call("xai_bilinear_up_f32", src.device, _ptr(src), B, h, w, int(H), int(W), float(scale), int(bool(take_abs)), _ptr(dst))
    return dst


# ------------------------------------------------------------------------------ RISE
def rise_apply(grid, shift, cell, image, want_masked=True, want_masks=False, out=None):
    """grid (n,s,s) uint8, shift (n,2) int32, image (C,H,W) -> masked (n,C,H,W) and/or masks (n,H,W)."""
    _need(grid, torch.uint8, "grid"); _need(shift, I32, "shift"); _need(image, F32, "image")
    n, s = grid.shape[0], grid.shape[1]
    Cc, H, W = image.shape
    masked = None
    if want_masked:
        masked = out if out is not None else torch.empty((n, Cc, H, W), dtype=F32, device=image.device)
        _need(masked, F32, "out")
    masks = torch.empty((n, H, W), dtype=F32, device=image.device) if want_masks else None
    _call("xai_rise_apply_f32", image.device, _ptr(grid), _ptr(shift), n, s, int(cell[0]), int(cell[1]), _ptr(image), Cc, H, W,
          _ptr(masked), _ptr(masks))
    if want_masked and want_masks:
        return masked, masks
    return masked if want_masked else masks


def rise_accum(grid, shift, scores, cell, H, W, scale, acc=None):
    """acc (H,W) float64 += scale * sum_n scores[n] * mask_n."""
    _need(grid, torch.uint8, "grid"); _need(shift, I32, "shift"); _need(scores, F32, "scores")
    n, s = grid.shape[0], grid.shape[1]
    if scores.numel() != n:
        raise ValueError("one score per mask")
    if acc is None:
        acc = torch.zeros((H, W), dtype=torch.float64, device=grid.device)
    else:
        _need(acc, torch.float64, "acc")
    _call("xai_rise_accum_f64", grid.device, _ptr(grid), _ptr(shift), _ptr(scores), n, s, int(cell[0]), int(cell[1]), int(H), int(W),
          float(scale), _ptr(acc))
    return acc


# ------------------------------------------------------------------------------ ins/del
def rank(sal):
    """sal (n_seg, hw) -> (order, rank) int32: stable ascending argsort and its inverse."""
    _need(sal, F32, "sal")
    n_seg, hw = sal.shape
    lib = _lib.load()
    ws = torch.empty(lib.xai_rank_workspace_bytes(n_seg, hw), dtype=torch.uint8, device=sal.device)
    order = torch.empty((n_seg, hw), dtype=I32, device=sal.device)
    rk = torch.empty((n_seg, hw), dtype=I32, device=sal.device)
    _call("xai_rank_f32", sal.device, _ptr(sal), n_seg, hw, _ptr(order), _ptr(rk), _ptr(ws), ws.numel())
    return order, rk


def flip_steps(rk, descending, step_size):
    """rank (hw,) int32 -> flip step per pixel (hw,) int32."""
    _need(rk, I32, "rank")
    out = torch.empty_like(rk)
    _call("xai_flip_steps_i32", rk.device, _ptr(rk), rk.numel(), int(bool(descending)), int(step_size), _ptr(out))
    return out


def perturb_batch(start, finish, flip, first_step, n_batch, out=None):
    """start, finish (C,H,W); flip (H*W,) int32 -> (n_batch, C, H, W)."""
    _need(start, F32, "start"); _need(finish, F32, "finish"); _need(flip, I32, "flip_step")
    Cc = start.shape[0]
    hw = start[0].numel()
    if finish.shape != start.shape or flip.numel() != hw:
        raise ValueError("start/finish/flip_step shapes disagree")
    if out is None:
        out = torch.empty((n_batch,) + tuple(start.shape), dtype=F32, device=start.device)
    else:
        _need(out, F32, "out")
        if out.numel() != n_batch * Cc * hw:
            raise ValueError("out has the wrong size")
    _call("xai_perturb_batch_f32", start.device, _ptr(start), _ptr(finish), _ptr(flip), Cc, hw, int(first_step), int(n_batch), _ptr(out))
    return out


def segment_sums(sal, order, descending, step_size, n_steps):
    """-> (seg (n_steps,), total (1,)) float32."""
    _need(sal, F32, "sal"); _need(order, I32, "order")
    seg = torch.empty(n_steps, dtype=F32, device=sal.device)
    total = torch.empty(1, dtype=F32, device=sal.device)
    _call("xai_segment_sums_f32", sal.device, _ptr(sal), _ptr(order), sal.numel(), int(bool(descending)), int(step_size), int(n_steps),
          _ptr(seg), _ptr(total))
    return seg, total


def blur_sep(x, k1d):
    """x (B,C,H,W), k1d (klen,) on device -> zero-padded separable blur.  Up to 63 taps both passes run in
    one launch (tile + halo in LDS); longer kernels take two 1-D passes through a scratch tensor."""
    _need(x, F32, "x"); _need(k1d, F32, "k1d")
    B, Cc, H, W = x.shape
    out = torch.empty_like(x)
    klen = k1d.numel()
    if klen <= 63:
        _call("xai_blur_sep_f32", x.device, _ptr(x), _ptr(k1d), klen, B, Cc, H, W, _ptr(out))
    else:
        tmp = torch.empty_like(x)
        _call("xai_blur_1d_f32", x.device, _ptr(x), _ptr(k1d), klen, 1, B, Cc, H, W, _ptr(tmp))
        _call("xai_blur_1d_f32", x.device, _ptr(tmp), _ptr(k1d), klen, 0, B, Cc, H, W, _ptr(out))
    return out


def softmax_stats(logits, target=None, want_entropy=True, want_argmax=True, out=None, offset=0):
    """logits (B,K); target: None (row argmax), int, or int32 device tensor (1,).
    -> (p_target (B,), entropy_bits (B,) or None, argmax (B,) int32 or None).
    `out=(p, entropy, argmax)` (1-D fp32/fp32/int32 device tensors) writes rows [offset, offset+B) of
    preallocated curves instead of allocating (the ins/del loop fills its curves in place)."""
    _need(logits, F32, "logits")
    B, K = logits.shape
    t_dev, t_host = None, -1
    if isinstance(target, torch.Tensor):
        t_dev = _need(target, I32, "target")
    elif target is not None:
        t_host = int(target)
    if out is None:
        p = torch.empty(B, dtype=F32, device=logits.device)
        ent = torch.empty(B, dtype=F32, device=logits.device) if want_entropy else None
        am = torch.empty(B, dtype=I32, device=logits.device) if want_argmax else None
    else:
        p, ent, am = out
        _need(p, F32, "out p")
        if ent is not None:
            _need(ent, F32, "out entropy")
        if am is not None:
            _need(am, I32, "out argmax")
        for t in (p, ent, am):
            if t is not None and (t.dim() != 1 or offset < 0 or offset + B > t.numel()):
                raise ValueError("out tensors must be 1-D with room for rows [offset, offset+B)")
        p, ent, am = p[offset:offset + B], (None if ent is None else ent[offset:offset + B]), (None if am is None else am[offset:offset + B])
    _call("xai_softmax_stats_f32", logits.device, _ptr(logits), B, K, _ptr(t_dev), t_host, _ptr(p), _ptr(ent), _ptr(am))
    return p, ent, am


# ------------------------------------------------------------------------------ feature-map maskers (ViT-CX)
def up_rownorm(src, H, W):
    """src (R,h,w) -> (R, H*W): bilinear up-sample (align_corners=False) and per-row min-max normalisation."""
    _need(src, F32, "src")
    R, h, w = src.shape
    out = torch.empty((R, int(H) * int(W)), dtype=F32, device=src.device)
    _call("xai_up_rownorm_f32", src.device, _ptr(src), R, h, w, int(H), int(W), _ptr(out))
    return out


def rownorm(x, out=None):
    """x (R,P) -> (x - rowmin) / (rowmax - rowmin); out may alias x."""
    _need(x, F32, "x")
    R, P = x.shape
    if out is None:
        out = torch.empty_like(x)
    else:
        _need(out, F32, "out")
        if out.shape != x.shape:
            raise ValueError("out must have the shape of x")
    _call("xai_rownorm_f32", x.device, _ptr(x), R, P, _ptr(out))
    return out


def cluster_sum(rows, members, offs):
    """rows (R,P); members (R,) int32 row ids grouped by cluster; offs (K+1,) int32 -> (K,P) ordered sums."""
    _need(rows, F32, "rows"); _need(members, I32, "members"); _need(offs, I32, "offs")
    R, P = rows.shape
    K_ = offs.numel() - 1
    if K_ < 1 or members.numel() > R * max(K_, 1) or members.dim() != 1 or offs.dim() != 1:
        raise ValueError("members must be 1-D and offs (K+1,) with K >= 1")
    out = torch.empty((K_, P), dtype=F32, device=rows.device)
    _call("xai_cluster_sum_f32", rows.device, _ptr(rows), _ptr(members), _ptr(offs), K_, P, _ptr(out))
    return out


def masked_sums(rows, weights):
    """rows (N,P), weights (N,) -> (sum_n w_n rows_n / N, sum_n rows_n / N), each (P,): one read of the stack."""
    _need(rows, F32, "rows"); _need(weights, F32, "weights")
    N, P = rows.shape
    if weights.numel() != N:
        raise ValueError("one weight per row")
    weighted = torch.empty(P, dtype=F32, device=rows.device)
    plain = torch.empty(P, dtype=F32, device=rows.device)
    _call("xai_masked_sums_f32", rows.device, _ptr(rows), _ptr(weights), N, P, _ptr(weighted), _ptr(plain))
    return weighted, plain


def causal_apply(x, masks, noise, noise_scale=0.1):
    """x (C,H,W); masks (N,H*W); noise (N,C,H,W) standard normal -> (2N,C,H,W): masked+noise rows then image+noise rows."""
    _need(x, F32, "x"); _need(masks, F32, "masks"); _need(noise, F32, "noise")
    Cc, H, W = x.shape
    N = masks.shape[0]
    if masks.shape != (N, H * W) or noise.shape != (N, Cc, H, W):
        raise ValueError(f"masks must be ({N},{H * W}) and noise ({N},{Cc},{H},{W})")
    stack = torch.empty((2 * N, Cc, H, W), dtype=F32, device=x.device)
    _call("xai_causal_apply_f32", x.device, _ptr(x), _ptr(masks), _ptr(noise), N, Cc, H * W, float(noise_scale), _ptr(stack))
    return stack


# ------------------------------------------------------------------------------ opt-in classifier-side fusion
def bn_act_fwd(x, identity, weight, bias, mean, var, eps, variant, relu=True, bn2=None):
    """y = act(bn(x) [+ identity]) for eval-mode BatchNorm2d statistics; x (N,C,H,W) contiguous.
    bn2 = (weight, bias, mean, var, eps): the identity operand gets its own BatchNorm first (down-sample branch)."""
    _need(x, F32, "x")
    for name, t in (("weight", weight), ("bias", bias), ("mean", mean), ("var", var)):
        _need(t, F32, name)
    if identity is not None:
        _need(identity, F32, "identity")
        if identity.shape != x.shape:
            raise ValueError("identity must have the shape of x")
    w2 = b2 = m2 = v2 = None
    eps2 = 0.0
    if bn2 is not None:
        w2, b2, m2, v2, eps2 = bn2
        for name, t in (("weight2", w2), ("bias2", b2), ("mean2", m2), ("var2", v2)):
            _need(t, F32, name)
    N, Cc = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    y = torch.empty_like(x)
    _call("xai_bn_act_fwd_f32", x.device, _ptr(x), _ptr(identity), _ptr(weight), _ptr(bias), _ptr(mean), _ptr(var), float(eps),
          _ptr(w2), _ptr(b2), _ptr(m2), _ptr(v2), float(eps2), int(variant), int(bool(relu)), N, Cc, HW, _ptr(y))
    return y


def bn_relu_bwd(gy, y, weight, var, eps, variant, want_identity=False, gy2=None, bn2=None):
    """-> (gx, g_identity or None) for y = relu(bn(x) [+ identity]); the incoming gradient is gy (+ gy2).
    bn2 = (weight2, var2, eps2): g_identity is the gradient of the identity operand BEFORE its own BatchNorm."""
    _need(gy, F32, "gy"); _need(y, F32, "y"); _need(weight, F32, "weight"); _need(var, F32, "var")
    if gy2 is not None:
        _need(gy2, F32, "gy2")
        if gy2.shape != gy.shape:
            raise ValueError("gy2 must have the shape of gy")
    w2 = v2 = None
    eps2 = 0.0
    if bn2 is not None:
        w2, v2, eps2 = bn2
        _need(w2, F32, "weight2"); _need(v2, F32, "var2")
        want_identity = True
    N, Cc = y.shape[0], y.shape[1]
    HW = y[0, 0].numel()
    gx = torch.empty_like(y)
    gid = torch.empty_like(y) if want_identity else None
    _call("xai_bn_relu_bwd_f32", y.device, _ptr(gy), _ptr(gy2), _ptr(y), _ptr(weight), _ptr(var), float(eps), _ptr(w2), _ptr(v2), float(eps2),
          int(variant), N, Cc, HW, _ptr(gx), _ptr(gid))
    return gx, gid


def maxpool_bwd(gy, indices, H, W, kernel, stride, pad):
    """gy, indices (N,C,PH,PW) (indices int64 from F.max_pool2d(..., return_indices=True)) -> gx (N,C,H,W)."""
    _need(gy, F32, "gy"); _need(indices, torch.int64, "indices")
    N, Cc, PH, PW = gy.shape
    gx = torch.empty((N, Cc, int(H), int(W)), dtype=F32, device=gy.device)
    _call("xai_maxpool_bwd_f32", gy.device, _ptr(gy), _ptr(indices), N * Cc, int(H), int(W), PH, PW, int(kernel), int(stride), int(pad), _ptr(gx))
    return gx


def bn_relu_maxpool_fwd(x, weight, bias, mean, var, eps, variant, kernel, stride, pad):
    """max_pool2d(relu(bn(x)), kernel, stride, pad) for inference; x (N,C,H,W) -> (N,C,PH,PW)."""
    _need(x, F32, "x")
    for name, t in (("weight", weight), ("bias", bias), ("mean", mean), ("var", var)):
        _need(t, F32, name)
    N, Cc, H, W = x.shape
    PH = (H + 2 * pad - kernel) // stride + 1
    PW = (W + 2 * pad - kernel) // stride + 1
    y = torch.empty((N, Cc, PH, PW), dtype=F32, device=x.device)
    _call("xai_bn_relu_maxpool_fwd_f32", x.device, _ptr(x), _ptr(weight), _ptr(bias), _ptr(mean), _ptr(var), float(eps), int(variant),
          N, Cc, H, W, PH, PW, int(kernel), int(stride), int(pad), _ptr(y))
    return y
