"""Installation self-check: `python -m xai_engine.selfcheck [--device cuda:0]`.

Runs every kernel of libxai_hip.so once on random data and compares it with the equivalent torch
expression evaluated ON THE SAME DEVICE (this is a smoke test for a deployment, not the parity suite --
that lives in tests/ and uses the CPU oracle).  Exit code 0 when everything agrees.
"""
import argparse
import sys

import torch
import torch.nn.functional as F

from . import kernels as K
from . import load_library


def _rel(a, b):
    a, b = a.double(), b.double()
    den = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / den)


def run(device="cuda:0", verbose=True):
    load_library()
    dev = torch.device(device)
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=gen)      # noqa: E731
    results = []

    def check(name, err, tol):
        ok = err <= tol
        results.append((name, err, tol, ok))
        if verbose:
            print(f"{'PASS' if ok else 'FAIL'}  {name:28s} err {err:.2e}  (tol {tol:.0e})")

    x, b = rnd(2, 3, 64, 64), rnd(2, 3, 64, 64) * 0.3
    al = torch.linspace(0, 1, 20).to(dev)
    check("ig_interp", _rel(K.ig_interp(x, b, al), b[:, None] + al.view(1, -1, 1, 1, 1) * (x - b)[:, None]), 0.0)
    g = rnd(2, 20, 3, 64, 64)
    out, out_abs = K.ig_accum(g, x, b, want_abs=True)
    ref = g.double().mean(1) * (x - b).double()
    check("ig_accum", _rel(out, ref), 2e-6)
    check("ig_accum |sum_c|", _rel(out_abs, ref.sum(1).abs()), 1e-5)
    lg = rnd(2, 20)
    nu = K.ig_cutoff(lg, 0.9)
    want = torch.stack([(row > row.max() * 0.9).nonzero()[0, 0].clamp_min(1) for row in lg]).int()
    check("ig_cutoff", float((nu - want).abs().max()), 0.0)
    acc = torch.zeros(1, 3, 64, 64, device=dev)
    K.ig_accum_add(g[0], acc[0])
    check("ig_accum_add/finish", _rel(K.ig_finish(acc, 20, x[:1], 0.0), g[0].double().mean(0) * x[0].double()), 2e-6)
    sq = K.sumsq(g[0])
    check("sumsq", _rel(sq, (g[0].double() ** 2).flatten(1).sum(1)), 2e-6)
    act, grad = rnd(2, 96, 7, 7), rnd(2, 96, 7, 7)
    cam = K.gradcam(act, grad, relu=True)
    ref = torch.relu((grad.double().mean((2, 3), keepdim=True) * act.double()).sum(1))
    check("gradcam", float((cam.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)), 1e-5)
    check("bilinear_up", _rel(K.bilinear_up(cam, 56, 56), F.interpolate(cam[None], size=(56, 56), mode="bilinear", align_corners=False)[0]), 2e-6)
    sal = rnd(3, 4096).relu()                                        # many ties
    order, rank = K.rank(sal)
    check("rank (stable argsort)", float((order.long() - torch.sort(sal, dim=1, stable=True)[1]).abs().max()), 0.0)
    flip = K.flip_steps(rank[0], True, 64)
    start, finish = rnd(3, 64, 64), rnd(3, 64, 64)
    imgs = K.perturb_batch(start, finish, flip, 0, 64)
    pos = 4095 - rank[0].long()
    ref = torch.where((pos // 64).view(1, 1, 64, 64) <= torch.arange(64, device=dev).view(-1, 1, 1, 1), finish[None], start[None])
    check("flip_steps + perturb_batch", _rel(imgs, ref), 0.0)
    seg, total = K.segment_sums(sal[0], order[0], True, 64, 64)
    check("segment_sums", _rel(seg, sal[0][order[0].long().flip(0)].view(64, 64).double().sum(1)), 2e-6)
    z = rnd(9, 1000) * 3
    p, ent, am = K.softmax_stats(z, 5)
    sm = torch.softmax(z.double(), 1)
    check("softmax_stats p", _rel(p, sm[:, 5]), 2e-6)
    check("softmax_stats entropy", _rel(ent, -(sm * sm.log2()).sum(1)), 1e-5)
    check("softmax_stats argmax", float((am.long() - z.argmax(1)).abs().max()), 0.0)
    k1 = torch.softmax(rnd(31), 0)
    kern = torch.zeros(3, 3, 31, 31, device=dev)
    for c in range(3):
        kern[c, c] = torch.outer(k1, k1)
    xb = rnd(2, 3, 40, 70)
    check("blur_sep", _rel(K.blur_sep(xb, k1), F.conv2d(xb.double(), kern.double(), padding=15)), 1e-5)
    grid = (torch.rand(6, 8, 8, device=dev, generator=gen) < 0.5).to(torch.uint8)
    sh = torch.randint(0, 8, (6, 2), device=dev, generator=gen, dtype=torch.int32)
    img = rnd(3, 64, 64)
    masked, masks = K.rise_apply(grid, sh, (8, 8), img, want_masked=True, want_masks=True)
    check("rise_apply (masked = image*mask)", _rel(masked, img[None] * masks[:, None]), 0.0)
    check("rise masks in [0,1]", float(max(-masks.min(), masks.max() - 1).clamp_min(0)), 0.0)
    sc = torch.rand(6, device=dev, generator=gen)
    check("rise_accum", _rel(K.rise_accum(grid, sh, sc, (8, 8), 64, 64, 0.5), 0.5 * (sc.double().view(-1, 1, 1) * masks.double()).sum(0)), 1e-9)
    fm = rnd(10, 5, 5)
    up = F.interpolate(fm[None].double(), size=(32, 32), mode="bilinear", align_corners=False)[0].flatten(1)
    lo, hi = up.min(1, keepdim=True)[0], up.max(1, keepdim=True)[0]
    rows = K.up_rownorm(fm, 32, 32)
    check("up_rownorm", float((rows.double() - (up - lo) / (hi - lo)).abs().max()), 2e-6)
    flat = rnd(7, 1000)
    lo, hi = flat.min(1, keepdim=True)[0], flat.max(1, keepdim=True)[0]
    check("rownorm", _rel(K.rownorm(flat), (flat - lo) / (hi - lo)), 0.0)
    labels = torch.tensor([2, 0, 1, 0, 2, 2, 1, 0, 1, 2])
    members = torch.argsort(labels, stable=True).int().to(dev)
    offs = torch.tensor([0, 3, 6, 10], dtype=torch.int32, device=dev)
    ref = torch.zeros(3, 1024, device=dev, dtype=torch.float64).index_add_(0, labels.to(dev), rows.double())
    check("cluster_sum", _rel(K.cluster_sum(rows, members, offs), ref), 2e-6)
    m, noise = torch.rand(4, 64 * 64, device=dev, generator=gen), rnd(4, 3, 64, 64)
    stack = K.causal_apply(img, m, noise, 0.1)
    add = (noise * 0.1) * (1 - m.view(4, 1, 64, 64))
    check("causal_apply", _rel(stack, torch.cat([img[None] * m.view(4, 1, 64, 64) + add, img[None] + add])), 0.0)
    # opt-in classifier fusion: must reproduce THIS PyTorch build's eval-mode BN / ReLU / max-pool kernels bit for bit
    from .prepare import BN_VARIANT
    xb_, idt_, gy_ = rnd(3, 8, 14, 14), rnd(3, 8, 14, 14), rnd(3, 8, 14, 14)
    wv, bv, mv, vv = rnd(8).abs() + 0.5, rnd(8), rnd(8), rnd(8).abs() + 0.2
    xr, ir = xb_.clone().requires_grad_(True), idt_.clone().requires_grad_(True)
    yv = F.relu(F.batch_norm(xr, mv, vv, wv, bv, False, 0.0, 1e-5) + ir)
    yv.backward(gy_)
    check("bn_act_fwd (bitwise vs PyTorch)", float((K.bn_act_fwd(xb_, idt_, wv, bv, mv, vv, 1e-5, BN_VARIANT) != yv.detach()).sum()), 0.0)
    gx, gid = K.bn_relu_bwd(gy_, yv.detach(), wv, vv, 1e-5, BN_VARIANT, want_identity=True)
    check("bn_relu_bwd (bitwise vs PyTorch)", float((gx != xr.grad).sum() + (gid != ir.grad).sum()), 0.0)
    xp = rnd(2, 4, 15, 17).requires_grad_(True)
    yp, ip = F.max_pool2d(xp, 3, 2, 1, 1, False, True)
    gp = rnd(*yp.shape)
    yp.backward(gp)
    check("maxpool_bwd (bitwise vs PyTorch)", float((K.maxpool_bwd(gp, ip, 15, 17, 3, 2, 1) != xp.grad).sum()), 0.0)
    want = F.max_pool2d(F.relu(F.batch_norm(xb_, mv, vv, wv, bv, False, 0.0, 1e-5)), 3, 2, 1)
    check("bn_relu_maxpool_fwd (bitwise vs PyTorch)", float((K.bn_relu_maxpool_fwd(xb_, wv, bv, mv, vv, 1e-5, BN_VARIANT, 3, 2, 1) != want).sum()), 0.0)
    # K16 and the stream workers
    rows, wts = rnd(37, 200).abs(), rnd(37)
    wsum, psum = K.masked_sums(rows, wts)
    check("masked_sums", max(_rel(wsum, (rows.double() * wts.double()[:, None]).sum(0) / 37), _rel(psum, rows.double().sum(0) / 37)), 2e-6)
    wrong = unprotected = 0.0
    for _ in range(3):          # which solver MIOpen serves the probe's shape with settles after its first uses in a process: look more than once
        w, u = streams_probe(dev)
        wrong, unprotected = max(wrong, w), max(unprotected, u)
    check("stream workers (wrong results)", float(wrong), 0.0)
    if verbose:
        print(f"info  one host thread on two streams, unprotected: {unprotected:.0%} of the probe's launches wrong on this stack "
              "(why xai_engine.streams uses one host thread per stream)")
    return all(r[3] for r in results), results


def streams_probe(dev, trials=6, reps=6):
    """The launch that exposes the one-thread-two-streams hazard of PyTorch-ROCm (xai_engine/streams.py): the backward-data of a 1x1
    convolution 512 -> 2048 on 7x7 at batch 50 (ResNet-50's layer4.0.conv3).  -> (fraction of wrong results when two stream workers
    issue it concurrently -- must be 0 --, fraction when one host thread issues it alternately on two streams -- informational)."""
    from .streams import workers
    keep = (torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic)
    # the configuration in which MIOpen serves this launch with its rocBLAS split-K GEMM solver (the parity configuration of the tests)
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
    conv = torch.nn.Conv2d(512, 2048, 1, bias=False).to(dev)
    torch.nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
    conv.weight.requires_grad_(False)
    gen = torch.Generator(device=dev).manual_seed(5)

    def grad(x, gy):
        xr = x.detach().requires_grad_(True)
        (gx,) = torch.autograd.grad(conv(xr), xr, gy)
        return gx
    ws = workers(dev, 2)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    bad_workers = bad_one_thread = 0
    for _ in range(trials):
        xs = [torch.randn(50, 512, 7, 7, device=dev, generator=gen) for _ in range(2)]
        gys = [torch.randn(50, 2048, 7, 7, device=dev, generator=gen) for _ in range(2)]
        want = [grad(xs[k], gys[k]) for k in range(2)]
        torch.cuda.synchronize(dev)
        got = [f.result() for f in [ws[k].submit(lambda k=k: [grad(xs[k], gys[k]) for _ in range(reps)]) for k in range(2)]]
        torch.cuda.synchronize(dev)
        bad_workers += sum(not torch.equal(a, want[k]) for k in range(2) for a in got[k])
        got = [[], []]
        for _ in range(reps):
            for k in range(2):
                with torch.cuda.stream(streams[k]):
                    got[k].append(torch.ops.aten.convolution_backward(gys[k], xs[k], conv.weight, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                                      [True, False, False])[0])
        torch.cuda.synchronize(dev)
        bad_one_thread += sum(not torch.equal(a, want[k]) for k in range(2) for a in got[k])
    total = 2 * reps * trials
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = keep
    return bad_workers / total, bad_one_thread / total


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args(argv)
    ok, _ = run(args.device)
    print("all kernels agree with torch on", args.device if ok else "-- FAILURES above")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
