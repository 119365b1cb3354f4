"""Randomised shapes (seeded) for the element-wise / index kernels against the CPU oracle: the fixed-shape tests pin the
BASELINE sizes, these walk odd widths, non-multiples of 4, single rows, batch tails.  Integer / byte / element-wise
results exact, fp32 reductions <= 1e-5."""
import numpy as np
import pytest
import torch

from conftest import rel_inf

import os

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SCALE = int(os.environ.get("XAI_FUZZ_SCALE", "1"))       # XAI_FUZZ_SCALE=20: a soak run of the same generators (minutes)


@pytest.fixture(scope="module")
def K():
    from xai_engine import kernels, load_library
    load_library()
    return kernels


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    return (t.to(dtype) if dtype is not None else t).to(DEV).contiguous()


def shapes(seed, n):
    rng = np.random.default_rng(seed)
    for _ in range(n * SCALE):
        yield int(rng.integers(1, 5)), int(rng.integers(1, 70)), int(rng.integers(1, 90)), rng


def test_interp_and_accumulate_random_shapes(K):
    from oracle import ig as oig
    for C, H, W, rng in shapes(1, 12):
        S = int(rng.integers(1, 23))
        x = rng.standard_normal((1, C, H, W)).astype(np.float32)
        b = rng.standard_normal((1, C, H, W)).astype(np.float32)
        al = np.sort(rng.random(S)).astype(np.float32)
        got = K.ig_interp(dev(x), dev(b), dev(al))[0].cpu().numpy()
        assert np.array_equal(got, np.asarray(oig.interpolate(x, b, al)).reshape(got.shape)), (C, H, W, S)   # separate multiply and add: bit-exact
        g = rng.standard_normal((1, S, C, H, W)).astype(np.float32)
        n_use = int(rng.integers(1, S + 1))
        nu = torch.tensor([n_use], dtype=torch.int32, device=DEV)
        out, out_abs = K.ig_accum(dev(g), dev(x), dev(b), n_use=nu, want_abs=True)
        want = oig.accumulate(g[0], n_use, x[0], b[0])
        assert rel_inf(out[0].cpu().numpy(), want) <= 1e-5, (C, H, W, S, n_use)
        assert rel_inf(out_abs[0].cpu().numpy(), np.abs(want.sum(0))) <= 1e-5


def test_rank_flip_perturb_segment_random_shapes(K):
    from oracle import perturb as op
    for C, H, W, rng in shapes(2, 12):
        hw = H * W
        sal = rng.standard_normal(hw).astype(np.float32)
        sal[rng.integers(0, hw, max(1, hw // 5))] = 0.0                                      # ties
        order, rank = K.rank(dev(sal[None]))
        want_order = np.argsort(sal, kind="stable")
        assert np.array_equal(order[0].cpu().numpy(), want_order)
        assert np.array_equal(rank[0].cpu().numpy()[want_order], np.arange(hw))
        step = int(rng.integers(1, hw + 1))
        n_steps = -(-hw // step)
        for desc in (True, False):
            flip = K.flip_steps(rank[0], desc, step)
            pos = (hw - 1 - rank[0].cpu().numpy()) if desc else rank[0].cpu().numpy()
            assert np.array_equal(flip.cpu().numpy(), pos // step)
            start = rng.standard_normal((C, H, W)).astype(np.float32)
            finish = rng.standard_normal((C, H, W)).astype(np.float32)
            first = int(rng.integers(0, n_steps))
            nb = int(rng.integers(1, n_steps - first + 1))
            imgs = K.perturb_batch(dev(start), dev(finish), flip, first, nb).cpu().numpy()
            f = (pos // step).reshape(1, H, W)
            for k in range(nb):
                assert np.array_equal(imgs[k], np.where(f <= first + k, finish, start)), (C, H, W, step, desc, k)
            seg, total = K.segment_sums(dev(sal), order[0], desc, step, n_steps)
            seq = want_order[::-1] if desc else want_order
            want_seg = np.array([sal[seq[t * step:(t + 1) * step]].astype(np.float64).sum() for t in range(n_steps)])
            assert np.abs(seg.cpu().numpy() - want_seg).max() <= 1e-5 * max(np.abs(want_seg).max(), 1e-6) + 1e-6
            assert abs(float(total) - sal.astype(np.float64).sum()) <= 1e-4 * max(abs(sal.astype(np.float64)).sum(), 1.0)


def test_softmax_stats_and_blur_random_shapes(K):
    from oracle import perturb as op
    rng = np.random.default_rng(3)
    for _ in range(10 * SCALE):
        B, Kc = int(rng.integers(1, 60)), int(rng.integers(2, 1200))
        z = (rng.standard_normal((B, Kc)) * rng.uniform(0.5, 6)).astype(np.float32)
        t = int(rng.integers(0, Kc))
        p, ent, am = K.softmax_stats(dev(z), t)
        sm = op.softmax_rows(z)
        assert rel_inf(p.cpu().numpy(), sm[:, t]) <= 1e-5
        assert rel_inf(ent.cpu().numpy(), op.entropy_bits(sm)) <= 1e-4
        assert np.array_equal(am.cpu().numpy(), z.argmax(1))
    for C, H, W, rng in shapes(4, 8):
        klen = int(rng.choice([3, 5, 11, 31, 63]))
        k1 = op.gkern1d(klen, max(1.0, klen / 2)).astype(np.float32)
        x = rng.standard_normal((2, C, H, W)).astype(np.float32)
        got = K.blur_sep(dev(x), dev(k1)).cpu().numpy()
        kern = torch.zeros(C, C, klen, klen)
        for c in range(C):
            kern[c, c] = torch.outer(torch.from_numpy(k1), torch.from_numpy(k1))
        want = torch.nn.functional.conv2d(torch.from_numpy(x).double(), kern.double(), padding=klen // 2).numpy()
        assert np.abs(got - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), (C, H, W, klen)


def test_gradcam_bilinear_and_maskers_random_shapes(K):
    from oracle import gradcam as ogc
    from oracle import vit_cx as ocx
    rng = np.random.default_rng(5)
    for _ in range(8 * SCALE):
        B, Cc, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 300)), int(rng.integers(1, 15)), int(rng.integers(1, 15))
        act = rng.standard_normal((B, Cc, h, w)).astype(np.float32)
        grad = rng.standard_normal((B, Cc, h, w)).astype(np.float32)
        cam = K.gradcam(dev(act), dev(grad), relu=True)
        want = ogc.cam_reduce(act, grad, relu=True)
        assert np.abs(cam.cpu().numpy() - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3)
        Ho, Wo = int(rng.integers(h, 8 * h + 2)), int(rng.integers(w, 8 * w + 2))
        up = K.bilinear_up(cam, Ho, Wo).cpu().numpy()
        assert np.abs(up - ogc.bilinear_up(cam.cpu().numpy(), Ho, Wo)).max() <= 2e-6 * max(np.abs(want).max(), 1e-3)
        if h * w >= 2:
            fm = rng.standard_normal((Cc, h, w)).astype(np.float32)
            rows = K.up_rownorm(dev(fm), Ho, Wo).cpu().numpy()
            up_ref = ocx.resize_maps(fm, Ho, Wo).reshape(Cc, Ho * Wo)
            ref = ocx.norm_matrix(up_ref)
            ok = np.isfinite(ref).all(axis=1)                                                  # constant maps are 0/0 in both
            # (v - min) / (max - min): the last-bit differences of the two up-samples are amplified by |v|max / (max - min),
            # so a nearly flat map is held to 4e-6 times that factor (a 25x soak run found 5.5e-6 on one of 58 000 maps)
            span = up_ref.max(axis=1) - up_ref.min(axis=1)
            amp = np.maximum(1.0, np.abs(up_ref).max(axis=1) / np.where(span > 0, span, 1.0))
            err = np.abs(rows - ref).max(axis=1)
            assert (err[ok] <= 4e-6 * amp[ok]).all(), (Cc, h, w, Ho, Wo, float(err[ok].max(initial=0.0)))


def test_rise_random_geometries(K):
    """RISE masks and accumulation for random image sizes, grid sizes s (the bit-packed s = 8 path and the generic one),
    keep probabilities and mask counts, against the oracle's scipy up-sampling."""
    from oracle import rise as orise
    rng0 = np.random.default_rng(8)
    for _ in range(10 * SCALE):
        H, W = int(rng0.integers(9, 120)), int(rng0.integers(9, 120))
        s = int(rng0.choice([2, 3, 5, 7, 8, 8, 11]))
        N = int(rng0.integers(1, 40))
        p1 = float(rng0.uniform(0.2, 0.8))
        rng = np.random.RandomState(int(rng0.integers(0, 1 << 30)))
        grid, shifts, cell = orise.draw_grid_and_shifts((H, W), N, s, p1, rng)
        image = rng0.standard_normal((3, H, W)).astype(np.float32)
        g8, sh = dev(grid.astype(np.uint8)), dev(shifts)
        masked, masks = K.rise_apply(g8, sh, cell, dev(image), want_masked=True, want_masks=True)
        want_masks = orise.masks_from(grid, shifts, (H, W), cell)[:, 0]
        assert np.abs(masks.cpu().numpy() - want_masks).max() <= 1e-6, (H, W, s, N)
        assert masks.min() >= 0 and masks.max() <= 1
        np.testing.assert_array_equal(masked.cpu().numpy(), image[None] * masks.cpu().numpy()[:, None])
        scores = rng0.random(N).astype(np.float32)
        acc = K.rise_accum(g8, sh, dev(scores), cell, H, W, 1.0 / N / p1)
        want = (scores.reshape(-1, 1).astype(np.float64) * want_masks.reshape(N, -1)).sum(0).reshape(H, W) / N / p1
        assert np.abs(acc.cpu().numpy() - want).max() <= 2e-6 * max(np.abs(want).max(), 1e-3), (H, W, s, N)


def test_single_run_random_plans():
    """The five metric classes on random image sizes, step sizes (ragged last steps), batch sizes (remainder batches, batch >
    steps), modes, tie-free and heavily tied maps, pixel and patch_mask branches, against the oracle driving the same
    device-resident tiny classifier: every return value at the 1e-5 bar, counts exact."""
    import importlib
    from conftest import load_golden
    from helpers import tiny_from, logits_fn_of
    from oracle import perturb as op
    g = load_golden("ig_small.npz")
    model = tiny_from(g, DEV)
    fn = logits_fn_of(model)
    table = [("MASTestFunctions", "MASMetric", ("ins", "del", "lerf", "morf"), op.mas), ("RISETestFunctions", "RISEMetric", ("ins", "del", "lerf", "morf"), op.rise_metric),
             ("AICTestFunctions", "AICMetric", ("ins", "del"), op.aic), ("PosNegPertFunctions", "PositiveNegativePerturbation", ("lerf", "morf"), op.pnp),
             ("MonotonicityTest", "MonotonicityMetric", ("positive", "negative"), op.mono)]
    rng = np.random.default_rng(11)
    for case in range(8 * SCALE):
        H, W = int(rng.integers(4, 41)) * 1, int(rng.integers(4, 41))
        HW = H * W
        x = rng.standard_normal((1, 3, H, W)).astype(np.float32)
        # non-negative maps, as every caller of the metrics passes (|sum over channels|, evaluatePerturbation.py:181): with signed
        # maps MAS's density = segment sum / total sum divides by a cancelling fp32 sum (a 1500x soak run: total ~ 1e-3 of the
        # segments, 1.9e-5 between two correct summation orders)
        sal = np.abs(rng.standard_normal((H, W))).astype(np.float32)
        if case % 3 == 2:
            sal = np.round(sal * 2) / 2                                  # many ties: the stable-order rule decides
        modname, clsname, modes, ofunc = table[int(rng.integers(0, len(table)))]
        mode = modes[int(rng.integers(0, len(modes)))]
        cls = getattr(importlib.import_module("util.test_methods." + modname), clsname)
        pm = None
        if case % 4 == 3 and H % 2 == 0 and W % 2 == 0:                  # patch branch: 2 x 2 blocks of patches
            ph, pw = H // 2, W // 2
            pm = np.arange(4).reshape(2, 2).repeat(ph, 0).repeat(pw, 1)
        step = int(rng.integers(1, HW + 1)) if pm is None else HW
        max_bs = int(rng.integers(1, 70))
        zeros = lambda t: torch.zeros_like(t)                           # noqa: E731
        res = cls(model, HW, mode, step, zeros).single_run(torch.from_numpy(x), sal.copy(), DEV, patch_mask=None if pm is None else torch.from_numpy(pm),
                                                             max_batch_size=max_bs)
        want = ofunc(fn, x, sal, mode, step, np.zeros_like, pm, max_bs)
        assert len(res) == len(want), (clsname, mode)
        for i, (r, w) in enumerate(zip(res, want)):
            tag = (clsname, mode, H, W, step, max_bs, pm is not None, i)
            if np.ndim(w) == 0 and isinstance(w, (int, np.integer)):
                assert int(r) == int(w), tag
            else:
                r64, w64 = np.asarray(r, np.float64), np.asarray(w, np.float64)
                assert r64.shape == w64.shape, tag
                both_nan = np.isnan(r64) & np.isnan(w64)
                assert np.abs(np.where(both_nan, 0, r64 - w64)).max(initial=0.0) <= 1e-5 * max(1.0, np.abs(np.where(np.isnan(w64), 0, w64)).max(initial=0.0)), tag


def test_ig_family_random_cases():
    """IG / Left-IG / IDG / IDGI on random image sizes, step counts, batch sizes (every divisor plan), alpha_star, scalar and
    tensor baselines, against the oracle driving the same device-resident tiny classifier: 1e-5, and the print-and-return-zeros
    convention when steps % batch_size != 0."""
    from conftest import load_golden
    from helpers import tiny_from
    from oracle import ig as oig
    from util.attribution_methods import saliencyMethods as attr
    g = load_golden("ig_small.npz")
    model = tiny_from(g, DEV)
    rng = np.random.default_rng(21)
    for case in range(6 * SCALE):
        H, W = 4 * int(rng.integers(1, 11)), 4 * int(rng.integers(1, 11))          # the tiny net pools to 4 x 4
        steps = int(rng.choice([4, 6, 10, 12, 20, 30, 50]))
        divisors = [d for d in range(1, steps + 1) if steps % d == 0]
        bs = int(rng.choice(divisors))
        x = rng.standard_normal((1, 3, H, W)).astype(np.float32)
        base = float(rng.choice([0.0, 0.25, -0.5])) if case % 2 == 0 else (rng.standard_normal((1, 3, H, W)) * 0.3).astype(np.float32)
        base_t = torch.from_numpy(base) if isinstance(base, np.ndarray) else base
        with torch.no_grad():
            t = int(model(torch.from_numpy(x).to(DEV)).argmax(1)[0])
        which = ("ig", "lig", "idg", "idgi")[case % 4]
        tag = (which, H, W, steps, bs, isinstance(base, np.ndarray))
        if which in ("ig", "lig"):
            a_star = 1 if which == "ig" else float(rng.choice([0.5, 0.9, 0.99]))
            got = attr.IG(torch.from_numpy(x), model, steps, bs, a_star, base_t, DEV, torch.tensor(t)).cpu().numpy()
            want = oig.ig(x, model, steps, bs, a_star, base, t)
        elif which == "idg":
            if steps < 6:
                continue                                                             # the slope schedule needs a few steps to be defined
            got = attr.IDG(torch.from_numpy(x), model, steps, bs, base_t, DEV, torch.tensor(t)).cpu().numpy()
            want = oig.idg(x, model, steps, bs, base, t)
        else:
            got = attr.IDGI(torch.from_numpy(x), model, steps, bs, base_t, DEV, torch.tensor(t)).cpu().numpy()
            want = oig.idgi(x, model, steps, bs, base, t)
        if not np.isfinite(want).all():
            assert np.array_equal(np.isfinite(got), np.isfinite(want)), tag           # degenerate schedules are NaN in both
            continue
        assert rel_inf(got, want) <= 1e-5, (tag, rel_inf(got, want))
    bad = attr.IG(torch.from_numpy(x), model, 10, 3, 1, 0, DEV, torch.tensor(t))
    assert bad == (0, 0, 0, 0)


def test_stream_workers_random_batches_bit_identical_to_one_stream(K):
    """Seeded random batches through the multi-stream drivers: ig_batch (random image count, images per pass -- ragged last pass --,
    step count, scalar / tensor baseline, IG / Left-IG, buffered / streaming, graphs on / off, 2-4 streams) and rise (random mask
    count and batch) must equal their one-stream results bit for bit; K16 against sequential fp32 sums.  (Deterministic solvers,
    conftest.)"""
    from xai_engine.ig import ig_batch
    from xai_engine.rise import rise, draw_masks
    from conftest import load_golden
    from helpers import tiny_from
    g = load_golden("ig_small.npz")
    model = tiny_from(g, DEV)
    rng = np.random.default_rng(11)
    for case in range(6 * SCALE):
        B = int(rng.integers(1, 9))
        ipp = int(rng.integers(1, 4))
        steps = int(rng.choice([10, 20, 50]))
        xs = torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(1000 + case)).to(DEV)
        with torch.no_grad():
            ts = model(xs).argmax(1)
        base = (torch.randn(B, 3, 32, 32, generator=torch.Generator().manual_seed(2000 + case)) * 0.3).to(DEV) if rng.random() < 0.4 else float(rng.random())
        kw = dict(steps=steps, images_per_pass=ipp, baseline=base, want_abs=True)
        if rng.random() < 0.4:
            kw["alpha_star"] = 0.9
        elif rng.random() < 0.5:
            kw["buffered"] = True
        one = ig_batch(xs, model, ts, **kw)
        many = ig_batch(xs, model, ts, streams=int(rng.integers(2, 5)), graphs=bool(rng.random() < 0.7), **kw)
        for a, b in zip(one, many):
            assert torch.equal(a, b), (case, B, ipp, steps, kw.keys())
        n = int(rng.integers(1, 90))
        bs = int(rng.integers(1, 40))
        np.random.seed(case)
        masks = draw_masks((32, 32), n, 4, 0.5)
        score = lambda b_: torch.softmax(model(b_), 1)[:, 3]                                # noqa: E731
        r1 = rise(model, xs[:1].cpu(), None, DEV, N=n, s=4, p1=0.5, score_fn=score, batch_size=bs, masks=masks)
        r3 = rise(model, xs[:1].cpu(), None, DEV, N=n, s=4, p1=0.5, score_fn=score, batch_size=bs, masks=masks, streams=3)
        assert torch.equal(r1, r3), (case, n, bs)
        N, P = int(rng.integers(1, 300)), int(rng.integers(1, 700))
        rows, w = rng.random((N, P)).astype(np.float32), rng.standard_normal(N).astype(np.float32)
        ws_, ps_ = K.masked_sums(dev(rows), dev(w))
        aw, ap = np.zeros(P, np.float32), np.zeros(P, np.float32)
        for i in range(N):
            aw += rows[i] * w[i]
            ap += rows[i]
        assert np.array_equal(ws_.cpu().numpy(), aw / np.float32(N)) and np.array_equal(ps_.cpu().numpy(), ap / np.float32(N)), (case, N, P)
