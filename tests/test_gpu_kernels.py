"""Kernel-level parity on a real MI355X: every C-ABI entry point (through ctypes, via
xai_engine.kernels) against the CPU oracle / the reference-made golden vectors on identical
inputs.  Integer/index/byte outputs are compared exactly; fp32 outputs with
||a-b||inf/||b||inf <= 1e-5 (BASELINE.json's bar), usually far tighter as noted per test.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_inf

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 1e-5


@pytest.fixture(scope="module")
def K():
    from xai_engine import kernels
    from xai_engine import load_library
    load_library()
    return kernels


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


# ------------------------------------------------------------------------------ K1
@pytest.mark.parametrize("shape", [(3, 32, 32), (3, 224, 224), (3, 30, 45), (1, 7, 9)])
def test_ig_interp_bit_exact(K, shape):
    from oracle import ig as oig
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2,) + shape).astype(np.float32)
    b = (rng.standard_normal((2,) + shape) * 0.3).astype(np.float32)
    al = oig.linspace01(50)
    want = np.stack([oig.interpolate(x[i], b[i], al) for i in range(2)])
    got = K.ig_interp(dev(x), dev(b), dev(al)).cpu().numpy()
    np.testing.assert_array_equal(got, want)                     # mul then add, no FMA: bit-exact
    want0 = np.stack([oig.interpolate(x[i], np.full(shape, 0.25, np.float32), al[10:17]) for i in range(2)])
    got0 = K.ig_interp(dev(x), 0.25, dev(al)[10:17]).cpu().numpy()
    np.testing.assert_array_equal(got0, want0)
    # per-image schedules
    al2 = np.stack([al[:5], al[20:25]])
    got2 = K.ig_interp(dev(x), dev(b), dev(al2)).cpu().numpy()
    for i in range(2):
        np.testing.assert_array_equal(got2[i], oig.interpolate(x[i], b[i], al2[i]))


def test_ig_interp_full_size_hbm_branch_bit_exact(K):
    """BASELINE's full size (32 images x 50 steps x 3x224x224 = 963 MB written): the HBM-sized launch plan of K1 (2 step rows
    per lane, non-temporal stores) against base + alpha * (x - base) evaluated by torch on the device with the same two
    roundings -- bit for bit, tensor and scalar baselines."""
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn((32, 3, 224, 224), device=DEV, generator=gen)
    b = torch.randn((32, 3, 224, 224), device=DEV, generator=gen) * 0.3
    al = torch.linspace(0, 1, 50).to(DEV)
    got = K.ig_interp(x, b, al)
    assert got.shape == (32, 50, 3, 224, 224)
    for i in (0, 13, 31):
        want = b[i][None] + al.view(-1, 1, 1, 1) * (x[i] - b[i])[None]
        assert torch.equal(got[i], want)
    got = K.ig_interp(x, 0.25, al)
    for i in (5, 31):
        want = torch.full_like(x[i], 0.25)[None] + al.view(-1, 1, 1, 1) * (x[i] - 0.25)[None]
        assert torch.equal(got[i], want)


# ------------------------------------------------------------------------------ K2
def test_ig_accum_on_reference_gradients(K):
    """The reference's own per-step gradients in, the reference's IG / Left-IG out."""
    g = load_golden("ig_small.npz")
    grads = dev(g["gradients"][None])                                     # (1,50,3,32,32)
    x = dev(g["x"])
    out, out_abs = K.ig_accum(grads, x, 0.0, want_abs=True)
    assert rel_inf(out[0].cpu().numpy(), g["ig"]) <= 2e-6
    assert rel_inf(out_abs[0].cpu().numpy(), np.abs(g["ig"].sum(0))) <= 2e-6
    n_use = K.ig_cutoff(dev(g["logits"][None]), 0.9)
    from oracle import ig as oig
    assert int(n_use[0]) == oig.left_cutoff(g["logits"], 0.9)
    lig = K.ig_accum(grads, x, 0.0, n_use=n_use)
    assert rel_inf(lig[0].cpu().numpy(), g["lig"]) <= 2e-6
    lig2 = K.ig_accum(grads, x, 0.0, n_use=int(n_use[0]))
    np.testing.assert_array_equal(lig2.cpu().numpy(), lig.cpu().numpy())


@pytest.mark.parametrize("shape,n_img,steps", [((3, 224, 224), 3, 50), ((3, 30, 45), 2, 7), ((1, 5, 5), 1, 1), ((3, 64, 64), 2, 19),
                                                ((1, 16, 16), 3, 11), ((4, 8, 8), 2, 5), ((3, 224, 224), 12, 6)])
def test_ig_accum_vs_oracle(K, shape, n_img, steps):
    from oracle import ig as oig
    rng = np.random.default_rng(2)
    grads = rng.standard_normal((n_img, steps) + shape).astype(np.float32)
    x = rng.standard_normal((n_img,) + shape).astype(np.float32)
    b = (rng.standard_normal((n_img,) + shape) * 0.2).astype(np.float32)
    n_use = rng.integers(1, steps + 1, n_img).astype(np.int32)
    out, out_abs = K.ig_accum(dev(grads), dev(x), dev(b), n_use=dev(n_use), want_abs=True)
    for i in range(n_img):
        want = oig.accumulate(grads[i], int(n_use[i]), x[i], b[i])
        assert rel_inf(out[i].cpu().numpy(), want) <= 2e-6
        assert rel_inf(out_abs[i].cpu().numpy(), np.abs(want.sum(0, dtype=np.float32))) <= 2e-6
    # weighted (IDG) form: mean over ALL steps of g*w1*w2
    w1 = rng.standard_normal((n_img, steps)).astype(np.float32)
    w2 = rng.random((n_img, steps)).astype(np.float32)
    got = K.ig_accum(dev(grads), dev(x), 0.5, w1=dev(w1), w2=dev(w2)).cpu().numpy()
    for i in range(n_img):
        wg = (grads[i] * w1[i].reshape(-1, 1, 1, 1)) * w2[i].reshape(-1, 1, 1, 1)
        want = (wg.sum(0, dtype=np.float32) / np.float32(steps)) * (x[i] - np.float32(0.5))
        assert rel_inf(got[i], want) <= 2e-6


def test_ig_streaming_form_equals_buffered(K):
    rng = np.random.default_rng(3)
    grads = rng.standard_normal((1, 50, 3, 64, 64)).astype(np.float32)
    x = rng.standard_normal((1, 3, 64, 64)).astype(np.float32)
    acc = torch.zeros((1, 3, 64, 64), device=DEV)
    g = dev(grads)
    K.ig_accum_add(g[0, :25], acc)
    K.ig_accum_add(g[0, 25:], acc)
    a, a_abs = K.ig_finish(acc, 50, dev(x), 0.0, want_abs=True)
    b, b_abs = K.ig_accum(g, dev(x), 0.0, want_abs=True)
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())       # same summation order
    np.testing.assert_array_equal(a_abs.cpu().numpy(), b_abs.cpu().numpy())


def test_ig_accum_linearity_full_size(K):
    """Size-independent property at BASELINE's full size (32 images x 50 steps x 3x224x224,
    963 MB): accum(a*G1 + G2) == a*accum(G1) + accum(G2) up to rounding, and a constant
    gradient field integrates to exactly (x - b)."""
    n_img, steps, shape = 32, 50, (3, 224, 224)
    gen = torch.Generator(device=DEV).manual_seed(0)
    g1 = torch.randn((n_img, steps) + shape, device=DEV, generator=gen)
    x = torch.randn((n_img,) + shape, device=DEV, generator=gen)
    ones = torch.ones_like(g1)
    out = K.ig_accum(ones, x, 0.25)
    np.testing.assert_array_equal(out.cpu().numpy(), (x - 0.25).cpu().numpy())
    a1 = K.ig_accum(g1, x, 0.0)
    ref = g1.double().mean(1) * x.double()
    assert rel_inf(a1.cpu().numpy(), ref.cpu().numpy()) <= 2e-6
    g2 = g1 * 2.0                                                          # exact scaling in fp32
    a2 = K.ig_accum(g2, x, 0.0)
    np.testing.assert_array_equal(a2.cpu().numpy(), (a1 * 2.0).cpu().numpy())
    del g2, ones
    # per-image Left-IG cutoffs on the balanced streaming path (lanes whose items straddle two images)
    n_use = torch.randint(1, steps + 1, (n_img,), generator=torch.Generator().manual_seed(1), dtype=torch.int32).to(DEV)
    a3, a3_abs = K.ig_accum(g1, x, 0.5, n_use=n_use, want_abs=True)
    for i in (0, 7, 31):
        n = int(n_use[i])
        want = g1[i, :n].double().mean(0) * (x[i].double() - 0.5)
        assert rel_inf(a3[i].cpu().numpy(), want.cpu().numpy()) <= 2e-6
        assert rel_inf(a3_abs[i].cpu().numpy(), want.sum(0).abs().cpu().numpy()) <= 1e-5
    # weighted (IDG) form on the same path
    w1 = torch.randn(n_img, steps, device=DEV, generator=gen)
    w2 = torch.rand(n_img, steps, device=DEV, generator=gen)
    a4 = K.ig_accum(g1, x, 0.0, w1=w1, w2=w2)
    for i in (3, 30):
        want = (g1[i].double() * (w1[i].double() * w2[i].double()).view(-1, 1, 1, 1)).sum(0) / steps * x[i].double()
        assert rel_inf(a4[i].cpu().numpy(), want.cpu().numpy()) <= 2e-6


def test_ig_accum_kernel_stamped_events(K):
    """xai_ig_accum_timed_f32: same result as the plain launch; the two events carry the kernel's own start / stop, so their
    elapsed time is positive and not longer than a pair of events bracketing the same launch."""
    gen = torch.Generator(device=DEV).manual_seed(9)
    g = torch.randn((8, 50, 3, 224, 224), device=DEV, generator=gen)
    x = torch.randn((8, 3, 224, 224), device=DEV, generator=gen)
    want, want_abs = K.ig_accum(g, x, 0.25, want_abs=True)
    for _ in range(3):
        b0, b1, k0, k1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        b0.record()
        got, got_abs = K.ig_accum(g, x, 0.25, want_abs=True, timing_events=(k0, k1))
        b1.record()
        torch.cuda.synchronize()
        assert torch.equal(got, want) and torch.equal(got_abs, want_abs)
        kernel_ms, bracket_ms = k0.elapsed_time(k1), b0.elapsed_time(b1)
        assert 0.0 < kernel_ms <= bracket_ms, (kernel_ms, bracket_ms)
        assert kernel_ms >= 240e6 / 8e12 * 1e3                      # 240 MB cannot move faster than the HBM peak: >= 0.03 ms


def test_idgi_kernels(K):
    g = load_golden("ig_small.npz")
    grads = dev(g["gradients"])
    sq = K.sumsq(grads)
    want_sq = (g["gradients"].astype(np.float64) ** 2).reshape(50, -1).sum(1)
    assert rel_inf(sq.cpu().numpy(), want_sq) <= 2e-6
    out = K.idgi_accum(grads, dev(g["logits"]), sq)
    assert rel_inf(out.cpu().numpy(), g["idgi"]) <= 5e-5      # logit differences amplify the GPU/CPU 1-ulp gap


# ------------------------------------------------------------------------------ K3
@pytest.mark.parametrize("B,C,h,w", [(1, 2048, 7, 7), (3, 64, 14, 14), (2, 17, 5, 9), (1, 8, 32, 32)])
def test_gradcam_and_upsample(K, B, C, h, w):
    from oracle import gradcam as ogc
    rng = np.random.default_rng(4)
    act = rng.standard_normal((B, C, h, w)).astype(np.float32)
    grad = rng.standard_normal((B, C, h, w)).astype(np.float32)
    for relu in (True, False):
        cam = K.gradcam(dev(act), dev(grad), relu=relu).cpu().numpy()
        want = ogc.cam_reduce(act, grad, relu=relu)
        # normalise by the un-ReLU'd magnitude: sums of ~C mixed-sign terms
        scale = np.abs(ogc.cam_reduce(act, grad, relu=False)).max()
        assert np.abs(cam - want).max() / scale <= TOL
    cam = K.gradcam(dev(act), dev(grad), relu=True)
    up = K.bilinear_up(cam, 224, 224).cpu().numpy()
    assert rel_inf(up, ogc.bilinear_up(cam.cpu().numpy(), 224, 224)) <= 2e-6
    sal = K.bilinear_up(cam, 224, 224, scale=3.0, take_abs=True).cpu().numpy()
    assert rel_inf(sal, ogc.gradcam_saliency(act, grad, 224, 224)) <= TOL
    # against the call the reference reaches (torch interpolate, antialias=True), on the GPU's own cam
    want = torch.nn.functional.interpolate(cam[None].cpu(), size=(224, 224), mode="bilinear", align_corners=False,
                                           antialias=True)[0].numpy()
    assert rel_inf(up, want) <= 2e-6


def test_gradcam_vs_reference_owned_cam_vectors(K):
    """tests/golden/cam.npz: produced by the reference's own ViT_CX CAM code (weights = spatial mean of the
    gradients, weighted channel sum, negatives clamped)."""
    g = load_golden("cam.npz")
    for tag in "abc":
        act, grad = dev(g[f"{tag}_act"]), dev(g[f"{tag}_grad"])
        scale = np.abs(g[f"{tag}_cam"]).max()
        assert np.abs(K.gradcam(act, grad, relu=False).cpu().numpy() - g[f"{tag}_cam"]).max() / scale <= TOL
        assert np.abs(K.gradcam(act, grad, relu=True).cpu().numpy() - g[f"{tag}_cam_relu"]).max() / scale <= TOL


# ------------------------------------------------------------------------------ K4 / K5
@pytest.mark.parametrize("H,W,s,N", [(224, 224, 8, 40), (30, 45, 7, 9), (32, 32, 4, 5)])
def test_rise_masks_and_accumulate(K, H, W, s, N):
    from oracle import rise as orise
    rng = np.random.RandomState(5)
    grid, shifts, cell = orise.draw_grid_and_shifts((H, W), N, s, 0.5, rng)
    image = np.random.default_rng(6).standard_normal((3, H, W)).astype(np.float32)
    g8, sh = dev(grid.astype(np.uint8)), dev(shifts)
    masked, masks = K.rise_apply(g8, sh, cell, dev(image), want_masked=True, want_masks=True)
    want_masks = orise.masks_from(grid, shifts, (H, W), cell)[:, 0]
    assert np.abs(masks.cpu().numpy() - want_masks).max() <= 1e-6          # masks live in [0,1]
    # masked = image * mask with the GPU's own mask is an exact fp32 product
    np.testing.assert_array_equal(masked.cpu().numpy(), image[None] * masks.cpu().numpy()[:, None])
    scores = np.random.default_rng(7).random(N).astype(np.float32)
    acc = K.rise_accum(g8, sh, dev(scores), cell, H, W, 1.0 / N / 0.5)
    want = (scores.reshape(-1, 1).astype(np.float64) * want_masks.reshape(N, -1)).sum(0).reshape(H, W) / N / 0.5
    assert rel_inf(acc.cpu().numpy(), want) <= 2e-6
    # accumulating in two calls == one call
    acc2 = K.rise_accum(g8[: N // 2], sh[: N // 2], dev(scores[: N // 2]), cell, H, W, 1.0 / N / 0.5)
    K.rise_accum(g8[N // 2:], sh[N // 2:], dev(scores[N // 2:]), cell, H, W, 1.0 / N / 0.5, acc=acc2)
    assert rel_inf(acc2.cpu().numpy(), acc.cpu().numpy()) <= 1e-12


def test_rise_full_rise_vs_oracle(K):
    from oracle import rise as orise
    from xai_engine.rise import rise
    H = W = 64
    N, s, p1 = 120, 8, 0.5
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.AdaptiveAvgPool2d(2),
                              torch.nn.Flatten(), torch.nn.Linear(16, 5)).to(DEV).eval()
    image = torch.randn(1, 3, H, W)
    score = lambda b: torch.softmax(net(b), 1)[:, 2]          # noqa: E731
    rng = np.random.RandomState(9)
    grid, shifts, cell = orise.draw_grid_and_shifts((H, W), N, s, p1, rng)
    got = rise(None, image, None, DEV, N=N, s=s, p1=p1, score_fn=score, masks=(grid.astype(np.uint8), shifts, cell))

    def score_np(b):
        with torch.no_grad():
            return score(torch.from_numpy(b).to(DEV)).cpu().numpy()
    want = orise.rise(score_np, image.numpy(), N, s, p1, grid, shifts, cell)
    assert rel_inf(got.cpu().numpy(), want) <= TOL
    # the host RNG stream is the reference's: same seed -> same draw
    from xai_engine.rise import draw_masks
    np.random.seed(11); a = draw_masks((H, W), 10, s, p1)
    np.random.seed(11); b = orise.draw_grid_and_shifts((H, W), 10, s, p1)
    np.testing.assert_array_equal(a[0], b[0].astype(np.uint8)); np.testing.assert_array_equal(a[1], b[1])


# ------------------------------------------------------------------------------ K8 / K6 / K10
@pytest.mark.parametrize("hw", [50176, 1024, 1000, 77, 1, 65536 + 3])
def test_rank_exact_with_ties_and_specials(K, hw):
    rng = np.random.default_rng(8)
    a = rng.standard_normal(hw).astype(np.float32)
    b = np.maximum(a, 0)                                           # ReLU'd: half the map ties at 0
    c = np.round(a * 4) / 4                                        # heavy ties, both signs, -0.0 present
    c[::7] = -0.0
    d = a.copy(); d[::5] = np.nan; d[1::11] = np.inf; d[2::13] = -np.inf
    sal = np.stack([a, b, c.astype(np.float32), d])
    order, rk = K.rank(dev(sal))
    order, rk = order.cpu().numpy(), rk.cpu().numpy()
    for i in range(sal.shape[0]):
        want = np.argsort(sal[i], kind="stable")
        np.testing.assert_array_equal(order[i], want)
        inv = np.empty(hw, dtype=np.int64); inv[want] = np.arange(hw)
        np.testing.assert_array_equal(rk[i], inv)


def test_rank_matches_reference_order_on_tie_free_maps(K):
    for name in ("perturb_small.npz", "perturb_224.npz"):
        g = load_golden(name)
        sal = g["saliency"].reshape(1, -1)
        order, _ = K.rank(dev(sal))
        np.testing.assert_array_equal(order.cpu().numpy()[0], g["salient_order_asc"][0])
        np.testing.assert_array_equal(order.cpu().numpy()[0][::-1], g["salient_order_desc"][0])


@pytest.mark.parametrize("fixture", ["perturb_small.npz", "perturb_224.npz"])
def test_perturb_batches_bit_exact_vs_reference_images(K, fixture):
    """K6 against the images the reference fed its model (all of them for 32x32; sha256 of
    every one of the 224 step images for 224x224)."""
    import hashlib
    g = load_golden(fixture)
    x = g["x"]
    hw = x.shape[-1] * x.shape[-2]
    step = int(g["step"])
    sal = dev(g["saliency"].reshape(1, -1))
    order, rk = K.rank(sal)
    blurred, zeros = g["substrate_blur"], np.zeros_like(x)
    for tag, desc, start, finish in (("MAS_ins", True, blurred, x), ("MAS_del", True, x, zeros), ("MAS_lerf", False, x, zeros)):
        flip = K.flip_steps(rk[0], desc, step)
        n_steps = (hw + step - 1) // step
        imgs = []
        at = 0
        for b in g[f"{tag}_batch_sizes"]:
            imgs.append(K.perturb_batch(dev(start[0]), dev(finish[0]), flip, at, int(b)).cpu().numpy())
            at += int(b)
        imgs = np.concatenate(imgs)
        assert imgs.shape[0] == n_steps
        sha = [hashlib.sha256(np.ascontiguousarray(im).tobytes()).hexdigest() for im in imgs]
        assert sha == list(g[f"{tag}_img_sha"]), tag
        if f"{tag}_images" in g:
            np.testing.assert_array_equal(imgs, g[f"{tag}_images"])
        np.testing.assert_array_equal(imgs[-1], finish[0])                 # last step == finish


def test_perturb_odd_shape_and_patch_steps(K):
    from oracle import perturb as op
    rng = np.random.default_rng(10)
    C, H, W, step = 2, 9, 11, 7                                            # hw = 99: scalar path, ragged last step
    start = rng.standard_normal((C, H, W)).astype(np.float32)
    finish = rng.standard_normal((C, H, W)).astype(np.float32)
    sal = rng.standard_normal(H * W).astype(np.float32)
    plan = op.Plan(H * W, step, 4, None)
    groups, _ = op.flip_groups(sal, H * W, plan, None, descending=True)
    want = np.stack(list(op.sequence(start[None], finish[None], groups)))
    _, rk = K.rank(dev(sal[None]))
    flip = K.flip_steps(rk[0], True, step)
    got = K.perturb_batch(dev(start), dev(finish), flip, 0, plan.n_steps).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    got_tail = K.perturb_batch(dev(start), dev(finish), flip, 5, 3).cpu().numpy()
    np.testing.assert_array_equal(got_tail, want[5:8])


def test_segment_sums(K):
    g = load_golden("perturb_224.npz")
    sal = g["saliency"].reshape(-1)
    order, _ = K.rank(dev(sal[None]))
    for desc in (True, False):
        seg, total = K.segment_sums(dev(sal), order[0], desc, 224, 224)
        o = np.argsort(sal, kind="stable")
        o = o[::-1] if desc else o
        want = np.array([sal[o[i * 224:(i + 1) * 224]].sum(dtype=np.float64) for i in range(224)])
        assert rel_inf(seg.cpu().numpy(), want) <= 2e-6
        assert abs(float(total[0]) - sal.sum(dtype=np.float64)) / sal.sum(dtype=np.float64) <= 2e-6


# ------------------------------------------------------------------------------ K7 / K9
def test_blur_vs_reference_conv2d(K):
    from oracle import perturb as op
    g = load_golden("kern.npz")
    for klen, sig, xk, wk in ((31, 31, "blur_x", "blur_31_31"), (11, 5, "blur_x", "blur_11_5"), (31, 31, "blur_small_x", "blur_small_31_31")):
        v = dev(op.gkern1d(klen, sig).astype(np.float32))
        got = K.blur_sep(dev(g[xk]), v).cpu().numpy()
        assert rel_inf(got, g[wk]) <= TOL, (klen, sig, rel_inf(got, g[wk]))


def test_softmax_stats(K):
    from oracle import perturb as op
    rng = np.random.default_rng(12)
    z = (rng.standard_normal((37, 1000)) * 3).astype(np.float32)
    p, ent, am = K.softmax_stats(dev(z), 5)
    want = op.softmax_rows(z)
    assert rel_inf(p.cpu().numpy(), want[:, 5]) <= 2e-6
    assert rel_inf(ent.cpu().numpy(), op.entropy_bits(want)) <= 2e-6
    np.testing.assert_array_equal(am.cpu().numpy(), z.argmax(1))
    p2, _, _ = K.softmax_stats(dev(z), None)                                # each row's own argmax
    assert rel_inf(p2.cpu().numpy(), want.max(1)) <= 2e-6
    t = torch.tensor([7], dtype=torch.int32, device=DEV)
    p3, _, _ = K.softmax_stats(dev(z), t)
    assert rel_inf(p3.cpu().numpy(), want[:, 7]) <= 2e-6
    # saturated row: p underflows to 0 -> entropy is NaN, like the reference's p*log2(p)
    zz = np.zeros((1, 10), dtype=np.float32); zz[0, 0] = 200.0
    _, e, _ = K.softmax_stats(dev(zz), 0)
    assert np.isnan(e.cpu().numpy()[0]) and np.isnan(op.entropy_bits(op.softmax_rows(zz))[0])
    # ties -> lowest index, 10-class rows (fewer classes than lanes)
    z10 = np.zeros((3, 10), dtype=np.float32); z10[1, 3] = z10[1, 8] = 2.0
    _, _, a = K.softmax_stats(dev(z10), 0)
    np.testing.assert_array_equal(a.cpu().numpy(), [0, 3, 0])


# ------------------------------------------------------------------------------ error behaviour
def test_cpu_tensors_are_refused_not_silently_computed(K):
    from xai_engine import XaiHipError
    with pytest.raises(XaiHipError):
        K.ig_interp(torch.zeros(1, 3, 4, 4), 0.0, torch.zeros(2))
    from xai_engine.ig import IG
    with pytest.raises(XaiHipError):
        IG(torch.zeros(1, 3, 8, 8), torch.nn.Identity(), 10, 5, 1, 0, "cpu", 0)


def test_abi_argument_errors(K):
    from xai_engine import load_library
    lib = load_library()
    assert lib.xai_ig_interp_f32(None, None, 0.0, None, 0, 1, 1, 4, None, None) == -1
    x = torch.zeros(8, device=DEV)
    assert lib.xai_ig_interp_f32(x.data_ptr(), None, 0.0, x.data_ptr(), 0, 0, 1, 4, x.data_ptr(), None) == -2
    assert lib.xai_gradcam_f32(x.data_ptr(), x.data_ptr(), 1, 1, 64, 64, 1, x.data_ptr(), None, 0, None) == -3
    assert lib.xai_blur_sep_f32(x.data_ptr(), x.data_ptr(), 4, 1, 1, 2, 2, x.data_ptr(), None) == -2
    p = x.data_ptr()
    assert lib.xai_bn_act_fwd_f32(p, None, p, p, p, None, 1e-5, None, None, None, None, 0.0, 9, 1, 1, 2, 4, p, None) == -1     # var missing
    assert lib.xai_bn_act_fwd_f32(p, None, p, p, p, p, 1e-5, None, None, None, None, 0.0, 99, 1, 1, 2, 4, p, None) == -2     # variant
    assert lib.xai_bn_act_fwd_f32(p, p, p, p, p, p, 1e-5, None, None, None, None, 0.0, 9, 0, 1, 2, 4, p, None) == -3         # identity needs relu
    assert lib.xai_bn_act_fwd_f32(p, None, p, p, p, p, 1e-5, p, p, p, p, 1e-5, 9, 1, 1, 2, 4, p, None) == -1                 # bn2 needs identity
    assert lib.xai_bn_relu_bwd_f32(p, None, p, p, p, 1e-5, p, p, 1e-5, 9, 1, 2, 4, p, None, None) == -1                     # bn2 needs g_identity
    assert lib.xai_maxpool_bwd_f32(p, p, 1, 2, 2, 1, 1, 0, 2, 0, p, None) == -2                                             # kernel 0
    assert lib.xai_bn_relu_maxpool_fwd_f32(p, p, p, p, p, 1e-5, 9, 70000, 1, 2, 2, 1, 1, 2, 2, 0, p, None) == -3           # > 65535 planes


# ------------------------------------------------------------------------------ f4 accumulators on K2's weighted form
def test_tis_and_vitcx_accumulators(K):
    from xai_engine.masked import tis_saliency, causal_saliency
    rng = np.random.default_rng(20)
    masks = (rng.random((1024, 196)) < 0.5).astype(np.float32)
    scores = rng.random(1024).astype(np.float32)
    got = tis_saliency(dev(scores), dev(masks)).cpu().numpy()
    m64, s64 = masks.astype(np.float64), scores.astype(np.float64)
    want = (s64[:, None] * m64).sum(0) / m64.sum(0)                      # TIS.py:337-356
    assert rel_inf(got, want) <= 2e-6
    fm = rng.random((130, 224 * 224)).astype(np.float32)                 # ViT-CX feature masks
    p = rng.standard_normal(130).astype(np.float32)
    got = causal_saliency(dev(p), dev(fm)).cpu().numpy()
    f64 = fm.astype(np.float64)
    want = p.astype(np.float64) @ (f64 / f64.sum(0)) / 130               # causal_score.py:57-60
    assert rel_inf(got, want) <= 5e-6


@pytest.mark.parametrize("N,P", [(1024, 196), (130, 224 * 224), (7, 30 * 45 + 1), (1, 4), (19, 3)])
def test_masked_sums_one_pass_equals_two_K2_launches_bit_for_bit(K, N, P):
    """K16 reads the mask stack once; the weighted and the plain mean it returns are bit-identical to the two K2 launches
    (weighted form / plain form) that served TIS and ViT-CX before, and to fp32 sequential sums."""
    rng = np.random.default_rng(N * 7 + P)
    rows = rng.random((N, P)).astype(np.float32)
    w = rng.standard_normal(N).astype(np.float32)
    weighted, plain = K.masked_sums(dev(rows), dev(w))
    g = dev(rows).view(1, N, 1, P)
    ones = torch.ones((1, 1, P), device=DEV)
    np.testing.assert_array_equal(weighted.cpu().numpy(), K.ig_accum(g, ones, 0.0, w1=dev(w).reshape(1, N))[0, 0].cpu().numpy())
    np.testing.assert_array_equal(plain.cpu().numpy(), K.ig_accum(g, ones, 0.0)[0, 0].cpu().numpy())
    aw, ap = np.zeros(P, np.float32), np.zeros(P, np.float32)
    for n in range(N):
        aw += rows[n] * w[n]
        ap += rows[n]
    np.testing.assert_array_equal(weighted.cpu().numpy(), aw / np.float32(N))
    np.testing.assert_array_equal(plain.cpu().numpy(), ap / np.float32(N))


# ------------------------------------------------------------------------------ K11-K14 feature-map maskers (ViT-CX)
@pytest.mark.parametrize("shape", [(12, 4, 4, 32, 32), (768, 14, 14, 224, 224), (5, 3, 7, 30, 45)])
def test_up_rownorm_vs_oracle(K, shape):
    from oracle import vit_cx as ocx
    R, h, w, H, W = shape
    fmap = np.random.default_rng(30).standard_normal((R, h, w)).astype(np.float32)
    got = K.up_rownorm(dev(fmap), H, W)
    torch.cuda.synchronize()
    want = ocx.norm_matrix(ocx.resize_maps(fmap, H, W).reshape(R, H * W))
    assert got.shape == (R, H * W)
    assert np.abs(got.cpu().numpy() - want).max() <= 2e-6              # values in [0,1]: absolute == relative to the row span
    assert float(got.min()) == 0.0 and float(got.max()) == 1.0


def test_rownorm_cluster_sum_and_causal_apply(K):
    from oracle import vit_cx as ocx
    from xai_engine.vit_cx import cluster_members
    g = load_golden("vit_cx.npz")
    assert rel_inf(K.rownorm(dev(g["act"])).cpu().numpy(), g["act_norm"]) <= 1e-6          # the reference's norm_matrix
    rng = np.random.default_rng(31)
    for R, P, n_cl in ((40, 1024, 7), (33, 1001, 5), (768, 224 * 224, 60)):
        rows = rng.random((R, P)).astype(np.float32)
        labels = rng.integers(0, n_cl, R)
        labels[:n_cl] = np.arange(n_cl)                                                      # every cluster non-empty
        members, offs = cluster_members(labels)
        got = K.cluster_sum(dev(rows), dev(members), dev(offs)).cpu().numpy()
        assert np.array_equal(got, ocx.cluster_sums(rows, labels))                           # same order of additions: bit-exact
    x, masks, noise = g["x"][0], g["masks"].reshape(6, -1), g["noise"]
    got = K.causal_apply(dev(x), dev(masks), dev(noise), 0.1).cpu().numpy()
    assert np.array_equal(got, ocx.causal_stack(x, masks, noise))                            # element-wise fp32: bit-exact
    x7 = rng.standard_normal((3, 30, 45)).astype(np.float32)
    m7 = rng.random((9, 30 * 45)).astype(np.float32)
    n7 = rng.standard_normal((9, 3, 30, 45)).astype(np.float32)
    assert np.array_equal(K.causal_apply(dev(x7), dev(m7), dev(n7), 0.1).cpu().numpy(), ocx.causal_stack(x7, m7, n7))
    lib = __import__("xai_engine")._lib.load()
    t = torch.zeros(8, device=DEV)
    assert lib.xai_up_rownorm_f32(t.data_ptr(), 1, 100, 100, 4, 4, t.data_ptr(), None) == -3
    assert lib.xai_cluster_sum_f32(t.data_ptr(), None, t.data_ptr(), 1, 4, t.data_ptr(), None) == -1
    assert lib.xai_causal_apply_f32(t.data_ptr(), t.data_ptr(), t.data_ptr(), 0, 3, 4, 0.1, t.data_ptr(), None) == -2


# ------------------------------------------------------------------------------ size-independent properties at full size
def test_full_size_perturbation_properties(K):
    """224x224, 224 steps, heavy ties (ReLU'd map): every step flips exactly `step` new pixels, the flipped
    set is monotone, the last image is `finish`, ascending and descending orders are mirror images."""
    H = W = 224
    rng = np.random.default_rng(30)
    sal = np.maximum(rng.standard_normal(H * W), 0).astype(np.float32)    # ~50 % exact zeros
    start = rng.standard_normal((3, H, W)).astype(np.float32)
    finish = start + 1.0
    order, rk = K.rank(dev(sal[None]))
    o = order.cpu().numpy()[0]
    assert (np.diff(sal[o]) >= 0).all()                                   # sorted
    tie = sal[o][:-1] == sal[o][1:]
    assert (np.diff(o)[tie] > 0).all()                                    # stable: ties in index order
    np.testing.assert_array_equal(np.sort(o), np.arange(H * W))           # a permutation
    f_desc = K.flip_steps(rk[0], True, 224)
    f_asc = K.flip_steps(rk[0], False, 224)
    np.testing.assert_array_equal(f_desc.cpu().numpy() + f_asc.cpu().numpy(), 223)    # mirror images (HW = 224*224)
    imgs = K.perturb_batch(dev(start), dev(finish), f_desc, 0, 224)
    flipped = (imgs != dev(start)[None]).all(1).reshape(224, -1)          # (step, pixel)
    counts = flipped.sum(1).cpu().numpy()
    np.testing.assert_array_equal(counts, 224 * np.arange(1, 225))
    assert bool((flipped[1:] | ~flipped[:-1]).all())                      # once flipped, stays flipped
    np.testing.assert_array_equal(imgs[-1].cpu().numpy(), finish)
    seg, total = K.segment_sums(dev(sal), order[0], True, 224, 224)
    assert abs(float(seg.sum()) - float(total[0])) <= 1e-3 * float(total[0])


def test_full_size_rise_properties(K):
    """N = 8000 masks on 224x224: accumulation is linear in the scores and equals N*p1*scale for unit scores
    up to the sampling noise of the masks' mean; mask values stay in [0, 1]."""
    from xai_engine.rise import draw_masks
    np.random.seed(0)
    grid, shifts, cell = draw_masks((224, 224), 8000, 8, 0.5)
    g8, sh = dev(grid), dev(shifts)
    ones = torch.ones(8000, device=DEV)
    a1 = K.rise_accum(g8, sh, ones, cell, 224, 224, 1.0 / 8000 / 0.5)
    assert abs(float(a1.mean()) - 1.0) < 0.02                              # E[mask] = p1
    s = torch.rand(8000, device=DEV)
    a2 = K.rise_accum(g8, sh, s, cell, 224, 224, 1.0)
    a3 = K.rise_accum(g8, sh, 2 * s + ones, cell, 224, 224, 1.0)
    assert rel_inf(a3.cpu().numpy(), (2 * a2 + a1 * 8000 * 0.5).cpu().numpy()) <= 1e-6      # 2*s+1 itself rounds in fp32
    m = K.rise_apply(g8[:64], sh[:64], cell, torch.ones(1, 224, 224, device=DEV), want_masked=False, want_masks=True)
    assert float(m.min()) >= 0.0 and float(m.max()) <= 1.0
    # store policy: 240 masks x 3 x 224 x 224 = 144 MB of masked images go out with non-temporal stores, 80 masks (48 MB) with
    # normal ones -- same arithmetic, so the large launch equals its three thirds bit for bit, and the masks equal the oracle's
    from oracle import rise as orise
    img = torch.randn(3, 224, 224, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    big, big_m = K.rise_apply(g8[:240], sh[:240], cell, img, want_masked=True, want_masks=True)
    for lo in (0, 80, 160):
        part, part_m = K.rise_apply(g8[lo:lo + 80], sh[lo:lo + 80], cell, img, want_masked=True, want_masks=True)
        assert torch.equal(big[lo:lo + 80], part) and torch.equal(big_m[lo:lo + 80], part_m)
    want = orise.masks_from(grid[:4].astype(np.float32), shifts[:4], (224, 224), cell)[:, 0]
    assert np.abs(big_m[:4].cpu().numpy() - want).max() <= 1e-6


def test_rise_with_a_device_side_mask_draw(K):
    """draw_masks_on_device: the performance-mode draw (device Philox instead of the reference's NumPy stream): same kernels,
    statistically the same masks -- E[mask] = p1, and the RISE map of a linear scorer agrees with the host-draw map within the
    Monte-Carlo error of N = 4000 masks."""
    from xai_engine.rise import draw_masks, draw_masks_on_device, rise
    gen = torch.Generator(device=DEV).manual_seed(5)
    grid, shifts, cell = draw_masks_on_device((64, 64), 4000, 8, 0.5, DEV, gen)
    assert grid.is_cuda and grid.dtype == torch.uint8 and shifts.dtype == torch.int32 and cell.tolist() == [8, 8]
    assert int(shifts.min()) >= 0 and int(shifts.max()) <= 7 and abs(float(grid.float().mean()) - 0.5) < 0.01
    image = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(6))
    w = torch.randn(3, 64, 64, generator=torch.Generator().manual_seed(7)).to(DEV)
    score = lambda b: (b * w).flatten(1).sum(1)                                   # noqa: E731   linear in the mask
    on_dev = rise(None, image, None, DEV, N=4000, s=8, p1=0.5, score_fn=score, batch_size=500, masks=(grid, shifts, cell))
    np.random.seed(8)
    on_host = rise(None, image, None, DEV, N=4000, s=8, p1=0.5, score_fn=score, batch_size=500, masks=draw_masks((64, 64), 4000, 8, 0.5))
    a, b = on_dev.cpu().numpy(), on_host.cpu().numpy()
    assert np.isfinite(a).all() and np.corrcoef(a.ravel(), b.ravel())[0, 1] > 0.9
    gen2 = torch.Generator(device=DEV).manual_seed(5)
    again = draw_masks_on_device((64, 64), 4000, 8, 0.5, DEV, gen2)
    assert torch.equal(again[0], grid) and torch.equal(again[1], shifts)          # reproducible under a seeded generator


# ------------------------------------------------------------------------------ ABI conventions: streams and graphs
def test_kernels_run_on_the_callers_stream_and_are_graph_capturable(K):
    """include/xai_hip.h promises: launches go to the stream handed in, never synchronise or allocate,
    so they can be captured into a hipGraph and replayed."""
    rng = np.random.default_rng(40)
    g1 = dev(rng.standard_normal((2, 10, 3, 32, 32)).astype(np.float32))
    x = dev(rng.standard_normal((2, 3, 32, 32)).astype(np.float32))
    sal = dev(rng.standard_normal((1, 1024)).astype(np.float32))
    start, finish = x[0].contiguous(), x[1].contiguous()
    eager = K.ig_accum(g1, x, 0.0)
    order_e, rank_e = K.rank(sal)
    flip_e = K.flip_steps(rank_e[0], True, 32)
    imgs_e = K.perturb_batch(start, finish, flip_e, 0, 8)
    # (1) a side stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out_s = K.ig_accum(g1, x, 0.0)
        order_s, _ = K.rank(sal)
    side.synchronize()
    np.testing.assert_array_equal(out_s.cpu().numpy(), eager.cpu().numpy())
    np.testing.assert_array_equal(order_s.cpu().numpy(), order_e.cpu().numpy())
    # (2) capture + two replays with fresh inputs written into the captured buffers
    graph = torch.cuda.CUDAGraph()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out_g = K.ig_accum(g1, x, 0.0)
            order_g, rank_g = K.rank(sal)
            flip_g = K.flip_steps(rank_g[0], True, 32)
            imgs_g = K.perturb_batch(start, finish, flip_g, 0, 8)
    for seed in (41, 42):
        r = np.random.default_rng(seed)
        g1.copy_(dev(r.standard_normal((2, 10, 3, 32, 32)).astype(np.float32)))
        sal.copy_(dev(r.standard_normal((1, 1024)).astype(np.float32)))
        graph.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out_g.cpu().numpy(), K.ig_accum(g1, x, 0.0).cpu().numpy())
        np.testing.assert_array_equal(order_g.cpu().numpy()[0], np.argsort(sal.cpu().numpy()[0], kind="stable"))
        o2, r2 = K.rank(sal)
        np.testing.assert_array_equal(imgs_g.cpu().numpy(), K.perturb_batch(start, finish, K.flip_steps(r2[0], True, 32), 0, 8).cpu().numpy())


def test_graph_replays_see_a_fresh_zero_fill_every_time(K):
    """K8's histogram words must be re-zeroed by EVERY replay of a captured sort.  On the HIP runtime bundled with the torch
    wheel a memset graph node only takes effect on the first launch (profiles/r02_repro_graph_memset3.txt),
    so the library zero-fills with a kernel; here the workspace is dirtied between replays, which makes a skipped zero-fill
    a wrong sort instead of a lucky one."""
    from xai_engine import _lib
    lib = _lib.load()
    sal = dev(np.random.default_rng(60).standard_normal((2, 5000)).astype(np.float32))
    ws = torch.empty(lib.xai_rank_workspace_bytes(2, 5000), dtype=torch.uint8, device=DEV)
    order = torch.empty((2, 5000), dtype=torch.int32, device=DEV)
    rk = torch.empty_like(order)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            rc = lib.xai_rank_f32(sal.data_ptr(), 2, 5000, order.data_ptr(), rk.data_ptr(), ws.data_ptr(), ws.numel(), side.cuda_stream)
    assert rc == 0
    for seed in range(61, 66):
        sal.copy_(dev(np.random.default_rng(seed).standard_normal((2, 5000)).astype(np.float32)))
        ws.fill_(0xAB)
        graph.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(order.cpu().numpy(), np.argsort(sal.cpu().numpy(), axis=1, kind="stable"))


def test_plain_c_host_drives_the_abi_without_torch(tmp_path):
    """examples/abi_host.c: a C99 program with hipMalloc'd buffers calls xai_ig_accum_f32 / xai_rank_f32 / xai_flip_steps_i32 /
    xai_perturb_batch_f32 and checks them against the reference expressions itself -- the C ABI as a cgo / JNI / ctypes binding
    would see it, on the system's HIP runtime rather than the one the torch wheel bundles."""
    import subprocess
    from test_cpu_host import _build_abi_host
    exe = _build_abi_host(tmp_path / "abi_host")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi host ok" in r.stdout


# ------------------------------------------------------------------------------ degenerate and ragged shapes
def test_edge_shapes(K):
    from oracle import ig as oig, perturb as op, rise as orise
    from xai_engine import load_library
    lib = load_library()
    rng = np.random.default_rng(50)
    # one step, one image, one pixel
    g = rng.standard_normal((1, 1, 1, 1, 1)).astype(np.float32); x = rng.standard_normal((1, 1, 1, 1)).astype(np.float32)
    np.testing.assert_array_equal(K.ig_accum(dev(g), dev(x), 0.0).cpu().numpy(), g[:, 0] * x)
    np.testing.assert_array_equal(K.ig_interp(dev(x), 0.5, dev(np.array([0.25], np.float32))).cpu().numpy()[0, 0],
                                  oig.interpolate(x[0], np.full_like(x[0], 0.5), np.array([0.25], np.float32))[0])
    # a batch that is not a multiple of anything: 7 steps x 3 x 5 x 7 image, ragged last perturbation step
    start = rng.standard_normal((3, 5, 7)).astype(np.float32); finish = -start
    sal = rng.standard_normal(35).astype(np.float32)
    _, rk = K.rank(dev(sal[None]))
    flip = K.flip_steps(rk[0], False, 4)                                  # 9 steps, last one has 3 pixels
    plan = op.Plan(35, 4, 50, None)
    groups, _ = op.flip_groups(sal, 35, plan, None, descending=False)
    want = np.stack(list(op.sequence(start[None], finish[None], groups)))
    np.testing.assert_array_equal(K.perturb_batch(dev(start), dev(finish), flip, 0, 9).cpu().numpy(), want)
    np.testing.assert_array_equal(K.perturb_batch(dev(start), dev(finish), flip, 8, 1).cpu().numpy(), want[8:9])
    seg, total = K.segment_sums(dev(sal), K.rank(dev(sal[None]))[0][0], False, 4, 9)
    assert abs(float(seg.sum()) - float(sal.sum())) <= 1e-5 * np.abs(sal).sum()
    # single mask, single class, single-tap blur
    grid, shifts, cell = orise.draw_grid_and_shifts((16, 16), 1, 4, 0.5, np.random.RandomState(1))
    m = K.rise_apply(dev(grid.astype(np.uint8)), dev(shifts), cell, dev(np.ones((1, 16, 16), np.float32)), want_masked=False, want_masks=True)
    assert np.abs(m.cpu().numpy()[0] - orise.masks_from(grid, shifts, (16, 16), cell)[0, 0]).max() <= 1e-6
    p, e, a = K.softmax_stats(dev(np.array([[3.0]], np.float32)), 0)
    assert float(p[0]) == 1.0 and float(e[0]) == 0.0 and int(a[0]) == 0
    xb = rng.standard_normal((1, 1, 3, 3)).astype(np.float32)
    np.testing.assert_array_equal(K.blur_sep(dev(xb), dev(np.array([2.0], np.float32))).cpu().numpy(), xb * 4.0)
    # empty / negative extents are argument errors, not launches
    t = torch.zeros(4, device=DEV)
    assert lib.xai_perturb_batch_f32(t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 4, 0, 0, t.data_ptr(), None) == -2
    assert lib.xai_rise_accum_f64(t.data_ptr(), t.data_ptr(), t.data_ptr(), 0, 8, 28, 28, 224, 224, 1.0, t.data_ptr(), None) == -2
    assert lib.xai_softmax_stats_f32(t.data_ptr(), 1, 4, None, 9, t.data_ptr(), None, None, None) == -2
    assert lib.xai_ig_accum_f32(t.data_ptr(), 1, 4, None, 5, None, None, t.data_ptr(), None, 0.0, 1, 1, t.data_ptr(), None, None) == -2


def test_long_blur_kernels_two_pass(K):
    """klen > 63 (the MDA growing-kernel search reaches 101 taps): two 1-D passes == dense zero-padded conv2d."""
    from oracle import perturb as op
    rng = np.random.default_rng(60)
    x = rng.standard_normal((1, 3, 40, 52)).astype(np.float32)
    for klen, sig in ((67, 67), (101, 101)):
        v = op.gkern1d(klen, sig)
        got = K.blur_sep(dev(x), dev(v.astype(np.float32))).cpu().numpy()
        kern = torch.from_numpy(op.gkern(klen, sig))
        want = torch.nn.functional.conv2d(torch.from_numpy(x), kern, padding=klen // 2).numpy()       # the reference's call
        assert rel_inf(got, want) <= TOL, (klen, rel_inf(got, want))
    # both code paths agree at the hand-over length
    v63 = dev(op.gkern1d(63, 20).astype(np.float32))
    xd = dev(x)
    fused = K.blur_sep(xd, v63)
    tmp = torch.empty_like(fused); out = torch.empty_like(fused)
    K._call("xai_blur_1d_f32", fused.device, xd.data_ptr(), v63.data_ptr(), 63, 1, 1, 3, 40, 52, tmp.data_ptr())
    K._call("xai_blur_1d_f32", fused.device, tmp.data_ptr(), v63.data_ptr(), 63, 0, 1, 3, 40, 52, out.data_ptr())
    np.testing.assert_array_equal(fused.cpu().numpy(), out.cpu().numpy())


def test_blur_until_unconfident(K):
    from xai_engine.blur import blur_until_unconfident, GaussianBlur
    from helpers import tiny_from
    g = load_golden("perturb_224.npz")
    model = tiny_from(g, DEV)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        t = int(model(x.to(DEV)).argmax(1)[0])
    blur, klen, ksig, pct = blur_until_unconfident(model, x, t, DEV, threshold_pct=0.0)     # never satisfied -> runs to the cap
    assert klen == 103 and ksig == 103 and isinstance(blur, GaussianBlur)                   # 31 + 18*4, first value > 101
    blur2, klen2, _, pct2 = blur_until_unconfident(model, x, t, DEV, threshold_pct=100.0)   # satisfied immediately
    assert klen2 == 31 and pct2 <= 100.0


def test_selfcheck_module(K):
    from xai_engine import selfcheck
    ok, results = selfcheck.run(DEV, verbose=False)
    assert ok, [r for r in results if not r[3]]
    assert len(results) >= 18


def test_stream_workers_run_jobs_in_order_on_their_own_streams_and_hand_errors_back(K):
    """xai_engine.streams: job i runs on worker i % n, i.e. on that worker's host thread and HIP stream, with autograd inline; results
    come back in job order; the caller's stream waits for the workers; a failing job re-raises in the caller after all jobs are in."""
    import threading
    from xai_engine.streams import run_on_streams, workers, backward_turn
    dev = torch.device(DEV)
    ws = workers(dev, 3)
    assert len({w.ident for w in ws}) == 3 and len({w.stream.cuda_stream for w in ws}) == 3
    src = torch.arange(12, dtype=torch.float32, device=DEV)

    def job(i):
        me = threading.current_thread()
        assert me is ws[i % 3] and torch.cuda.current_stream(dev) == me.stream
        assert torch.is_grad_enabled() and not torch.autograd.is_multithreading_enabled()      # backward nodes run on this thread
        x = (src[i:i + 1] * 2).requires_grad_(True)
        with backward_turn(dev):                                                               # a no-op on a worker
            (g,) = torch.autograd.grad((x * x).sum(), x)
        return i, g
    out = run_on_streams(dev, 3, [lambda i=i: job(i) for i in range(12)], kind="test")
    assert [o[0] for o in out] == list(range(12))
    got = torch.cat([o[1] for o in out])                       # consumed on the caller's stream: run_on_streams made it wait for the workers
    np.testing.assert_array_equal(got.cpu().numpy(), 4.0 * np.arange(12, dtype=np.float32))
    assert torch.autograd.is_multithreading_enabled()          # thread-local: the caller's autograd mode is untouched

    def bad(i):
        if i == 4:
            raise ValueError("job 4 failed")
        return i
    with pytest.raises(ValueError, match="job 4 failed"):
        run_on_streams(dev, 3, [lambda i=i: bad(i) for i in range(8)])
    assert run_on_streams(dev, 2, [lambda: 7]) == [7]          # the workers are still alive afterwards
    # nested use (an attribution that itself fans out, called from a worker): runs on the calling worker, no dead-lock
    nested = run_on_streams(dev, 2, [lambda: run_on_streams(dev, 3, [lambda j=j: (j, threading.current_thread().name) for j in range(4)])])
    assert [j for j, _ in nested[0]] == [0, 1, 2, 3] and len({name for _, name in nested[0]}) == 1


def test_reentrant_from_two_host_threads_on_two_streams(K):
    """include/xai_hip.h: no global mutable state, re-entrant from several host threads on different streams."""
    import threading
    rng = np.random.default_rng(70)
    inputs = []
    for i in range(2):
        g = dev(rng.standard_normal((2, 12, 3, 48, 48)).astype(np.float32))
        x = dev(rng.standard_normal((2, 3, 48, 48)).astype(np.float32))
        sal = dev(rng.standard_normal((2, 48 * 48)).astype(np.float32))
        inputs.append((g, x, sal))
    want = [(K.ig_accum(g, x, 0.0).cpu().numpy(), K.rank(sal)[0].cpu().numpy()) for g, x, sal in inputs]
    torch.cuda.synchronize()
    errors = []

    def worker(i):
        try:
            g, x, sal = inputs[i]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(25):
                    out = K.ig_accum(g, x, 0.0)
                    order, _ = K.rank(sal)
                st.synchronize()
                np.testing.assert_array_equal(out.cpu().numpy(), want[i][0])
                np.testing.assert_array_equal(order.cpu().numpy(), want[i][1])
        except Exception as e:          # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in threads]
    [t.join(timeout=120) for t in threads]
    assert not errors, errors
