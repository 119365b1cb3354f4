"""Pin the oracle (oracle/*.py) to vectors produced by the reference itself
(tests/golden/*.npz, made by tests/golden/make_golden.py).  CPU only.

Tolerances: integer/index/byte results are compared exactly.  Floating results are
float32 restated with NumPy instead of torch, so sums may associate differently:
||a-b||inf/||b||inf <= 2e-6, an order below the 1e-5 parity bar of BASELINE.json.
"""
import hashlib

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_inf
from helpers import tiny_from, logits_fn_of
from oracle import ig as oig
from oracle import perturb as op
from oracle import gradcam as ogc
from oracle import rise as orise

TOL = 2e-6


# ------------------------------------------------------------------ IG family
@pytest.mark.parametrize("name", ["ig_small.npz", "ig_224.npz"])
def test_ig_and_left_ig(name):
    g = load_golden(name)
    model = tiny_from(g)
    x, t = g["x"], int(g["target"])
    out, grads, logits, n_use = oig.ig(x, model, 50, 25, 1, 0, t, return_path=True)
    assert n_use == 50
    assert rel_inf(logits, g["logits"]) <= TOL
    assert rel_inf(out, g["ig"]) <= TOL
    if "gradients" in g:
        assert rel_inf(grads, g["gradients"]) <= TOL
        # K2 alone, on the reference's own per-step gradients
        assert rel_inf(oig.accumulate(g["gradients"], 50, x[0], np.zeros_like(x[0])), g["ig"]) <= TOL
    assert rel_inf(oig.ig(x, model, 50, 25, 0.9, 0, t), g["lig"]) <= TOL
    assert rel_inf(oig.ig(x, model, 50, 50, 1, g["baseline_tensor"], t), g["ig_tensor_baseline"]) <= TOL
    out, _, lg, n_use = oig.ig(x, model, 50, 10, 0.5, 0.25, t, return_path=True)
    assert rel_inf(lg, g["lig_a05_b025_logits"]) <= TOL
    assert 1 <= n_use < 50
    assert rel_inf(out, g["lig_a05_b025"]) <= TOL


def test_ig_bad_batch_returns_four_zeros():
    g = load_golden("ig_small.npz")
    assert oig.ig(g["x"], tiny_from(g), 50, 7, 1, 0, int(g["target"])) == (0, 0, 0, 0)


def test_left_cutoff_rules():
    lg = np.array([0.1, 0.2, 0.95, 1.0], dtype=np.float32)
    assert oig.left_cutoff(lg, 0.9) == 2
    assert oig.left_cutoff(np.array([1.0, 0.5], dtype=np.float32), 0.9) == 1      # hit at 0 -> 1
    assert oig.left_cutoff(np.array([-1.0, -2.0], dtype=np.float32), 0.9) == 1    # max<=0: -1 > -0.9 false, -2 false -> none -> 1


def test_smoothgrad_against_the_seeded_reference_run():
    """tests/golden/smoothgrad.npz: the reference's smoothGrad("IG", ..., vis=True) under torch.manual_seed(77)."""
    g, gi = load_golden("smoothgrad.npz"), load_golden("ig_small.npz")
    model = tiny_from(gi)
    for tag in ("a", "b"):
        torch.manual_seed(int(g["seed"]))
        mean, total, noisy = oig.smoothgrad_ig(gi["x"], model, int(g[f"{tag}_steps"]), float(g[f"{tag}_baseline"]), int(gi["target"]),
                                               sigma_spread=float(g[f"{tag}_sigma_spread"]), samples=int(g[f"{tag}_samples"]))
        np.testing.assert_array_equal(noisy, g[f"{tag}_noisy_imgs"])                   # same generator, same draw, same add
        assert rel_inf(total, g[f"{tag}_total_gradients"]) <= TOL
        assert rel_inf(mean, g[f"{tag}_mean"]) <= TOL
        assert np.array_equal(total[:, 0], total[:, 1]) and np.array_equal(total[:, 0], total[:, 2])     # the :196 quirk


def test_idg_family():
    g = load_golden("ig_small.npz")
    model = tiny_from(g)
    x, t = g["x"], int(g["target"])
    sl, dx = oig.slopes(x[0], np.zeros_like(x[0]), model, 50, 25, t)
    assert dx == pytest.approx(float(g["slope_step"]), rel=0, abs=0)
    assert rel_inf(sl, g["slopes"]) <= 1e-4          # differences of nearly equal logits amplify rounding
    al, sub = oig.alpha_parameters(g["slopes"], 50, float(g["slope_step"]))
    np.testing.assert_array_equal(al, g["idg_alphas"])
    np.testing.assert_array_equal(sub, g["idg_substep"])
    assert rel_inf(oig.idgi(x, model, 50, 25, 0, t), g["idgi"]) <= 1e-4
    assert rel_inf(oig.idg(x, model, 50, 25, 0, t), g["idg"]) <= 1e-3


def test_input_grad():
    g = load_golden("ig_small.npz")
    gr, _ = oig.grads_and_logits(tiny_from(g), g["x"], int(g["target"]))
    assert rel_inf(gr[0], g["input_grad"]) <= TOL


# ------------------------------------------------------------------ blur kernel, auc
def test_gkern_blur_auc():
    g = load_golden("kern.npz")
    np.testing.assert_array_equal(op.gkern(31, 31), g["gkern_31_31"])
    np.testing.assert_array_equal(op.gkern(11, 5), g["gkern_11_5"])
    for klen, sig in ((31, 31), (11, 5)):
        v = op.gkern1d(klen, sig)
        assert rel_inf(np.outer(v, v), g[f"gkern_{klen}_{sig}"][0, 0]) <= 1e-7     # separable
        assert abs(g[f"gkern_{klen}_{sig}"][1, 1].sum() - 1.0) < 1e-6
    assert rel_inf(op.blur_dense(g["blur_x"], g["gkern_11_5"]), g["blur_11_5"]) <= TOL
    assert rel_inf(op.blur_dense(g["blur_small_x"], g["gkern_31_31"]), g["blur_small_31_31"]) <= TOL
    assert rel_inf(op.blur_dense(g["blur_x"], g["gkern_31_31"]), g["blur_31_31"]) <= TOL
    for c, v in zip(g["auc_curves"], g["auc_values"]):
        assert op.auc(c) == v
    assert op.auc(np.linspace(0, 1, 225)) == pytest.approx(0.5, abs=1e-15)
    assert float(g["auc_linspace"]) == pytest.approx(0.5, abs=1e-15)


# ------------------------------------------------------------------ ins/del
def _blur_fn(g):
    kern = op.gkern(int(g["blur_klen"]), int(g["blur_sig"]))
    return lambda im: op.blur_dense(im, kern)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest()


CASES = [
    ("MAS_ins", "mas", "ins", True), ("MAS_del", "mas", "del", False), ("MAS_lerf", "mas", "lerf", False),
    ("MAS_morf", "mas", "morf", False), ("RISE_ins", "rise_metric", "ins", True),
    ("RISE_del", "rise_metric", "del", False), ("RISE_lerf", "rise_metric", "lerf", False),
    ("AIC_ins", "aic", "ins", True), ("AIC_del", "aic", "del", False),
    ("PNP_lerf", "pnp", "lerf", False), ("PNP_morf", "pnp", "morf", False),
    ("MONO_positive", "mono", "positive", True), ("MONO_negative", "mono", "negative", False),
]


@pytest.mark.parametrize("fixture", ["perturb_small.npz", "perturb_patch.npz", "perturb_224.npz"])
def test_pixel_sequences_bit_exact(fixture):
    """Every perturbed image the reference fed its model, byte for byte (sha256)."""
    g = load_golden(fixture)
    x, sal = g["x"], g["saliency"]
    HW = x.shape[-1] * x.shape[-2]
    pm = g["patch_mask"] if "patch_mask" in g else None
    zeros = np.zeros_like(x)
    # the reference's substrate (torch conv2d) is taken from the fixture so that the byte
    # comparison isolates the pixel surgery from blur rounding
    blurred = g["substrate_blur"]
    if pm is None:
        np.testing.assert_array_equal(op.pixel_order(sal, HW, True), g["salient_order_desc"][0])
        np.testing.assert_array_equal(op.pixel_order(sal, HW, False), g["salient_order_asc"][0])
    for tag, _, mode, uses_blur in CASES:
        plan = op.Plan(HW, int(g["step"]), int(g["max_bs"]), pm, always_leftover=tag.startswith("MONO"))
        np.testing.assert_array_equal(np.array(plan.batches), g[f"{tag}_batch_sizes"])
        inserting = mode in ("ins", "positive")
        start, finish = (blurred, x) if inserting else (x, zeros)
        groups, _ = op.flip_groups(sal, HW, plan, pm, descending=(mode != "lerf"))
        imgs = list(op.sequence(start, finish, groups))
        assert [_sha(i) for i in imgs] == list(g[f"{tag}_img_sha"]), tag
        if f"{tag}_images" in g:
            np.testing.assert_array_equal(np.stack(imgs), g[f"{tag}_images"])
        # size-independent properties
        np.testing.assert_array_equal(imgs[-1], finish[0])
        if pm is None:
            changed = (imgs[1] != imgs[0]).reshape(3, -1).any(0).sum()
            assert changed <= int(g["step"])


@pytest.mark.parametrize("fixture", ["perturb_small.npz", "perturb_patch.npz", "perturb_224.npz"])
def test_metric_return_tuples(fixture):
    g = load_golden(fixture)
    model = tiny_from(g)
    fn = logits_fn_of(model)
    x, sal = g["x"], g["saliency"]
    pm = g["patch_mask"] if "patch_mask" in g else None
    blur = _blur_fn(g)
    zeros = np.zeros_like
    for tag, func, mode, uses_blur in CASES:
        res = getattr(op, func)(fn, x, sal, mode, int(g["step"]), blur if uses_blur else zeros, pm, int(g["max_bs"]))
        for i, r in enumerate(res):
            want = g[f"{tag}_ret{i}"]
            if np.ndim(want) == 0 and float(want) == int(want) and func != "mono":
                assert int(r) == int(want), (tag, i)
            else:
                assert rel_inf(r, want) <= 2e-5, (tag, i, rel_inf(r, want))
    if pm is None:
        score, resp = op.aic(fn, x, sal, "del", int(g["step"]), zeros, None, int(g["max_bs"]), decision_flip=True)
        assert score == float(g["AIC_delflip_ret0"])
        np.testing.assert_array_equal(resp, g["AIC_delflip_ret1"])


def test_sweep_counter():
    g = load_golden("sweep_small.npz")
    fn = logits_fn_of(tiny_from(g))
    kern = op.gkern(31, 31)
    blur = lambda im: op.blur_dense(im, kern)     # noqa: E731
    total = np.zeros(10)
    for i in range(3):
        c = op.run_perturbation(fn, g["x"][i:i + 1], g["saliency"][i], 32, blur, 50)
        got = np.array([c[k] for k in op.SWEEP_KEYS])
        assert list(op.SWEEP_KEYS) == list(g["keys"])
        assert np.abs(got - g[f"counter_{i}"]).max() <= 2e-5
        total += got
    assert np.abs(total - g["counter_sum"]).max() <= 5e-5


def test_sweep_counter_reference_fold():
    """sweep_counter.npz: the oracle's ten numbers per image against the reference's, and the oracle's restatement of the
    reference's Counter `+=` fold / CSV rows (evaluatePerturbation.py:594-596,612-615) against the reference-made rows."""
    g = load_golden("sweep_counter.npz")
    fn = logits_fn_of(tiny_from(g))
    kern = op.gkern(31, 31)
    blur = lambda im: op.blur_dense(im, kern)     # noqa: E731
    n = int(g["images_used"])
    for i in range(n):
        c = op.run_perturbation(fn, g["x"][i:i + 1], g["saliency"][i], 32, blur, 50)
        assert np.abs(np.array([c[k] for k in op.SWEEP_KEYS]) - g[f"counter_{i}"]).max() <= 2e-5
    keys, values = op.reference_fold([g[f"counter_{i}"] for i in range(n)])
    assert keys == g["csv_keys"].tolist()
    assert [str(v / n) for v in values] == g["csv_values"].tolist()
    assert keys[-1] == "AIC_ins" and keys.index("MONO_pos") == 7                        # dropped and re-entered at the end


# ------------------------------------------------------------------ unpinned pieces: internal consistency
def test_bilinear_matches_torch_interpolate():
    """The call the reference reaches through torchvision Resize(antialias=True)."""
    rng = np.random.default_rng(0)
    src = rng.standard_normal((2, 7, 7)).astype(np.float32)
    want = torch.nn.functional.interpolate(torch.from_numpy(src)[None], size=(224, 224), mode="bilinear",
                                           align_corners=False, antialias=True)[0].numpy()
    assert rel_inf(ogc.bilinear_up(src, 224, 224), want) <= TOL
    want = torch.nn.functional.interpolate(torch.from_numpy(src)[None], size=(30, 45), mode="bilinear",
                                           align_corners=False)[0].numpy()
    assert rel_inf(ogc.bilinear_up(src, 30, 45), want) <= TOL


def test_gradcam_against_autograd_definition():
    rng = np.random.default_rng(1)
    act = rng.standard_normal((2, 16, 7, 7)).astype(np.float32)
    grad = rng.standard_normal((2, 16, 7, 7)).astype(np.float32)
    a, g_ = torch.from_numpy(act), torch.from_numpy(grad)
    want = torch.relu((g_.mean(dim=(2, 3), keepdim=True) * a).sum(1)).numpy()
    assert rel_inf(ogc.cam_reduce(act, grad), want) <= TOL
    sal = ogc.gradcam_saliency(act, grad, 224, 224)
    assert sal.shape == (2, 224, 224) and (sal >= 0).all()


def test_rise_upsample_formula_equals_scipy_zoom():
    rng = np.random.RandomState(3)
    grid, shifts, cell = orise.draw_grid_and_shifts((224, 224), 5, 8, 0.5, rng)
    assert cell.tolist() == [28, 28] and shifts.min() >= 0 and shifts.max() < 28
    for gi in grid:
        a = orise.upsample_grid(gi, 9 * cell)
        b = orise.upsample_grid_formula(gi, 9 * cell)
        assert a.shape == (252, 252)
        assert np.abs(a - b).max() <= 1e-6
    m = orise.masks_from(grid, shifts, (224, 224), cell)
    assert m.shape == (5, 1, 224, 224) and m.min() >= 0 and m.max() <= 1
    # odd geometry: s=7 on 30x45
    grid, shifts, cell = orise.draw_grid_and_shifts((30, 45), 3, 7, 0.5, rng)
    for gi in grid:
        assert np.abs(orise.upsample_grid(gi, 8 * cell) - orise.upsample_grid_formula(gi, 8 * cell)).max() <= 1e-6


def test_vit_mini_pixel_ig_and_attention_ig():
    """config 4: the reference's hooked ViT (rebuilt from its state dict) through the oracle."""
    from helpers import vit_mini_from
    from oracle import vit_attr
    g = load_golden("vit_mini.npz")
    model = vit_mini_from(g)
    assert rel_inf(oig.ig(g["x"], model, 50, 25, 1, 0, int(g["target"])), g["ig"]) <= 1e-5
    assert rel_inf(vit_attr.attention_ig(model, g["x"], int(g["target"]), 20), g["attn_ig"]) <= 1e-5


def test_vit_mini_inflow_rollout():
    """vit_inflow.npz: the reference's compute_RAVE / generate_rollout(InFlow=True) on the mini ViT, against the oracle's NumPy
    restatement driven through the build's hooked ViT (same state dict, same residual-stream accessors)."""
    from helpers import vit_mini_from
    from oracle import vit_attr
    g, gi = load_golden("vit_mini.npz"), load_golden("vit_inflow.npz")
    roll, mats = vit_attr.inflow_rollout(vit_mini_from(g), g["x"])
    assert rel_inf(roll, gi["inflow_rollout"]) <= 1e-5 and rel_inf(mats, gi["inflow_matrices"]) <= 1e-5


def test_gradcam_reduce_matches_reference_owned_cam_code():
    """cam.npz comes from ViT_CX/get_feature_map.get_cam_weights + ViT_CX/base_cam.get_cam_image."""
    g = load_golden("cam.npz")
    for tag in "abc":
        act, grad = g[f"{tag}_act"], g[f"{tag}_grad"]
        scale = np.abs(g[f"{tag}_cam"]).max()
        assert np.abs(ogc.cam_reduce(act, grad, relu=False) - g[f"{tag}_cam"]).max() / scale <= TOL
        assert np.abs(ogc.cam_reduce(act, grad, relu=True) - g[f"{tag}_cam_relu"]).max() / scale <= TOL
        assert rel_inf(grad.mean(axis=(2, 3), dtype=np.float32), g[f"{tag}_weights"]) <= TOL


# ------------------------------------------------------------------ ViT-CX / TIS (f4 maskers)
def test_vitcx_pieces_match_reference_functions():
    from oracle import vit_cx as ocx
    g = load_golden("vit_cx.npz")
    assert np.array_equal(ocx.reshape_function_vit(g["tokens"]), g["tokens_reshaped"])
    assert rel_inf(ocx.norm_matrix(g["act"]), g["act_norm"]) <= TOL
    assert rel_inf(ocx.cos_similar_matrix(g["cos_in"], g["cos_in"]), g["cos"]) <= TOL
    assert (g["cos"][3] == 0).all() and (g["cos"][:, 3] == 0).all()           # the zero row: NaN -> 0
    model = tiny_from(g)
    soft = lambda b: torch.softmax(model(torch.from_numpy(b)), 1).numpy()      # noqa: E731
    sal = ocx.causal_score(soft, g["x"][0], g["masks"], g["class_p"], g["noise"], gpu_batch=4)
    assert sal.shape == g["sal"].shape and rel_inf(sal, g["sal"]) <= TOL


def test_vitcx_cluster_sums_and_members():
    from oracle import vit_cx as ocx
    rng = np.random.default_rng(7)
    base = rng.random((4, 16, 16)).astype(np.float32)
    fmap = np.concatenate([base + 0.01 * rng.random((4, 16, 16)).astype(np.float32) for _ in range(3)])   # 3 near-copies of 4 maps
    masks, labels, rows = ocx.masks_from_feature_maps(fmap, 32, 32, distance_threshold=0.1)
    assert len(set(labels)) == 4 and masks.shape == (4, 1024)
    for j in range(4):
        assert len({labels[j], labels[j + 4], labels[j + 8]}) == 1
    assert masks.min() == 0 and masks.max() == 1


def test_tis_stages_match_reference():
    from oracle import tis as otis
    from helpers import vit_mini_from
    g, gv = load_golden("tis.npz"), load_golden("vit_mini.npz")
    model = vit_mini_from(gv)
    pred, acts = otis.encoder_activations(model, gv["x"])
    assert pred == int(g["a_pred"]) and rel_inf(acts, g["a_acts"]) <= 5e-6
    for tag, ratio, bs in (("a", 0.5, 3), ("b", [0.25, 0.75], 4)):
        masks, idx = otis.binary_masks(g[f"{tag}_raw"], ratio)
        assert np.array_equal(masks, g[f"{tag}_masks"])
        sc = otis.scores(model, gv["x"], pred, idx, bs)
        assert rel_inf(sc, g[f"{tag}_scores"]) <= 5e-6
        assert rel_inf(otis.saliency(g[f"{tag}_scores"], masks, 4, 4, False), g[f"{tag}_sal"]) <= TOL
        assert rel_inf(otis.saliency(g[f"{tag}_scores"], masks, 4, 4, True), g[f"{tag}_sal_norm"]) <= 1e-5
        if tag == "a":
            assert np.array_equal(np.stack(idx), g["a_idx"])
            assert np.array_equal(otis.mask_input_zero(gv["x"], idx[:3], 8), g["a_masked_zero"])


def test_kmeans_restatement_recovers_separated_clusters():
    from oracle import tis as otis
    rng = np.random.RandomState(3)
    centres = rng.randn(5, 16).astype(np.float32) * 10
    pts = np.concatenate([c + 0.01 * rng.randn(40, 16).astype(np.float32) for c in centres])
    got = otis.kmeans_centroids(pts, 5, rng=np.random.RandomState(11))
    d = np.abs(got[:, None] - centres[None]).max(-1)            # every true centre is found by some centroid or merged
    assert (d.min(0) < 0.1).sum() >= 3


def test_config1_gradcam_resnet50_on_the_cpu_reference_path():
    """BASELINE.json configs[0]: "Grad-CAM on ResNet-50, single 224x224 image, CPU reference path (plumbing, no GPU)".
    The oracle's Grad-CAM (captum's published arithmetic, cross-checked against the reference-owned CAM code by cam.npz) on
    the full-size ResNet-50 with seeded random weights, on the host: shapes, the ReLU, the channel-weight identity
    cam = sum_c mean_hw(grad_c) * act_c, and the 3-channel |.| map the harness turns it into (evaluatePerturbation.py:147-153,181)."""
    from xai_engine.zoo import resnet50
    torch.set_num_threads(4)
    model = resnet50(seed=0)
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        t = int(model(x).argmax(1)[0])
    act, grad = ogc.layer_act_and_grad(model, model.layer4, x, t)
    assert act.shape == grad.shape == (1, 2048, 7, 7)
    cam = ogc.cam_reduce(act, grad, relu=True)
    raw = ogc.cam_reduce(act, grad, relu=False)
    assert cam.shape == (1, 7, 7) and (cam >= 0).all() and np.array_equal(cam, np.maximum(raw, 0))
    w = grad.astype(np.float64).mean(axis=(2, 3), keepdims=True)
    assert rel_inf(raw, (w * act.astype(np.float64)).sum(1)) <= TOL
    sal = ogc.gradcam_saliency(act, grad, 224, 224)
    assert sal.shape == (1, 224, 224) and sal.dtype == np.float32 and (sal >= 0).all()
    up = torch.nn.functional.interpolate(torch.from_numpy(cam)[None], size=(224, 224), mode="bilinear", align_corners=False)[0].numpy()
    assert rel_inf(sal, 3 * up) <= TOL                                         # |cam_up + cam_up + cam_up|


def test_tied_map_with_the_reference_own_order():
    """tests/golden/perturb_ties.npz: a ReLU'd, quantised map (9 distinct values in 1024 pixels).  The reference's default
    np.argsort is unstable, so its pixel order on ties is whatever its sort produced on the machine that ran it (1014 of 1024
    positions differ from the stable order); given THAT order (`order=`), the oracle reproduces every perturbed image byte for
    byte and every return tuple -- the tie rule is the only divergence."""
    g = load_golden("perturb_ties.npz")
    assert int(g["n_positions_differing_from_the_stable_order"]) > 900
    model = tiny_from(g)
    fn = logits_fn_of(model)
    x, sal = g["x"], g["saliency"]
    HW = x.shape[-1] * x.shape[-2]
    blur, zeros = _blur_fn(g), np.zeros_like
    differs = 0
    for tag, func, mode, uses_blur in CASES:
        descending = (mode != "lerf") if func != "pnp" else (mode == "morf")
        order = g["salient_order_desc"][0] if descending else g["salient_order_asc"][0]
        plan = op.Plan(HW, int(g["step"]), int(g["max_bs"]), None, always_leftover=tag.startswith("MONO"))
        inserting = mode in ("ins", "positive")
        start, finish = (g["substrate_blur"], x) if inserting else (x, np.zeros_like(x))
        groups, _ = op.flip_groups(sal, HW, plan, None, descending, order=order)
        assert [_sha(i) for i in op.sequence(start, finish, groups)] == list(g[f"{tag}_img_sha"]), tag
        res = getattr(op, func)(fn, x, sal, mode, int(g["step"]), blur if uses_blur else zeros, None, int(g["max_bs"]), order=order)
        stable = getattr(op, func)(fn, x, sal, mode, int(g["step"]), blur if uses_blur else zeros, None, int(g["max_bs"]))
        for i, r in enumerate(res):
            want = g[f"{tag}_ret{i}"]
            if np.ndim(want) == 0 and float(want) == int(want) and func != "mono":
                assert int(r) == int(want), (tag, i)
            else:
                assert rel_inf(r, want) <= TOL, (tag, i, rel_inf(r, want))
                differs += rel_inf(stable[i], want) > 1e-3
    assert differs >= 5                       # with the stable tie rule the curves of a tied map are genuinely different curves
