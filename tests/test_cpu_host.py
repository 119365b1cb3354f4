"""CPU-only tests (-m "not gpu"): the C-ABI library loads and exports exactly what
include/xai_hip.h declares (no compute call is made -- there is no GPU here), the host-side
curve arithmetic reproduces the reference's golden tuples, the product refuses to run without a
HIP device, and the multi-rank paths work over gloo with world_size 2.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, PKG, load_golden


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

HEADER = os.path.join(ROOT, "include", "xai_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xai_[a-z0-9_]+)\s*\(", src)))


def _lib_path():
    from xai_engine import LIB_PATH
    if not os.path.exists(LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(PKG, "csrc"), "-j8"], check=True)
    return LIB_PATH


def test_library_exports_every_declared_symbol_and_nothing_else():
    lib = _lib_path()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], check=True, capture_output=True, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("xai_"))
    assert exported == _declared()


def test_ctypes_table_matches_header_and_library_loads():
    from xai_engine import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    hdr = open(HEADER).read()
    assert lib.xai_version() == _lib.ABI_VERSION == int(re.search(r"#define XAI_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.xai_version_minor() == _lib.ABI_MINOR == int(re.search(r"#define XAI_ABI_MINOR (\d+)", hdr).group(1))
    assert b"NULL" in lib.xai_strerror(-1) and lib.xai_strerror(0) == b"success"
    assert lib.xai_rank_workspace_bytes(2, 50176) == 2 * (4 * 50176 + 5 * 256 * 49 + 8) * 4      # keys+idx ping-pong, 4 hists + offs, flags
    # argument validation happens before any HIP call, so it is checkable without a GPU
    assert lib.xai_ig_interp_f32(None, None, 0.0, None, 0, 1, 1, 4, None, None) == -1
    assert lib.xai_rank_f32(None, 1, 4, None, None, None, 0, None) == -1


def test_every_header_entry_cites_the_reference_line_it_replaces():
    src = open(HEADER).read()
    for name in _declared():
        if name in ("xai_version", "xai_version_minor", "xai_strerror", "xai_rank_workspace_bytes", "xai_gradcam_workspace_bytes", "xai_flip_steps_i32", "xai_ig_finish_f32",
                    "xai_idgi_accum_f32"):
            continue
        at = src.index(name + "(")
        comment = src[src.rfind("/*", 0, at):at]
        assert re.search(r"\.py:\d+", comment), f"{name}: no reference file:line in its header comment"


def test_product_refuses_cpu_devices_and_never_imports_the_oracle():
    from xai_engine import XaiHipError
    from util.attribution_methods import saliencyMethods as attr
    from util.test_methods import MASTestFunctions as MAS
    with pytest.raises(XaiHipError):
        attr.IG(torch.zeros(1, 3, 8, 8), torch.nn.Identity(), 10, 5, 1, 0, "cpu", 0)
    with pytest.raises(XaiHipError):
        MAS.MASMetric(torch.nn.Identity(), 64, "del", 8, torch.zeros_like).single_run(torch.zeros(1, 3, 8, 8), np.zeros((8, 8), np.float32), "cpu")
    from util.attribution_methods.ViT_CX.ViT_CX import ViT_CX
    from util.attribution_methods.ViT_CX.causal_score import causal_score
    from util.attribution_methods.TIS import TIS
    ident = torch.nn.Identity()
    with pytest.raises(XaiHipError):
        ViT_CX(ident, torch.zeros(1, 3, 8, 8), ident, device="cpu")
    with pytest.raises(XaiHipError):
        causal_score(ident, (8, 8), device="cpu")(torch.zeros(1, 3, 8, 8), torch.zeros(2, 8, 8), 0.5)
    with pytest.raises(XaiHipError):
        TIS(ident)(torch.zeros(1, 3, 8, 8))
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f"{f} imports the oracle"


# ------------------------------------------------------------------------------ host curve arithmetic vs golden
@pytest.mark.parametrize("fixture", ["perturb_small.npz", "perturb_224.npz", "perturb_patch.npz"])
def test_curve_arithmetic_reproduces_reference_tuples(fixture):
    from xai_engine import curves
    g = load_golden(fixture)
    HW = g["x"].shape[-1] * g["x"].shape[-2]
    pm = g["patch_mask"] if "patch_mask" in g else None
    for tag, always in (("MAS_ins", False), ("MONO_negative", True)):
        n_steps, step, batches = curves.step_plan(HW, int(g["step"]), int(g["max_bs"]), pm, always)
        np.testing.assert_array_equal(np.array(batches), g[f"{tag}_batch_sizes"])
    for mode in ("ins", "del", "lerf", "morf"):
        corrected, dens, norm = g[f"MAS_{mode}_ret1"], g[f"MAS_{mode}_ret3"], g[f"MAS_{mode}_ret4"]
        np.testing.assert_array_equal(curves.mas_correct(norm, dens, mode), corrected)
        # density from NumPy segment sums in the reference's pixel order
        sal = g["saliency"].reshape(-1)
        n_steps = len(dens) - 1
        if pm is None:
            o = np.argsort(sal, kind="stable")
            o = o if mode == "lerf" else o[::-1]
            s = int(g["step"])
            seg = np.array([np.sum(sal[o[i * s:(i + 1) * s]]) for i in range(n_steps)], dtype=np.float32)
            total = np.sum(sal.reshape(1, 1, HW))
        else:
            flip, _ = curves.patch_flip_steps(sal, pm, HW, n_steps, descending=(mode != "lerf"))
            seg, total = curves.patch_density_sums(sal, flip, HW, n_steps)
        np.testing.assert_allclose(curves.density_curve(seg, total, mode == "ins"), dens, rtol=0, atol=1e-6)
    # monotone normalisation is idempotent on its own output and bounded
    norm = g["MAS_del_ret4"]
    assert (np.diff(norm) <= 0).all() and norm.min() >= 0 and norm.max() <= 1
    from oracle import perturb as op
    r = np.random.default_rng(0).random(40)
    np.testing.assert_array_equal(curves.monotone_normalise(r, 0.1, 0.9, True), op.monotone(r, 0.1, 0.9, True))
    np.testing.assert_array_equal(curves.monotone_normalise(r, 0.1, 0.9, False), op.monotone(r, 0.1, 0.9, False))


def test_patch_flip_steps_match_oracle_groups():
    from xai_engine import curves
    from oracle import perturb as op
    g = load_golden("perturb_patch.npz")
    HW = 32 * 32
    for desc in (True, False):
        plan = op.Plan(HW, 32, 50, g["patch_mask"])
        groups, _ = op.flip_groups(g["saliency"], HW, plan, g["patch_mask"], desc)
        flip, _ = curves.patch_flip_steps(g["saliency"], torch.from_numpy(g["patch_mask"]), HW, plan.n_steps, desc)
        for t, grp in enumerate(groups):
            np.testing.assert_array_equal(np.nonzero(flip == t)[0], grp)


def test_auc_gkern_alpha_parameters_on_host():
    from util.test_methods import MASTestFunctions as MAS
    from util.attribution_methods import saliencyMethods as attr
    g = load_golden("kern.npz")
    np.testing.assert_array_equal(MAS.gkern(31, 31).numpy(), g["gkern_31_31"])
    np.testing.assert_array_equal(MAS.gkern(11, 5).numpy(), g["gkern_11_5"])
    for c, v in zip(g["auc_curves"], g["auc_values"]):
        assert MAS.auc(c) == v
    gi = load_golden("ig_small.npz")
    al, sub = attr.getAlphaParameters(torch.from_numpy(gi["slopes"]), 50, float(gi["slope_step"]))
    np.testing.assert_array_equal(al.numpy(), gi["idg_alphas"])
    np.testing.assert_array_equal(sub.numpy(), gi["idg_substep"])
    from xai_engine.blur import gkern1d
    v = gkern1d(31, 31).double().numpy()
    assert np.abs(np.outer(v, v) - g["gkern_31_31"][0, 0]).max() <= 1e-9


def test_rise_mask_draw_is_the_reference_rng_stream():
    from xai_engine.rise import draw_masks
    from oracle import rise as orise
    np.random.seed(3)
    a = draw_masks((224, 224), 20, 8, 0.5)
    np.random.seed(3)
    b = orise.draw_grid_and_shifts((224, 224), 20, 8, 0.5)
    np.testing.assert_array_equal(a[0], b[0].astype(np.uint8))
    np.testing.assert_array_equal(a[1], b[1])
    assert a[2].tolist() == [28, 28]


# ------------------------------------------------------------------------------ multi-rank over gloo
WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import numpy as np, torch, torch.distributed as dist
from xai_engine import dist as xd, sweep
rank, world, device = xd.init_from_env("gloo")
assert world == 2 and device.type == "cpu"
# (1) image sharding + the 11-element all-reduce
owned = sweep.shard_indices(7, rank, world)
local = {k: float(sum((i + 1) * (j + 1) for i in owned)) for j, k in enumerate(sweep.KEYS)}
total, used = sweep.reduce_counters(local, len(owned))
assert used == 7
for j, k in enumerate(sweep.KEYS):
    assert total[k] == sum((i + 1) * (j + 1) for i in range(7)), (k, total[k])
# (2) RISE mask ranges tile [0, N) and every rank adopts rank 0's draw
lo, hi = xd.mask_range(8001, rank, world)
sizes = torch.tensor([hi - lo]); dist.all_reduce(sizes); assert int(sizes) == 8001
rng = np.random.RandomState(100 + rank)
masks = ((rng.rand(6, 8, 8) < 0.5).astype(np.uint8), rng.randint(0, 28, (6, 2)).astype(np.int32), np.array([28 + rank, 28]))
calls = []
real = dist.broadcast
dist.broadcast = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
g, s, c = xd.broadcast_masks(masks, device)
dist.broadcast = real
assert len(calls) == 1                                          # ONE packed broadcast, not one per array
r0 = np.random.RandomState(100)
np.testing.assert_array_equal(g, (r0.rand(6, 8, 8) < 0.5).astype(np.uint8))
np.testing.assert_array_equal(s, r0.randint(0, 28, (6, 2)).astype(np.int32))
assert g.dtype == np.uint8 and s.dtype == np.int32 and c.tolist() == [28, 28]
# (2b) a draw held as torch tensors (rise.draw_masks_on_device's kind) goes through the same ONE broadcast and comes back as tensors
tm = (torch.from_numpy(masks[0].copy()), torch.from_numpy(masks[1].copy()), masks[2])
calls.clear()
dist.broadcast = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
tg, ts, tc = xd.broadcast_masks(tm, device)
dist.broadcast = real
assert len(calls) == 1 and torch.is_tensor(tg) and torch.is_tensor(ts) and tg.dtype == torch.uint8 and ts.dtype == torch.int32
np.testing.assert_array_equal(tg.numpy(), g); np.testing.assert_array_equal(ts.numpy(), s); assert tc.tolist() == [28, 28]
# (3) partial-map all-reduce
part = torch.full((4, 4), float(rank + 1), dtype=torch.float64)
assert float(xd.all_reduce_sum(part)[0, 0]) == 3.0
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_sharding_and_reduction(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(script), PKG, ROOT], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def test_harness_image_loading_and_name_parsing(tmp_path):
    from PIL import Image
    from xai_engine import harness
    assert harness.image_number("ILSVRC2012_val_00000123.JPEG") == 122
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, (300, 400, 3), dtype=np.uint8)          # landscape: shorter side 300 -> 224
    Image.fromarray(arr).save(tmp_path / "a.png")
    t = harness.load_image(str(tmp_path / "a.png"), 224)
    assert t.shape == (3, 224, 224) and t.dtype == torch.float32 and 0 <= float(t.min()) and float(t.max()) <= 1
    want = Image.fromarray(arr).resize((int(224 * 400 / 300), 224), Image.BILINEAR)
    left = int(round((want.size[0] - 224) / 2.0))
    want = np.asarray(want.crop((left, 0, left + 224, 224)), dtype=np.float32).transpose(2, 0, 1) / 255
    np.testing.assert_array_equal(t.numpy(), want)
    Image.fromarray(arr[:, :, 0]).save(tmp_path / "g.png")             # grey-scale -> 1 channel -> rejected by callers
    assert harness.load_image(str(tmp_path / "g.png"), 224).shape == (1, 224, 224)
    n = harness.normalize(t, harness.CNN_MEAN, harness.CNN_STD)
    assert abs(float(n[0, 0, 0]) - (float(t[0, 0, 0]) - 0.485) / 0.229) < 1e-6


def test_sweep_state_checkpoint_roundtrip(tmp_path):
    from xai_engine.sweep import SweepState, KEYS
    prefix = str(tmp_path / "ckpt")
    st = SweepState(10, 1, 4)
    st.sums["MAS_ins"], st.sums["MONO_neg"], st.used, st.next_pos, st.attr_time = 1.25, -0.5, 2, 2, 3.5
    st.save(prefix)
    again = SweepState.load_or_new(prefix, 10, 1, 4)
    assert again.sums == st.sums and (again.used, again.next_pos, again.attr_time) == (2, 2, 3.5)
    assert set(again.sums) == set(KEYS)
    fresh = SweepState.load_or_new(prefix, 10, 1, 8)                 # another world size: another file, nothing to resume
    assert fresh.used == 0 and fresh.next_pos == 0
    assert SweepState.load_or_new(None, 10, 0, 1).used == 0
    # the same file under another split or another identity (method, model, weights, flow, image list) is refused loudly
    from xai_engine.sweep import CheckpointMismatch, sweep_identity, model_fingerprint
    with pytest.raises(CheckpointMismatch):
        SweepState.load_or_new(prefix, 11, 1, 4)
    ida = sweep_identity(attr_func="ig", model_name="R50", files="a|b|c", fused=True)
    idb = sweep_identity(attr_func="gc", model_name="R50", files="a|b|c", fused=True)
    assert ida != idb and ida == sweep_identity(model_name="R50", fused=True, files="a|b|c", attr_func="ig")
    assert sweep_identity(files="x" * 500) != sweep_identity(files="x" * 499 + "y") and len(sweep_identity(files="x" * 500)) < 60
    st2 = SweepState(10, 0, 1, ida)
    st2.used = st2.next_pos = 10
    st2.save(prefix)
    assert SweepState.load_or_new(prefix, 10, 0, 1, ida).used == 10
    with pytest.raises(CheckpointMismatch, match="different sweep"):
        SweepState.load_or_new(prefix, 10, 0, 1, idb)
    m1, m2 = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)
    assert model_fingerprint(m1) == model_fingerprint(m1) != model_fingerprint(m2)


def test_vitcx_and_tis_host_logic_on_reference_vectors():
    """The device-independent pieces of the two masker drivers: cluster member lists, token reshape, TIS's binary masks."""
    from xai_engine.vit_cx import cluster_members, reshape_function_vit
    from xai_engine.tis import TIS
    members, offs = cluster_members([2, 0, 1, 0, 2, 2, 1])
    assert members.tolist() == [1, 3, 2, 6, 0, 4, 5] and offs.tolist() == [0, 2, 4, 7]
    assert members.dtype == np.int32 and offs.dtype == np.int32
    g = load_golden("vit_cx.npz")
    assert np.array_equal(reshape_function_vit(torch.from_numpy(g["tokens"])).numpy(), g["tokens_reshaped"])
    t = load_golden("tis.npz")
    for tag, ratio in (("a", 0.5), ("b", [0.25, 0.75])):
        masks, idx = TIS(None, n_masks=8, tokens_ratio=ratio).generate_binary_masks(torch.from_numpy(t[f"{tag}_raw"]))
        assert np.array_equal(masks.numpy(), t[f"{tag}_masks"])
        if tag == "a":
            assert np.array_equal(idx[0].numpy(), t["a_idx"])


def test_fuse_bn_relu_is_a_no_op_off_the_gpu():
    """On CPU tensors the fused blocks fall back to the PyTorch modules: identical results, same parameter names."""
    from xai_engine.prepare import fuse_bn_relu
    from xai_engine.zoo import resnet50
    model = resnet50(seed=0, width=8, num_classes=10)
    fused = fuse_bn_relu(model)
    assert list(fused.state_dict()) == list(model.state_dict())
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        assert torch.equal(fused(x), model(x))
    with pytest.raises(ValueError):
        fuse_bn_relu(torch.nn.Linear(3, 3))


def test_tolerances_are_tied_to_measured_errors():
    """profiles/r02_parity.json is the ledger conftest.check wrote on an MI355X (tests/parity_report.sh): every comparison of
    the GPU suite with its measured error and the tolerance asserted.  No tolerance may exceed max(the 1e-5 bar, 2 x the largest error measured in its test family):
    a tolerance is a measurement with head-room, not slack."""
    import json
    from conftest import BAR
    led = json.load(open(os.path.join(ROOT, "profiles", "r02_parity.json")))
    assert led["meta"]["exitstatus"] == 0 and led["meta"]["deterministic"] is True
    rows = led["comparisons"]
    assert len(rows) >= 300
    # a tolerance above the bar is shared by the cases of one test family (first path component of the name): the family's
    # largest measured error must reach at least half of it
    fam = {}
    for r in rows:
        if r["tol"] > BAR * (1 + 1e-9):
            key = (r["name"].split("/")[0], r["against"], r["tol"])
            fam[key] = max(fam.get(key, 0.0), r["measured"])
    loose = {k: v for k, v in fam.items() if k[2] > 2 * v * (1 + 1e-9)}
    assert not loose, loose
    golden = [r for r in rows if r["against"] == "golden"]
    assert len(golden) >= 150 and all(r["measured"] <= r["tol"] for r in rows)
    # what is above the bar is exactly what DESIGN.md section 2 explains: one ReLU gate of ig_224 and the finite-difference slopes
    above = sorted({r["name"] for r in golden if r["measured"] > BAR})
    assert above == ["IG/ig_224.npz/ig_tensor_baseline/under_ill_conditioned_gates", "getSlopes/ig_small"], above


def test_model_zoo_matches_the_architectures_the_reference_harness_names():
    """evaluatePerturbation.py:627-659 instantiates torchvision's resnet101 / resnext101_64x4d and timm-layout ViT-B/16, /32; the
    build's definitions must take their state dicts as they are: same parameter counts (torchvision / timm's published numbers),
    same number of state-dict entries, same key names."""
    from xai_engine import zoo
    from xai_engine.evaluate_perturbation import MODELS
    want = {"resnet50": (25557032, 320), "resnet101": (44549160, 626), "resnet152": (60192808, 932), "resnext101_64x4d": (83455272, 626),
            "vit_base_patch16_224": (86567656, 152), "vit_base_patch32_224": (88224232, 152)}
    for name, (n_params, n_keys) in want.items():
        m = getattr(zoo, name)()
        sd = m.state_dict()
        assert sum(p.numel() for p in m.parameters()) == n_params, name
        assert len(sd) == n_keys, (name, len(sd))
    sd = zoo.resnext101_64x4d().state_dict()
    assert tuple(sd["layer1.0.conv2.weight"].shape) == (256, 4, 3, 3)                 # 64 groups x 4 channels
    for k in ("conv1.weight", "bn1.running_var", "layer3.22.bn3.num_batches_tracked", "layer4.0.downsample.1.weight", "fc.bias"):
        assert k in sd, k
    sd = zoo.vit_base_patch32_224().state_dict()
    assert tuple(sd["patch_embed.proj.weight"].shape) == (768, 3, 32, 32) and tuple(sd["pos_embed"].shape) == (1, 50, 768)
    for k in ("cls_token", "blocks.11.attn.qkv.bias", "blocks.0.mlp.fc2.weight", "norm.weight", "head.bias"):
        assert k in sd, k
    # the CLI's table: the reference's names, batch sizes and patch counts
    assert {k: (v[1], v[3]) for k, v in MODELS.items()} == {"R50": (50, 0), "R101": (50, 0), "R152": (50, 0), "RNXT": (25, 0),
                                                             "VIT16": (25, 14), "VIT32": (50, 7)}


def _build_abi_host(out_path):
    """gcc -std=c99 examples/abi_host.c against include/xai_hip.h and the in-tree library -> executable path"""
    lib_dir = os.path.dirname(_lib_path())
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "abi_host.c"), "-L" + lib_dir, "-lxai_hip", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib", "-o", str(out_path)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(out_path)


def test_the_header_is_plain_c_and_a_c_host_links_against_the_library(tmp_path):
    """The drop-in boundary is a C ABI: include/xai_hip.h compiles as C99 (no C++, no torch types) and a plain-C host
    (examples/abi_host.c) links against libxai_hip.so.  Running it needs a GPU: tests/test_gpu_kernels.py does."""
    exe = _build_abi_host(tmp_path / "abi_host")
    out = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True, check=True).stdout
    used = sorted({l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("xai_")})
    assert used == ["xai_flip_steps_i32", "xai_ig_accum_f32", "xai_perturb_batch_f32", "xai_rank_f32", "xai_rank_workspace_bytes", "xai_strerror", "xai_version", "xai_version_minor"]


def test_class_quota_replays_the_reference_loop_order():
    """harness.ClassQuota (the order-dependent tail of the selection pre-pass) against the reference loop written out
    (evaluatePerturbation.py:520-576: stop at image_count, RGB filter, sanity filter, ceil(count / classes) per class)."""
    from xai_engine.harness import ClassQuota
    rng = np.random.default_rng(4)
    for count, n_cls in ((7, 3), (10, 10), (5, 1000), (1000, 1000), (12, 5)):
        verdicts = [(f"f{i:04d}", bool(rng.random() < 0.9), bool(rng.random() < 0.6), int(rng.integers(0, n_cls))) for i in range(400)]
        q = ClassQuota(count, n_cls)
        for v in verdicts:
            q.offer(*v)
        per_class = int(np.ceil(count / n_cls))
        used, want = [0] * n_cls, []
        for name, rgb, sane, t in verdicts:
            if len(want) == count:
                break
            if not rgb:
                continue
            if not sane:
                continue
            if used[t] == per_class:
                continue
            used[t] += 1
            want.append((name, t))
        assert q.chosen == want and q.full == (len(want) == count)
