#!/usr/bin/env python3
"""Evidence behind the end-to-end parity tolerances.  Test infrastructure: may use oracle/.

    python tests/parity_report.py gates   --out profiles/r02_gate_flips.json
    python tests/parity_report.py resnet  --mode {immediate,deterministic,finddb} --out profiles/r02_resnet_<mode>.json

gates   For the IG fixtures (tests/golden/ig_small.npz, ig_224.npz -- outputs of the reference on a CPU): run the tiny
        ReLU network's convolution on the host (fp32 = what the reference's oneDNN computes, fp64 = ground truth) and on
        the GPU (MIOpen) at all 50 path points, list the ReLU gates on which the two fp32 convolutions disagree together
        with the fp64 pre-activation there, split the error of the HIP IG map against the golden map into pixels inside /
        outside the 3x3 footprints of those gates, and recompute the GPU map with the HOST's gates forced (everything
        else -- transposed convolution, K2 accumulation -- on the device) to show what remains.
resnet  ResNet-50 (seeded random weights), one 224x224 image: is a classifier pass bit-reproducible run to run (forward
        of the 50-image perturbation batch, forward+backward of 50 interpolants; first module whose output / gradient
        differs), and do the fused 3-sequence sweep and the reference's 8-run flow give the same response curves and AUCs
        -- with the raw-probability difference separated from the 1/|orig - base| amplification of the normalised curves.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "image-classification-xai_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

DEV = "cuda:0"


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


# ------------------------------------------------------------------------------------------- gates
def tiny_manual_grads(w_conv, w_fc, target, gate, HW):
    """d logit[target] / d input of the tiny net for GIVEN ReLU gates (S,8,H,W bool), on gate.device:
    fc row -> adaptive-avg-pool(4) backward (each cell spreads w/(area)) -> gate -> transposed 3x3 convolution."""
    H, W = HW
    up = w_fc[target].reshape(1, 8, 4, 4)
    up = up.repeat_interleave(H // 4, 2).repeat_interleave(W // 4, 3) / float((H // 4) * (W // 4))
    return F.conv_transpose2d(up * gate.to(up.dtype), w_conv, padding=1)


def gates_report(out_path):
    from conftest import load_golden
    from helpers import tiny_from
    from util.attribution_methods import saliencyMethods as attr
    from xai_engine import kernels as K
    torch.backends.cudnn.benchmark = False
    report = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "fixtures": {}}
    for fx in ("ig_small.npz", "ig_224.npz"):
        g = load_golden(fx)
        cpu, gpu = tiny_from(g, "cpu"), tiny_from(g, DEV)
        x = torch.from_numpy(g["x"])
        t = int(g["target"])
        H, W = x.shape[2:]
        cases = {"ig": (50, 25, 1, 0), "lig": (50, 25, .9, 0), "ig_tensor_baseline": (50, 50, 1, torch.from_numpy(g["baseline_tensor"])),
                 "lig_a05_b025": (50, 10, .5, .25)}
        fx_rep = {}
        for key, (steps, bs, a_star, base) in cases.items():
            base_t = base if torch.is_tensor(base) else torch.full_like(x, float(base))
            alphas = torch.linspace(0, 1, steps)
            pts = base_t + alphas.reshape(-1, 1, 1, 1) * (x - base_t)                 # (S,3,H,W), the reference's expression (:44)
            with torch.no_grad():
                z_cpu = cpu.conv(pts)                                                # oneDNN fp32 -- what the reference computed
                z_gpu = gpu.conv(pts.to(DEV)).cpu()                                  # MIOpen fp32
                z64 = F.conv2d(pts.double(), cpu.conv.weight.double(), cpu.conv.bias.double(), padding=1)
            flip = (z_cpu > 0) != (z_gpu > 0)
            idx = flip.nonzero().tolist()
            flips = [{"step": s, "ch": c, "y": yy, "x": xx, "z_cpu": float(z_cpu[s, c, yy, xx]), "z_gpu": float(z_gpu[s, c, yy, xx]),
                      "z_fp64": float(z64[s, c, yy, xx])} for s, c, yy, xx in idx]
            conv_noise = float((z_cpu - z_gpu).abs().max())                          # the rounding-sized disagreement of the two convs
            foot = F.max_pool2d(flip.any(1, keepdim=True).any(0, keepdim=True).float(), 3, 1, 1)[0, 0].bool().numpy()   # (H,W) 3x3 footprints

            # per-step gradients on both devices
            pc = pts.clone().requires_grad_(True)
            (g_cpu,) = torch.autograd.grad(cpu(pc)[:, t].sum(), pc)
            pg = pts.to(DEV).requires_grad_(True)
            (g_gpu,) = torch.autograd.grad(gpu(pg)[:, t].sum(), pg)
            g_gpu = g_gpu.cpu()
            d = (g_gpu - g_cpu).abs() / g_cpu.abs().max()
            big = (d > 1e-5)
            big_pix = big.any(1).any(0).numpy()                                       # (H,W): some step/channel differs by > 1e-5
            per_step_foot = F.max_pool2d(flip.any(1, keepdim=True).float(), 3, 1, 1)[:, 0].bool()     # (S,H,W)
            big_outside_own_step = int((big.any(1) & ~per_step_foot).sum())

            got = attr.IG(x.clone(), gpu, steps, bs, a_star, base, DEV, torch.tensor(t)).cpu().numpy()
            gold = g[key]
            den = np.abs(gold).max()
            err_map = np.abs(got - gold).max(0) / den                                  # (H,W)
            # GPU map with the HOST's gates forced; Left-IG: the golden run's own cutoff (from its logits when recorded)
            with torch.no_grad():
                grads_forced = tiny_manual_grads(gpu.conv.weight, gpu.fc.weight, t, (z_cpu > 0).to(DEV), (H, W))
            n_use = None
            if a_star != 1:
                from oracle import ig as oig
                lg = g[key + "_logits"] if key + "_logits" in g else g["logits"]
                n_use = torch.tensor([oig.left_cutoff(lg, a_star)], dtype=torch.int32, device=DEV)
            base_dev = base.to(DEV) if torch.is_tensor(base) else float(base)
            forced = K.ig_accum(grads_forced.reshape(1, steps, 3, H, W).contiguous(), x.to(DEV), base_dev, n_use=n_use)[0].cpu().numpy()
            from oracle import ig as oig_
            host = oig_.ig(g["x"], cpu, steps, bs, a_star, base.numpy() if torch.is_tensor(base) else base, t)
            fx_rep[key] = {
                "oracle_with_cpu_classifier_on_this_host_vs_golden_rel_inf": rel(host, gold),
                "n_gates": int(flip.numel()), "n_flipped_gates": len(flips), "flipped_gates": flips[:40],
                "max_abs_fp64_preactivation_at_flipped_gates": max([abs(f["z_fp64"]) for f in flips], default=0.0),
                "max_abs_conv_disagreement_cpu_vs_gpu": conv_noise,
                "n_gates_within_conv_noise_of_zero_fp64": int((z64.abs() <= conv_noise).sum()),
                "per_step_gradient": {"n_pixels_differing_gt_1e-5": int(big_pix.sum()),
                                      "of_which_outside_flipped_gate_footprints_of_their_step": big_outside_own_step,
                                      "max_rel_diff_outside_footprints": float(d.max(1).values[~per_step_foot].max())
                                      if (~per_step_foot).any() else 0.0},
                "ig_map_vs_golden": {"rel_inf_all_pixels": float(err_map.max()),
                                     "rel_inf_outside_footprints": float(err_map[~foot].max()) if (~foot).any() else 0.0,
                                     "rel_inf_inside_footprints": float(err_map[foot].max()) if foot.any() else 0.0,
                                     "n_footprint_pixels": int(foot.sum()), "n_pixels": int(foot.size)},
                "ig_map_with_host_gates_forced_vs_golden_rel_inf": rel(forced, gold),
            }
        report["fixtures"][fx] = fx_rep
    with open(out_path, "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps({k: {c: {"flipped": v["n_flipped_gates"], "all": v["ig_map_vs_golden"]["rel_inf_all_pixels"],
                              "outside": v["ig_map_vs_golden"]["rel_inf_outside_footprints"],
                              "forced": v["ig_map_with_host_gates_forced_vs_golden_rel_inf"]} for c, v in r.items()}
                      for k, r in report["fixtures"].items()}, indent=1))


# ------------------------------------------------------------------------------------------- resnet
def first_difference(model, run, n=2):
    """Run `run(model)` n times recording every leaf module's output; -> (all identical?, first differing module name,
    type, max abs difference there)."""
    names = {m: k for k, m in model.named_modules()}
    records = []
    for _ in range(n):
        rec = []
        hooks = [m.register_forward_hook(lambda mod, i, o, rec=rec: rec.append((names[mod], type(mod).__name__, o.detach().clone())))
                 for m in model.modules() if not list(m.children())]
        out = run(model)
        for h in hooks:
            h.remove()
        records.append((rec, out.detach().clone()))
    base_rec, base_out = records[0]
    for rec, out in records[1:]:
        for (n0, t0, a), (_, _, b) in zip(base_rec, rec):
            if not torch.equal(a, b):
                return False, n0, t0, float((a - b).abs().max()), float((out - base_out).abs().max())
    return all(torch.equal(base_out, o) for _, o in records[1:]), None, None, 0.0, 0.0


def resnet_report(mode, out_path):
    import time
    from xai_engine.zoo import resnet50
    from xai_engine.sweep import PerturbationSweep, run_perturbation, KEYS
    from xai_engine.ig import IG
    from xai_engine.prepare import use_tuned_miopen_db, fuse_bn_relu
    if mode in ("finddb", "finddb_deterministic"):
        torch.backends.cudnn.benchmark = use_tuned_miopen_db(0)
    else:
        torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = mode in ("deterministic", "finddb_deterministic")
    rep = {"mode": mode, "benchmark": bool(torch.backends.cudnn.benchmark), "deterministic": bool(torch.backends.cudnn.deterministic),
           "device": torch.cuda.get_device_name(0)}
    model = resnet50(seed=0).to(DEV)
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1000))
    with torch.no_grad():
        t = model(x.to(DEV)).argmax(1)[0]
    sal = IG(x, model, 50, 50, 1, 0, DEV, t).sum(0).abs().cpu().numpy()

    # (1) run-to-run reproducibility of the classifier itself
    batch = torch.randn(50, 3, 224, 224, generator=torch.Generator().manual_seed(7)).to(DEV)

    def fwd(m):
        with torch.no_grad():
            return m(batch)
    same, name, typ, dmod, dout = first_difference(model, fwd, n=3)
    rep["forward_50_images"] = {"bit_identical_over_3_runs": same, "first_differing_module": name, "module_type": typ,
                                "max_abs_diff_there": dmod, "max_abs_diff_logits": dout}
    gs = []
    for _ in range(3):
        b = batch.clone().requires_grad_(True)
        (gr,) = torch.autograd.grad(model(b)[:, int(t)].sum(), b)
        gs.append(gr)
    rep["forward_backward_50_images"] = {"input_gradient_bit_identical_over_3_runs": bool(torch.equal(gs[0], gs[1]) and torch.equal(gs[0], gs[2])),
                                         "max_rel_diff": max(rel(gs[i].cpu().numpy(), gs[0].cpu().numpy()) for i in (1, 2))}
    a = IG(x, model, 50, 50, 1, 0, DEV, t)
    b = IG(x, model, 50, 50, 1, 0, DEV, t)
    rep["IG_twice"] = {"bit_identical": bool(torch.equal(a, b)), "rel_inf": rel(b.cpu().numpy(), a.cpu().numpy())}

    # (2) fused sweep vs the reference's 8-run flow, same image, same map
    for tag, mdl in (("classifier_as_given", model), ("fused_bn_relu", None)):
        if mdl is None:
            mdl = fuse_bn_relu(model, verify=torch.randn(2, 3, 224, 224, device=DEV), fork_residual=True)
        td = {"models": [mdl], "img_hw": 224, "batch_size": 50, "device": DEV}
        sw = PerturbationSweep(mdl, 224, DEV)
        f1, c1 = sw.run(x, sal, return_curves=True)
        f2, c2 = sw.run(x, sal, return_curves=True)
        t0 = time.perf_counter()
        eight = run_perturbation(x, sal, td)
        t8 = time.perf_counter() - t0
        # raw curves of the 8-run flow for the comparison: PNP returns the raw response (morf == deletion sequence, lerf)
        from xai_engine.perturb import PositiveNegativePerturbation, MonotonicityMetric
        from xai_engine.blur import GaussianBlur
        _, morf = PositiveNegativePerturbation(mdl, 224 * 224, "morf", 224, torch.zeros_like).single_run(x, sal, DEV, max_batch_size=50)
        _, lerf = PositiveNegativePerturbation(mdl, 224 * 224, "lerf", 224, torch.zeros_like).single_run(x, sal, DEV, max_batch_size=50)
        ins_raw, _ = MonotonicityMetric(mdl, 224 * 224, "positive", 224, GaussianBlur(31, 31, DEV)).single_run(x, sal, DEV, max_batch_size=50)
        raw = {"ins": ins_raw, "dele": morf, "lerf": lerf}
        entry = {"auc_abs_diff_fused_vs_8_runs": {k: abs(float(f1[k]) - float(eight[k])) for k in KEYS},
                 "auc_abs_diff_fused_run_to_run": {k: abs(float(f1[k]) - float(f2[k])) for k in KEYS},
                 "max_auc_abs_diff_fused_vs_8_runs": max(abs(float(f1[k]) - float(eight[k])) for k in KEYS),
                 "eight_run_seconds": t8, "raw_response": {}}
        for k in ("ins", "dele", "lerf"):
            dd = np.abs(np.asarray(c1[k]) - np.asarray(raw[k]))
            nz = np.nonzero(dd)[0]
            entry["raw_response"][k] = {"max_abs_diff_fused_vs_8_runs": float(dd.max()), "first_differing_step": int(nz[0]) if nz.size else None,
                                        "n_differing_steps": int(nz.size), "max_abs_diff_fused_run_to_run": float(np.abs(np.asarray(c1[k]) - np.asarray(c2[k])).max()),
                                        "p_original": float(c1["dele"][0]), "p_range": [float(np.min(c1[k])), float(np.max(c1[k]))]}
        # amplification of the normalised curves: (r - base) / |orig - base|
        with torch.no_grad():
            lg = mdl(torch.cat([x.to(DEV), GaussianBlur(31, 31, DEV)(x), torch.zeros_like(x).to(DEV)]))
            p = torch.softmax(lg, 1)[:, int(t)].cpu().numpy()
        entry["p_original_blur_zero"] = [float(v) for v in p]
        entry["amplification_1_over_abs_orig_minus_base"] = {"ins(blur)": float(1 / abs(p[0] - p[1])), "del(zero)": float(1 / abs(p[0] - p[2]))}
        rep[tag] = entry
    with open(out_path, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) or k in ("forward_50_images", "forward_backward_50_images", "IG_twice") else
                          {"max_auc_diff": v.get("max_auc_abs_diff_fused_vs_8_runs"), "raw": {kk: vv["max_abs_diff_fused_vs_8_runs"] for kk, vv in v.get("raw_response", {}).items()},
                           "amp": v.get("amplification_1_over_abs_orig_minus_base")}) for k, v in rep.items()}, indent=1))


def kernels_report(mode, batch, out_path, module=None):
    """Which module of ResNet-50 is not bit-reproducible for a `batch`-image forward in `mode`, and -- when run under
    `rocprofv3 --kernel-trace` -- which kernels it launches: after the search the module is run alone 10 times between two
    marker launches (the library's own sumsq kernel), so its dispatches are the ones between the markers in the trace
    (profiles/between_markers.py lists them)."""
    from xai_engine.zoo import resnet50
    from xai_engine.prepare import use_tuned_miopen_db
    torch.backends.cudnn.benchmark = use_tuned_miopen_db(0) if mode in ("finddb", "finddb_deterministic") else False
    torch.backends.cudnn.deterministic = mode in ("deterministic", "finddb_deterministic")
    model = resnet50(seed=0).to(DEV)
    x = torch.randn(batch, 3, 224, 224, generator=torch.Generator().manual_seed(7)).to(DEV)
    names = {m: k for k, m in model.named_modules()}
    runs = []
    for _ in range(4):
        rec = []
        hooks = [m.register_forward_hook(lambda mod, i, o, rec=rec: rec.append((mod, i[0].detach(), o.detach().clone())))
                 for m in model.modules() if not list(m.children())]
        with torch.no_grad():
            model(x)
        for h in hooks:
            h.remove()
        runs.append(rec)
    culprit = None
    for j in range(len(runs[0])):
        if any(not torch.equal(runs[0][j][2], r[j][2]) for r in runs[1:]):
            culprit = runs[0][j]
            break
    if culprit is None and module:                 # under the profiler the race may not show: still run the named module alone
        culprit = next(r for r in runs[0] if names[r[0]] == module)
        rep_named = True
    else:
        rep_named = False
    rep = {"mode": mode, "batch": batch, "deterministic": bool(torch.backends.cudnn.deterministic), "benchmark": bool(torch.backends.cudnn.benchmark),
           "module_was_named_not_found": rep_named}
    if culprit is None:
        rep["first_non_reproducible_module"] = None
    else:
        mod, inp, out = culprit
        rep["first_non_reproducible_module"] = {"name": names[mod], "module": repr(mod), "input_shape": list(inp.shape), "output_shape": list(out.shape)}
        torch.cuda.synchronize()
        from xai_engine import kernels as K
        marker = torch.ones(1, 64, device=DEV)
        K.sumsq(marker)
        outs = []
        with torch.no_grad():
            for _ in range(10):
                outs.append(mod(inp).clone())
        K.sumsq(marker)
        torch.cuda.synchronize()
        rep["module_alone_10_runs_distinct_results"] = len({o.cpu().numpy().tobytes() for o in outs})
        rep["module_alone_max_abs_diff"] = max(float((o - outs[0]).abs().max()) for o in outs)
    with open(out_path, "w") as f:
        json.dump(rep, f, indent=1)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--module", default=None, help="kernels: run this module alone when no non-reproducible one shows up")
    ap.add_argument("what", choices=["gates", "resnet", "kernels"])
    ap.add_argument("--mode", default="immediate", choices=["immediate", "deterministic", "finddb", "finddb_deterministic"])
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    if a.what == "gates":
        gates_report(a.out)
    elif a.what == "kernels":
        kernels_report(a.mode, a.batch, a.out, a.module)
    else:
        resnet_report(a.mode, a.out)
