#!/usr/bin/env python3
"""Secondary benchmark: the other BASELINE.json configurations on synthetic inputs (the driver's
contract line is bench.py; this script produces the supporting numbers kept under profiles/).

    python bench_configs.py [--configs 1,3,4,5] [--rise-masks 8000] [--sweep-images 4]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench_configs.py ...

  1  Grad-CAM, ResNet-50, one 224x224 image           (latency; parity vs oracle)
  2  IG / Left-IG through the reference's one-image API (bench.py measures the batched path)
  3  RISE, N masks, ResNet-50, masks sharded over ranks (masks/s; one all-reduce of the partial map)
  4  IG 50 steps on ViT-B/16 (hooked), batch 25 + attention-space IG 20 steps
  5  insertion/deletion sweep, images sharded over ranks (images/s; one 96-byte all-reduce)
  6  RISE-family maskers on ViT-B/16: ViT-CX and TIS end to end, one image (latency)
One JSON line per configuration on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "image-classification-xai_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def sync(dev):
    torch.cuda.synchronize(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2,3,4,5,6")
    ap.add_argument("--rise-masks", type=int, default=8000)
    ap.add_argument("--rise-batch", type=int, default=250)
    ap.add_argument("--sweep-images", type=int, default=12, help="images per rank")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams for RISE's mask batches and the sweep's images (bit-identical to 1)")
    ap.add_argument("--check", type=int, default=1, help="1 = also compare a reduced case with the CPU oracle")
    ap.add_argument("--miopen-db", type=int, default=1, help="1 = MIOpen find mode on the shipped find-db (see xai_engine/prepare.py)")
    ap.add_argument("--fuse-bn-relu", type=int, default=1, help="1 = ResNet-50's eval BN + ReLU (+ add) through the fused HIP kernels "
                    "(xai_engine/prepare.py: fuse_bn_relu, verified per call site); 0 = classifier exactly as given")
    ap.add_argument("--record-db", default=None, help="directory to record a find-db into (exhaustive find: minutes)")
    ap.add_argument("--deterministic", type=int, default=0, help="1 = torch.backends.cudnn.deterministic: bit-reproducible classifier passes "
                    "(immediate-mode throughput; the find-db is not used)")
    args = ap.parse_args()
    want = {int(c) for c in args.configs.split(",")}

    from xai_engine import dist as xd
    rank, world, dev = xd.init_from_env()
    import xai_engine
    xai_engine.load_library()
    if args.record_db:
        os.makedirs(args.record_db, exist_ok=True)
        os.environ["MIOPEN_USER_DB_PATH"] = args.record_db
        torch.backends.cudnn.benchmark = True
    elif args.miopen_db and not args.deterministic:
        from xai_engine.prepare import use_tuned_miopen_db
        torch.backends.cudnn.benchmark = use_tuned_miopen_db(rank)
    torch.backends.cudnn.deterministic = bool(args.deterministic)
    from xai_engine.zoo import resnet50, vit_base_patch16_224
    from xai_engine.ig import IG, ig_batch
    from xai_engine.gradcam import gradcam_saliency
    from xai_engine.rise import draw_masks, rise
    from xai_engine.sweep import PerturbationSweep, sweep_images, KEYS, run_perturbation
    from xai_engine.vit_attr import Baselines

    def emit(d):
        if rank == 0:
            if d["config"] in (1, 2, 3, 5):
                d["classifier_prep"] = prep
            print(json.dumps(d), flush=True)

    def rel(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))

    resnet = resnet50(seed=0).to(dev) if want & {1, 2, 3, 5} else None
    prep = "none"
    if resnet is not None and args.fuse_bn_relu and not args.record_db:
        from xai_engine.prepare import fuse_bn_relu
        resnet = fuse_bn_relu(resnet, verify=torch.randn(2, 3, 224, 224, device=dev), fork_residual=True)
        prep = "BN+ReLU(+add) fused, call sites verified bit-identical"

    if 2 in want:
        xs = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(2))
        with torch.no_grad():
            ts = resnet(xs.to(dev)).argmax(1)
        for bs in (50, 25):
            IG(xs[:1], resnet, 50, bs, 1, 0, dev, ts[0])
            sync(dev); t0 = time.perf_counter()
            for i in range(8):
                IG(xs[i:i + 1], resnet, 50, bs, 1, 0, dev, ts[i])
            sync(dev); dt = (time.perf_counter() - t0) / 8
            emit({"config": 2, "workload": f"IG 50 steps ResNet-50, reference API (one image per call, batch_size={bs})",
                  "ms_per_attribution": dt * 1e3, "attributions_per_s": 1 / dt, "n_gpus": world})
        sync(dev); t0 = time.perf_counter()
        for i in range(8):
            IG(xs[i:i + 1], resnet, 50, 50, .9, 0, dev, ts[i])
        sync(dev); dt = (time.perf_counter() - t0) / 8
        emit({"config": 2, "workload": "Left-IG (alpha_star=0.9) 50 steps ResNet-50, reference API, batch_size=50",
              "ms_per_attribution": dt * 1e3, "attributions_per_s": 1 / dt, "n_gpus": world})

    if 1 in want:
        x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
        with torch.no_grad():
            t = resnet(x).argmax(1)[0]
        for _ in range(3):
            sal = gradcam_saliency(resnet, resnet.layer4, x, t, (224, 224))
        sync(dev); t0 = time.perf_counter()
        for _ in range(20):
            sal = gradcam_saliency(resnet, resnet.layer4, x, t, (224, 224))
        sync(dev); dt = (time.perf_counter() - t0) / 20
        err = None
        if args.check:
            from oracle import gradcam as ogc
            act, grad = ogc.layer_act_and_grad(resnet, resnet.layer4, x, int(t))
            err = rel(sal[0].cpu().numpy(), ogc.gradcam_saliency(act, grad, 224, 224)[0])
        from xai_engine.gradcam import CapturedGradCam
        cap = CapturedGradCam(resnet, resnet.layer4, x, (224, 224))
        cap(x, t)
        sync(dev); t0 = time.perf_counter()
        for _ in range(20):
            sal_g = cap(x, t)
        sync(dev); dtg = (time.perf_counter() - t0) / 20
        emit({"config": 1, "workload": "Grad-CAM ResNet-50 layer4, one 3x224x224 image", "ms_per_attribution": dt * 1e3,
              "attributions_per_s": 1 / dt, "rel_err_vs_oracle": err, "ms_per_attribution_hipgraph_replay": dtg * 1e3,
              "attributions_per_s_hipgraph_replay": 1 / dtg, "hipgraph_vs_eager_rel_diff": rel(sal_g.cpu().numpy(), sal.cpu().numpy()),
              "n_gpus": world})

    if 3 in want:
        N, s, p1 = args.rise_masks, 8, 0.5
        x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            t = int(resnet(x.to(dev)).argmax(1)[0])
        score = lambda b: torch.softmax(resnet(b), 1)[:, t]       # noqa: E731
        np.random.seed(3)
        masks = draw_masks((224, 224), N, s, p1)
        # warm-up on the same streams: every stream worker meets its first batch here (in find mode that is a find per thread handle)
        xd.rise_sharded(resnet, x, None, dev, N=min(N, 1000), s=s, p1=p1, score_fn=score, batch_size=args.rise_batch,
                        masks=tuple(m[:min(N, 1000)] if i < 2 else m for i, m in enumerate(masks)), streams=args.streams)
        sync(dev); t0 = time.perf_counter()
        sal = xd.rise_sharded(resnet, x, None, dev, N=N, s=s, p1=p1, score_fn=score, batch_size=args.rise_batch, masks=masks, streams=args.streams)
        sync(dev); dt = time.perf_counter() - t0
        err = None
        if args.check and rank == 0:
            from oracle import rise as orise
            n = 100
            sub = (masks[0][:n], masks[1][:n], masks[2])
            got = rise(resnet, x, None, dev, N=n, s=s, p1=p1, score_fn=score, masks=sub).cpu().numpy()

            def score_np(b):
                with torch.no_grad():
                    return score(torch.from_numpy(b).to(dev)).cpu().numpy()
            err = rel(got, orise.rise(score_np, x.numpy(), n, s, p1, sub[0].astype(np.float32), sub[1], sub[2]))
        emit({"config": 3, "workload": f"RISE N={N} s=8 p1=0.5, ResNet-50, masks sharded x{world}", "seconds": dt,
              "masks_per_s": N / dt, "saliency_maps_per_s": 1 / dt, "rel_err_vs_oracle_100_masks": err, "n_gpus": world,
              "collective": "1 all_reduce(SUM) of (224,224) fp64 = 401 KB + broadcast of the mask draw"})

    if want & {4, 6}:
        # a ViT has one convolution (the patch embedding): MIOpen's search for it costs a minute (its candidates include a
        # naive kernel at 7 s per trial) and buys nothing -- immediate mode for the ViT configurations
        torch.backends.cudnn.benchmark = False

    if 4 in want:
        vit = vit_base_patch16_224(seed=0).to(dev)
        x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(4))
        with torch.no_grad():
            t = vit(x.to(dev)).argmax(1)[0]
        IG(x, vit, 50, 25, 1, 0, dev, t)
        sync(dev); t0 = time.perf_counter()
        for _ in range(5):
            out = IG(x, vit, 50, 25, 1, 0, dev, t)
        sync(dev); dt = (time.perf_counter() - t0) / 5
        xb = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(5)).to(dev)
        with torch.no_grad():
            tb = vit(xb).argmax(1)
        ig_batch(xb, vit, tb, images_per_pass=2)
        sync(dev); t0 = time.perf_counter()
        ig_batch(xb, vit, tb, images_per_pass=2)
        sync(dev); dtb = (time.perf_counter() - t0) / 8
        # pixel-space IG never reads the state the hooked ViT keeps on its modules, so its passes may overlap on stream workers too
        xs24 = torch.cat([xb, xb.flip(0), xb.roll(1, 0)])
        ts24 = torch.cat([tb, tb.flip(0), tb.roll(1, 0)])
        ig_batch(xs24, vit, ts24, images_per_pass=1, streams=args.streams)
        sync(dev); t0 = time.perf_counter()
        got_s = ig_batch(xs24, vit, ts24, images_per_pass=1, streams=args.streams)
        sync(dev); dts = (time.perf_counter() - t0) / 24
        one_s = ig_batch(xs24, vit, ts24, images_per_pass=1)
        streams_equal = bool(torch.equal(got_s, one_s))
        b = Baselines(vit)
        b.IG(x, t, steps=20, device=dev)
        sync(dev); t0 = time.perf_counter()
        for _ in range(5):
            a = b.IG(x, t, steps=20, device=dev)
        sync(dev); dta = (time.perf_counter() - t0) / 5
        err = erra = None
        if args.check:
            from oracle import ig as oig
            from oracle import vit_attr as ovit
            err = rel(out.cpu().numpy(), oig.ig(x.numpy(), vit, 50, 25, 1, 0, int(t)))
            erra = rel(a.cpu().numpy(), ovit.attention_ig(vit, x.numpy(), int(t), 20))
        emit({"config": 4, "workload": "IG 50 steps batch 25, ViT-B/16 (hooked, seeded random weights), 3x224x224",
              "ms_per_attribution_reference_api": dt * 1e3, "attributions_per_s_reference_api": 1 / dt,
              "attributions_per_s_ig_batch": 1 / dtb, f"attributions_per_s_ig_batch_{args.streams}_streams_1_image_per_pass": 1 / dts,
              "streams_bit_identical_to_one_stream": streams_equal, "attention_ig_20_steps_ms": dta * 1e3,
              "rel_err_vs_oracle_same_device_model": err, "attention_ig_rel_err_vs_oracle": erra, "n_gpus": world})

    if 5 in want:
        n_img = args.sweep_images * world
        images = [torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(1000 + i)) for i in range(n_img)]

        def attr_fn(x, target):
            return IG(x, resnet, 50, 50, 1, 0, dev, target).sum(0).abs()      # stays on the device: no drain between images
        sweep_images(images[:args.streams * world], resnet, dev, attr_fn, rank=rank, world=world, streams=args.streams, kind="ig")   # warm-up: every worker
        sync(dev); t0 = time.perf_counter()
        total, used, attr_t = sweep_images(images, resnet, dev, attr_fn, rank=rank, world=world, streams=args.streams, kind="ig")
        sync(dev); dt = time.perf_counter() - t0
        extra = {}
        if args.check and rank == 0:
            x0 = images[0]
            with torch.no_grad():
                t0_ = resnet(x0.to(dev)).argmax(1)[0]
            sal = attr_fn(x0, t0_).cpu().numpy()
            fused = PerturbationSweep(resnet, 224, dev).run(x0, sal)
            eight = run_perturbation(x0, sal, {"models": [resnet], "img_hw": 224, "batch_size": 50, "device": str(dev)})
            extra["max_abs_diff_fused_vs_8_runs"] = max(abs(fused[k] - eight[k]) for k in KEYS)
        emit({"config": 5, "workload": f"IG attribution + 10 ins/del metrics (224 steps), ResNet-50, {n_img} synthetic images, "
                                       f"image-sharded x{world}", "images": used, "seconds": dt, "images_per_s": used / dt,
              "attr_seconds_all_ranks": attr_t, "metric_means": {k: total[k] / used for k in KEYS}, "n_gpus": world,
              "collective": "1 all_reduce(SUM) of 12 fp64 = 96 B", **extra})

    if 6 in want:
        from xai_engine.vit_cx import ViT_CX
        from xai_engine.tis import TIS
        vit = vit_base_patch16_224(seed=0).to(dev)
        x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(6))
        torch.manual_seed(6); np.random.seed(6)
        res = {}
        for name, kw in (("host_noise", {}), ("device_noise", {"device_noise": True})):
            ViT_CX(vit, x, vit.blocks[-1].norm1, device=str(dev), return_feature_map=False, **kw)
            sync(dev); t0 = time.perf_counter()
            for _ in range(3):
                ViT_CX(vit, x, vit.blocks[-1].norm1, device=str(dev), return_feature_map=False, **kw)
            sync(dev); res[name] = (time.perf_counter() - t0) / 3 * 1e3
        from xai_engine import kernels as K
        from xai_engine.vit_cx import feature_maps, cluster_masks, reshape_function_vit
        import torch.nn as nn
        _, fmap = feature_maps(nn.Sequential(vit, nn.Softmax(dim=1)), x.to(dev), vit.blocks[-1].norm1, reshape_function_vit)
        sync(dev); t0 = time.perf_counter()
        masks, labels = cluster_masks(K.up_rownorm(fmap, 224, 224), 0.1)
        sync(dev); t_masks = (time.perf_counter() - t0) * 1e3
        tis = TIS(vit, batch_size=64)
        tis(x.to(dev))
        sync(dev); t0 = time.perf_counter()
        for _ in range(3):
            tis(x.to(dev))
        sync(dev); t_tis = (time.perf_counter() - t0) / 3 * 1e3
        emit({"config": 6, "workload": "ViT-CX (feature-map maskers) and TIS (1024 token masks) on ViT-B/16, one 3x224x224 image, "
                                       "seeded random weights", "vit_cx_ms": res["host_noise"], "vit_cx_device_noise_ms": res["device_noise"],
              "vit_cx_clusters": int(masks.shape[0]), "vit_cx_mask_build_ms_incl_host_clustering": t_masks, "tis_ms": t_tis,
              "n_gpus": world})

    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
