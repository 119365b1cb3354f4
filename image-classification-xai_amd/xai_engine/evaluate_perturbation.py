"""Command line of the harness counterpart -- the reference's flags
(XAI_Survey/evaluations/evaluatePerturbation.py:726-750: --image_count --model --attr_func --cuda_num
--dataset_path) plus what an offline box needs (weights from a local file, class-map path).

    python -m xai_engine.evaluate_perturbation --model R50 --attr_func ig --image_count 1000 \
        --dataset_path /data/ImageNet/val --class_map /data/correctly_classified_R50.txt --weights r50.pt
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m xai_engine.evaluate_perturbation ...

Pretrained torchvision / timm weights cannot be fetched here; `--weights` takes a state_dict saved with
torch.save (loaded with weights_only=True); without it the architecture runs with seeded random weights.
"""
import argparse

import torch

from . import dist as xd
from . import harness
from .sweep import CNN_ATTR_FUNCS, VIT_ATTR_FUNCS
from .zoo import resnet50, resnet101, resnet152, resnext101_64x4d, vit_base_patch16_224, vit_base_patch32_224

MODELS = {
    # name: (constructor, batch size, normalisation, num_patches) -- names, batch sizes and patch counts of the reference's
    # table (evaluatePerturbation.py:627-659); "R50" is this build's extra (BASELINE.json's configurations name ResNet-50)
    "R50": (resnet50, 50, (harness.CNN_MEAN, harness.CNN_STD), 0),
    "R101": (resnet101, 50, (harness.CNN_MEAN, harness.CNN_STD), 0),
    "R152": (resnet152, 50, (harness.CNN_MEAN, harness.CNN_STD), 0),
    "RNXT": (resnext101_64x4d, 25, (harness.CNN_MEAN, harness.CNN_STD), 0),
    "VIT16": (vit_base_patch16_224, 25, (harness.VIT_MEAN, harness.VIT_STD), 14),
    "VIT32": (vit_base_patch32_224, 50, (harness.VIT_MEAN, harness.VIT_STD), 7),
}


def build_parser():
    p = argparse.ArgumentParser("")
    p.add_argument("--image_count", type=int, default=1000, help="How many images to test with.")
    p.add_argument("--model", type=str, default="R50", help="Classifier to use: " + ", ".join(MODELS))
    p.add_argument("--attr_func", type=str, default="ig", help="attr to use: R50/R101/R152/RNXT: {" + ", ".join(CNN_ATTR_FUNCS) + "}, VIT16/VIT32: {" +
                   ", ".join(VIT_ATTR_FUNCS) + "}")
    p.add_argument("--cuda_num", type=int, default=0, help="GPU to use when not launched by torchrun.")
    p.add_argument("--dataset_path", type=str, default="../../../ImageNet", help="The path to your dataset input")
    p.add_argument("--class_map", type=str, default=None, help="correctly_classified_<MODEL>.txt (optional)")
    p.add_argument("--weights", type=str, default=None, help="state_dict file for the chosen architecture")
    p.add_argument("--eight_runs", action="store_true", help="drive the eight single_run calls like the reference instead of the fused sweep")
    p.add_argument("--fuse_bn_relu", action="store_true", help="CNN models: run eval-mode BatchNorm + ReLU (+ residual add) through the fused "
                   "HIP kernels (prepare.fuse_bn_relu; every call site is verified bit-identical to the PyTorch kernels first)")
    p.add_argument("--nondeterministic", action="store_true", help="allow MIOpen's non-deterministic solvers (atomics-based split-K implicit "
                   "GEMMs: run-to-run noise of ~1e-6 on probabilities, profiles/r02_resnet_determinism_*.json).  Default: "
                   "torch.backends.cudnn.deterministic = True -- bit-reproducible sweeps, fused flow == eight-run flow exactly, at the "
                   "throughput of MIOpen's immediate mode")
    p.add_argument("--batch_size", type=int, default=None, help="images per classifier pass in the metrics (`max_batch_size`); default: the "
                   "reference's value for the model (50 / 25).  Larger batches are faster (ResNet-50: 98 ms per image-sweep at 50, 91 at 112, "
                   "88 at 225) and move the ten numbers by <= 3e-6 (profiles/r02_exp_sweep_batch.txt)")
    p.add_argument("--reference_counter", action="store_true", help="fold the images' results exactly like the reference's `pert_result_counter "
                   "+= ...` (evaluatePerturbation.py:594-596: a key whose running sum is <= 0 is dropped) and write only the surviving CSV rows "
                   "(:612-615).  Default: plain sums, always ten rows")
    p.add_argument("--streams", type=int, default=3, help="HIP streams consecutive images are queued on (fused sweep; results are bit-identical "
                   "to 1 with the default deterministic solvers)")
    p.add_argument("--out_dir", type=str, default="pert_test_results")
    p.add_argument("--checkpoint", type=str, default=None, help="path prefix for per-rank resume files (the reference loses a crashed run)")
    return p


def main(argv=None):
    args, _ = build_parser().parse_known_args(argv)
    if args.model not in MODELS:
        raise SystemExit(f"unknown --model {args.model}; choose from {sorted(MODELS)}")
    if args.attr_func not in (VIT_ATTR_FUNCS if "VIT" in args.model else CNN_ATTR_FUNCS):
        print("Model-attribution mismatch, please use --help.")
        raise SystemExit(1)
    torch.backends.cudnn.deterministic = not args.nondeterministic
    rank, world, device = xd.init_from_env()
    if world == 1:
        device = torch.device("cuda", args.cuda_num)
        torch.cuda.set_device(device)
    ctor, batch_size, norm, num_patches = MODELS[args.model]
    batch_size = args.batch_size or batch_size
    model = ctor()
    if args.weights:
        model.load_state_dict(torch.load(args.weights, map_location="cpu", weights_only=True))
    model = model.to(device).eval()
    for p in model.parameters():
        p.requires_grad_(False)
    if args.fuse_bn_relu and "VIT" not in args.model:
        from .prepare import fuse_bn_relu
        model = fuse_bn_relu(model, verify=torch.randn(2, 3, 224, 224, device=device), fork_residual=True)
    testing_dict = {"models": [model, model], "imagenet_dataset": args.dataset_path, "normalize": norm, "img_hw": 224,
                    "batch_size": batch_size, "attr_func": args.attr_func, "model_name": args.model,
                    "image_count": args.image_count, "device": str(device), "class_map_path": args.class_map,
                    "weights_path": args.weights or "", "num_patches": num_patches}
    total, used, _ = harness.evaluate_perturbation(testing_dict, rank=rank, world=world, fused=not args.eight_runs, out_dir=args.out_dir,
                                                     checkpoint=args.checkpoint, streams=args.streams, reference_counter=args.reference_counter)
    if rank == 0:
        fold = "reference Counter += (keys with a running sum <= 0 dropped)" if args.reference_counter else "plain sums"
        print(f"{used} images; {fold}; means: " + ", ".join(f"{k}={total[k] / max(used, 1):.6f}" for k in total))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
