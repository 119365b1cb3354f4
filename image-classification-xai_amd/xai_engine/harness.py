"""Dataset-facing side of the harness counterpart (reference
XAI_Survey/evaluations/evaluatePerturbation.py: evaluate_perturbation :499-620, main :622-751).

Formats kept from the reference:
  * images `ILSVRC2012_val_XXXXXXXX.JPEG` in one directory, visited in sorted order (:520);
    the validation index is `int(name.split("_")[2].split(".")[0]) - 1` (:528);
  * `correctly_classified_<MODEL>.txt`: 50 000 lines of 0/1 indexed by that number (:507,:530);
  * transform = Resize(img_hw) -> CenterCrop(img_hw) -> ToTensor, then Normalize (:680-694) -- done
    with PIL exactly as torchvision does for PIL inputs (torchvision itself is not required);
  * output `pert_test_results/<model>/<attr>_<count>_images.csv`, rows `key,mean` (:606-618).

The order-dependent image filters (:530,:540,:569,:573-576) are run as a deterministic
selection pre-pass (`select_images`) that every rank executes identically, so the image list --
and therefore the result -- does not depend on the number of GPUs; the selected images are
then sharded round-robin (`sweep.sweep_images`) and reduced with one all-reduce.
"""
import os
import time

import numpy as np
import torch

from . import sweep as _sweep
from .blur import GaussianBlur
from .ig import hip_device, _logits_of

CNN_MEAN, CNN_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)          # evaluatePerturbation.py:683
VIT_MEAN, VIT_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)                      # :686


def load_image(path, img_hw):
    """PIL image -> float32 (C, h, w) in [0,1]: Resize(img_hw) (shorter side, bilinear, PIL's
    antialiasing), CenterCrop(img_hw), ToTensor.  Grey-scale files come back with C == 1 and are
    rejected by the caller's shape test, as in the reference (:540)."""
    from PIL import Image
    img = Image.open(path)
    w, h = img.size
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = img_hw, int(img_hw * long_ / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    if (w, h) != (nw, nh):
        img = img.resize((nw, nh), Image.BILINEAR)
    left, top = int(round((nw - img_hw) / 2.0)), int(round((nh - img_hw) / 2.0))
    img = img.crop((left, top, left + img_hw, top + img_hw))
    arr = np.array(img)                                   # writable copy
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).float().div(255)


def normalize(t, mean, std):
    m = torch.tensor(mean, dtype=t.dtype).view(-1, 1, 1)
    s = torch.tensor(std, dtype=t.dtype).view(-1, 1, 1)
    return (t - m) / s


def image_number(name):
    """0-based validation index from `ILSVRC2012_val_00000123.JPEG` (:528)."""
    return int((name.split("_")[2]).split(".")[0]) - 1


def _pred(model, x, dev, cls=None):
    """(class, softmax probability of that class) like get_classifier_pred (:76-80)."""
    with torch.no_grad():
        out = _logits_of(model(x.to(dev)))
    c = int(out.argmax(1)[0]) if cls is None else cls
    return c, float(torch.softmax(out, 1)[0, c])


class SelectedImages:
    """The selected images as a sequence of normalised (1,3,H,W) CPU tensors, loaded from disk on access: a rank of a
    sharded sweep only ever touches the images it owns."""

    def __init__(self, root, names, img_hw, mean, std):
        self.root, self.names, self.img_hw, self.mean, self.std = root, list(names), img_hw, mean, std

    def __len__(self):
        return len(self.names)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self.names)))]
        return normalize(load_image(os.path.join(self.root, self.names[i]), self.img_hw), self.mean, self.std).unsqueeze(0)


def _verdict(model, blur, dev, path, img_hw, mean, std):
    """What the reference's loop learns about ONE candidate file, independently of every other file:
    (is RGB of the right shape (:540), passes the blur / black sanity check (:563-569), predicted class)."""
    trans = load_image(path, img_hw)
    if tuple(trans.shape) != (3, img_hw, img_hw):
        return False, False, -1
    x = normalize(trans, mean, std).unsqueeze(0)
    target, p_orig = _pred(model, x, dev)
    blur_cls, p_blur = _pred(model, blur(x), dev)              # each substrate's OWN top class and its probability (:563-566)
    black_cls, p_black = _pred(model, torch.zeros_like(x), dev)
    sane = not (p_blur >= p_orig or p_black >= p_orig or target == black_cls or target == blur_cls)
    return True, sane, target


class ClassQuota:
    """The order-dependent part of the selection (evaluatePerturbation.py:573-576 and the stop at :520-524): candidates are
    offered in file order with their verdicts; at most ceil(count / num_classes) images per predicted class, `count` in all."""

    def __init__(self, count, num_classes):
        self.count = count
        self.per_class = int(np.ceil(count / num_classes))
        self.used = [0] * num_classes
        self.chosen = []                                    # (name, class)

    @property
    def full(self):
        return len(self.chosen) == self.count

    def offer(self, name, rgb, sane, target):
        if self.full or not rgb or not sane or self.used[target] == self.per_class:
            return
        self.used[target] += 1
        self.chosen.append((name, target))


def select_images(testing_dict, correctly_classified=None, names=None, rank=0, world=1, chunk_per_rank=32, lazy=False):
    """Deterministic pre-pass reproducing the reference's in-loop filters, in its order:
    bitmap (:530) -> RGB shape (:540) -> blur/black sanity (:569) -> per-class quota (:573-576),
    stopping at `image_count`.  Returns a list of (file name, normalised (1,3,H,W) CPU tensor, class); with `lazy`
    the tensors are not loaded: (names, SelectedImages, classes).

    Only the quota and the stop are order-dependent, and they need nothing but each candidate's verdict (RGB?, sane?, class)
    -- three single-image classifier passes and a file read that do not depend on any other file.  So the verdicts are
    computed SHARDED: candidates are taken in file order in chunks of `chunk_per_rank * world`, rank r judges the
    candidates r, r + world, ... of the chunk, one all-reduce(SUM) of the chunk's small int32 verdict table (disjoint rows)
    hands every rank all verdicts, and every rank replays quota + stop over them in file order -- the same list on every
    rank and for every world size, at 1/world of the serial cost (a replicated pre-pass would cost more than the 8-GPU
    sweep it precedes: ~15 ms per candidate against ~12 ms of sweep per selected image and rank)."""
    dev = hip_device(testing_dict["device"])
    model = testing_dict["models"][0]
    root = testing_dict["imagenet_dataset"]
    img_hw = testing_dict["img_hw"]
    mean, std = testing_dict.get("normalize", (CNN_MEAN, CNN_STD))
    count = testing_dict["image_count"]
    quota = ClassQuota(count, testing_dict.get("num_classes", 1000))
    blur = GaussianBlur(31, 31, dev)
    candidates = [n for n in (names if names is not None else sorted(os.listdir(root)))
                  if correctly_classified is None or correctly_classified[image_number(n)] != 0]          # bitmap filter: host only
    step = max(1, chunk_per_rank) * world
    for lo in range(0, len(candidates), step):
        if quota.full:
            break
        part = candidates[lo:lo + step]
        table = torch.zeros((len(part), 3), dtype=torch.int32)                 # [is_rgb (-1: could not be judged), sane, class + 1] per candidate
        errors = {}
        for j in range(rank, len(part), world):
            try:
                rgb, sane, target = _verdict(model, blur, dev, os.path.join(root, part[j]), img_hw, mean, std)
                table[j] = torch.tensor([int(rgb), int(sane), target + 1], dtype=torch.int32)
            except Exception as e:                                               # unreadable / corrupt file: recorded in the table, never raised
                table[j] = torch.tensor([-1, 0, 0], dtype=torch.int32)          # here -- the other ranks are on their way into the collective
                errors[j] = e
        if world > 1:
            from . import dist as _xd
            import torch.distributed as _dist
            if not _dist.is_initialized():
                raise RuntimeError("select_images(world > 1) needs an initialised torch.distributed process group (xai_engine.dist.init_from_env)")
            if _dist.get_backend() == "nccl":
                table = _xd.all_reduce_sum(table.to(dev)).cpu()
            else:
                table = _xd.all_reduce_sum(table)
        for j, (name, (rgb, sane, tplus)) in enumerate(zip(part, table.tolist())):   # order-dependent part, replicated, host only
            if quota.full:
                break                # the reference stops READING files here (:522-524): a bad file past this point never mattered
            if rgb < 0:
                # every rank replays the same table, so every rank raises here, with the file's name -- no rank is left in a collective
                why = f": {type(errors[j]).__name__}: {errors[j]}" if j in errors else f" (judged on rank {j % world})"
                raise RuntimeError(f"select_images: {os.path.join(root, name)} could not be read or classified{why}") from errors.get(j)
            quota.offer(name, bool(rgb), bool(sane), tplus - 1)
    chosen = quota.chosen
    images = SelectedImages(root, [c[0] for c in chosen], img_hw, mean, std)
    if lazy:
        return [c[0] for c in chosen], images, [c[1] for c in chosen]
    return [(name, images[i], target) for i, (name, target) in enumerate(chosen)]


def evaluate_perturbation(testing_dict, rank=0, world=1, fused=True, out_dir="pert_test_results", checkpoint=None, streams=1,
                          reference_counter=False):
    """Selection pre-pass, attribution + ten metrics per selected image (sharded over ranks), CSV on
    rank 0.  `testing_dict` has the reference's keys (:705-718): models, imagenet_dataset, img_hw,
    batch_size, attr_func, model_name, image_count, device (+ optional normalize=(mean, std),
    class_map_path).  `reference_counter`: fold the images' Counters and write the CSV exactly as the reference does
    (`+=` in file order drops keys with a running sum <= 0, only surviving keys are written; :594-596,:612-615) -- the
    default is plain sums and always ten rows.  `streams`: HIP streams consecutive images are queued on."""
    t_start = time.time()
    cc = None
    path = testing_dict.get("class_map_path")
    if path:
        cc = np.loadtxt(path).astype(np.int64)
    names, images, _classes = select_images(testing_dict, cc, rank=rank, world=world, lazy=True)
    model = testing_dict["models"][0]
    dev = hip_device(testing_dict["device"])

    is_vit = "VIT" in testing_dict["model_name"]                # the reference picks the dispatch table by model name (:577-582)

    # CNN methods + fused sweep: attribution maps stay on the device (no host sync between attribution and sweep)
    td_attr = dict(testing_dict, device_maps=True) if (fused and not is_vit) else testing_dict

    def attr_fn(x, target):
        return (_sweep.get_VIT_attr if is_vit else _sweep.get_CNN_attr)(x, None, target, td_attr)

    identity = _sweep.sweep_identity(attr_func=testing_dict["attr_func"], model_name=testing_dict["model_name"],
                                     image_count=testing_dict["image_count"], files="|".join(names),
                                     weights=testing_dict.get("weights_path", ""), fold="reference" if reference_counter else "sums")
    total, used, attr_time = _sweep.sweep_images(images, model, dev, attr_fn, img_hw=testing_dict["img_hw"],
                                                 batch_size=testing_dict["batch_size"], fused=fused, rank=rank, world=world,
                                                 testing_dict=testing_dict, checkpoint=checkpoint, identity=identity,
                                                 streams=1 if is_vit else streams,      # the hooked ViT keeps per-pass state on its modules
                                                 reference_counter=reference_counter, kind=testing_dict["attr_func"])
    if rank == 0 and used:
        name = f'{testing_dict["attr_func"]}_{testing_dict["image_count"]}_images.csv'
        _sweep.write_csv(os.path.join(out_dir, testing_dict["model_name"], name), total, used, attr_time, time.time() - t_start,
                         reference_counter=reference_counter)
    return total, used, names
