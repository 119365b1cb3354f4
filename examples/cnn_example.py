#!/usr/bin/env python3
"""The calls of the reference's XAI_Survey/notebooks/CNN_example.ipynb (cell 3: attr.IG, LayerGradCam) followed by
the perturbation metrics of XAI_Survey/evaluations/evaluatePerturbation.py:448-497, written exactly as a user of
the reference would write them -- only sys.path points at this repository's drop-in `util` package and the
classifier is a seeded random ResNet-50 (no network for pretrained weights).

    python examples/cnn_example.py            # needs an MI355X / HIP device
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-classification-xai_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from util.attribution_methods import saliencyMethods as attr          # noqa: E402  reference module path
from util.test_methods import MASTestFunctions as MAS                  # noqa: E402
from util.test_methods import AICTestFunctions as PIC                  # noqa: E402
from util.test_methods import PosNegPertFunctions as PNP               # noqa: E402
from util.test_methods import MonotonicityTest as MONO                 # noqa: E402
from util import model_utils                                           # noqa: E402
from xai_engine.gradcam import LayerGradCam                            # noqa: E402  instead of captum.attr.LayerGradCam
from xai_engine.blur import GaussianBlur                               # noqa: E402  device-side substrate_fn
from xai_engine.zoo import resnet50                                    # noqa: E402

device = "cuda:0"
img_hw, batch_size, steps = 224, 50, 50
model = resnet50(seed=0).to(device).eval()
input_tensor = torch.randn(1, 3, img_hw, img_hw, generator=torch.Generator().manual_seed(1))

target_class = model_utils.getClass(input_tensor, model, device)
print("class", int(target_class), "p =", float(model_utils.getPrediction(input_tensor, model, device, target_class)[0]))

# --- attributions (evaluatePerturbation.py:109-111, :147-153)
ig = attr.IG(input_tensor, model, steps, batch_size, 1, 0, device, target_class)
lig = attr.IG(input_tensor, model, steps, batch_size, .9, 0, device, target_class)
gc = LayerGradCam(model, model.layer4).attribute(input_tensor.to(device), target_class, relu_attributions=True)
saliency = {name: np.abs(np.sum(m.detach().cpu().numpy(), axis=0)) for name, m in (("ig", ig), ("lig", lig))}
print("IG", tuple(ig.shape), "Left-IG", tuple(lig.shape), "Grad-CAM", tuple(gc.shape))

# --- perturbation metrics (evaluatePerturbation.py:456-497)
blur = GaussianBlur(31, 31, device)              # reference: lambda x: conv2d(x, MAS.gkern(31, 31), padding=15)
HW = img_hw * img_hw
for name, attribution in saliency.items():
    _, MAS_ins, _, _, RISE_ins = MAS.MASMetric(model, HW, 'ins', img_hw, substrate_fn=blur).single_run(input_tensor, attribution, device, max_batch_size=batch_size)
    _, MAS_del, _, _, RISE_del = MAS.MASMetric(model, HW, 'del', img_hw, substrate_fn=torch.zeros_like).single_run(input_tensor, attribution, device, max_batch_size=batch_size)
    _, AIC_ins = PIC.AICMetric(model, HW, 'ins', img_hw, substrate_fn=blur).single_run(input_tensor, attribution, device, max_batch_size=batch_size)
    _, MORF = PNP.PositiveNegativePerturbation(model, HW, 'morf', img_hw, substrate_fn=torch.zeros_like).single_run(input_tensor, attribution, device, max_batch_size=batch_size)
    _, MONO_pos = MONO.MonotonicityMetric(model, HW, 'positive', img_hw, substrate_fn=blur).single_run(input_tensor, attribution, device, max_batch_size=batch_size)
    print(f"{name:4s} MAS_ins {MAS.auc(MAS_ins):.4f}  MAS_del {MAS.auc(MAS_del):.4f}  RISE_ins {MAS.auc(RISE_ins):.4f}  "
          f"RISE_del {MAS.auc(RISE_del):.4f}  AIC_ins {MAS.auc(AIC_ins):.4f}  MORF {MAS.auc(MORF):.4f}  MONO_pos {MONO_pos:.4f}")
