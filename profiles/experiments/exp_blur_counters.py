#!/usr/bin/env python3
"""K7 under rocprofv3 --pmc: 10 launches of the 32-image 31-tap blur (and 10 of the one-image case) so that SQ counters
can say what the waves wait for.  usage: rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d DIR -- python3 exp_blur_counters.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-classification-xai_amd"))
import torch  # noqa: E402
from xai_engine import kernels as K  # noqa: E402
from xai_engine.blur import gkern1d  # noqa: E402
x = torch.randn(32, 3, 224, 224, device="cuda:0")
k = gkern1d(31, 31).to("cuda:0")
for _ in range(10):
    K.blur_sep(x, k)
    K.blur_sep(x[:1], k)
torch.cuda.synchronize()
