"""Drop-in mirror of the reference's `util` package for the accelerated hot path only
(attribution_methods.saliencyMethods, attribution_methods.CLIP.generate_emap [RISE part],
attribution_methods.{TIS,ViT_CX,VIT_LRP.ViT_explanation_generator}, test_methods.*, model_utils).
Put `image-classification-xai_amd/` first on sys.path, the reference root after it: the harness
imports of the hot path resolve here, every other `util.*` module resolves to the reference tree
(xai_engine/_shim.py)."""
from xai_engine._shim import extend as _extend

__path__ = _extend(__path__, __name__)

import os as _os

if _os.environ.get("XAI_PATCH_CAPTUM") == "1":      # opt-in, see xai_engine.gradcam.patch_captum
    from xai_engine.gradcam import patch_captum as _patch_captum
    _patch_captum()
