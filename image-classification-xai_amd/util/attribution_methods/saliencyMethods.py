"""Reference module path `util.attribution_methods.saliencyMethods`, served by the HIP engine.
Same names, argument order and return conventions as the reference file (see xai_engine/ig.py
for the per-function reference line ranges).  `device` must be a HIP device ('cuda:N')."""
from xai_engine.ig import (  # noqa: F401
    IG, IDG, IDGI, input_grad, getGradientsParallel, getPredictionParallel, getSlopes, getAlphaParameters,
)
from xai_engine.smooth import smoothGrad  # noqa: F401
