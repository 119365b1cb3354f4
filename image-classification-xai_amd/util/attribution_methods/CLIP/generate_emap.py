"""Reference module path `util.attribution_methods.CLIP.generate_emap`, RISE part only
(generate_masks :65-81, rise :85-101 of the reference file) on the HIP engine.  The other
explainers of that file (Grad-ECLIP, GAME, M2IB, CLIP-Surgery, LRP) are out of scope."""
from xai_engine.rise import generate_masks, rise  # noqa: F401
