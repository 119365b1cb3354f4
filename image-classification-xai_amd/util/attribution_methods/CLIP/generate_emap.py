"""Reference module path `util.attribution_methods.CLIP.generate_emap`, RISE part only
(generate_masks :65-81, rise :85-101 of the reference file) on the HIP engine.  The other
explainers of that file (Grad-ECLIP, GAME, M2IB, CLIP-Surgery, LRP: imgprocess_keepsize, mm_interpret,
clip_encode_dense, grad_eclip, mask_clip, ...; imported by name at evaluatePerturbation.py:51-53) are out
of scope: asking this module for one of them loads the same-named file of the next `util` on sys.path
on first use (xai_engine/_shim.py) and hands its attribute over."""
from xai_engine._shim import fall_through as _fall_through
from xai_engine.rise import generate_masks, rise  # noqa: F401

__getattr__ = _fall_through(__name__, __file__)
