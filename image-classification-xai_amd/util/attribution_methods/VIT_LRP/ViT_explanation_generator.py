"""Reference module path `util.attribution_methods.VIT_LRP.ViT_explanation_generator`
(imported at evaluatePerturbation.py:45): `Baselines` (:135-520 of the reference file -- attention-space
IG, rollouts, transition attention maps, bidirectional attribution) on the HIP engine.  `LRP` and the
helpers of that file are out of scope and are taken, on first use, from the same-named file of the next
`util` on sys.path (xai_engine/_shim.py)."""
from xai_engine._shim import fall_through as _fall_through
from xai_engine.vit_attr import Baselines  # noqa: F401

__getattr__ = _fall_through(__name__, __file__)
