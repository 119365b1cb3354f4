"""Reference module path `util.attribution_methods.ViT_CX.causal_score` on the HIP engine (causal_score :9-61)."""
from xai_engine.vit_cx import causal_score  # noqa: F401
