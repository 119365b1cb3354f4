"""Mirror of `util.attribution_methods.VIT_LRP`; the model zoo files (ViT_new_timm, ViT_LRP_timm, ViT_ig, util/)
come from the next `util` on sys.path."""
from xai_engine._shim import extend as _extend

__path__ = _extend(__path__, __name__)
