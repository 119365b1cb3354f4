"""Mirror of `util.attribution_methods.ViT_CX`; get_feature_map / base_cam / utils come from the next `util` on sys.path."""
from xai_engine._shim import extend as _extend

__path__ = _extend(__path__, __name__)
