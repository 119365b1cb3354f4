"""Mirror of `util.attribution_methods.CLIP` (a namespace directory in the reference); the vendored CLIP
variants (Game_MM_CLIP, CLIP_Surgery, CLIP_lrp, M2IB) come from the next `util` on sys.path."""
from xai_engine._shim import extend as _extend

__path__ = _extend(__path__, __name__)
