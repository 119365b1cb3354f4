"""Reference module path `util.attribution_methods.ViT_CX.ViT_CX` on the HIP engine
(ViT_CX :61-117, get_cos_similar_matrix :22-28, norm_matrix :29-34, reshape_function_vit :41-46)."""
from xai_engine.vit_cx import ViT_CX, get_cos_similar_matrix, norm_matrix, reshape_function_vit  # noqa: F401
