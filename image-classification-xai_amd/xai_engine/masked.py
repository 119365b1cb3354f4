"""Score-weighted mask sums of the RISE-family token/feature maskers, served by
K2's weighted form (xai_ig_accum_f32 with step_w1 = scores): one streaming read of the stored masks.

  TIS      saliency = sum_n s_n m_n / sum_n m_n            (reference util/attribution_methods/TIS.py:331-366)
  ViT-CX   sal      = sum_n p_n * m_n / (sum_n m_n) / N    (reference ViT_CX/causal_score.py:54-61, one class row)
"""
import torch

from . import kernels as K


def _weighted_and_plain_sums(masks, scores):
    """masks (N,P) fp32 on the device, scores (N,) -> (sum_n s_n m_n / N, sum_n m_n / N), each (P,)."""
    N, P = masks.shape
    g = masks.contiguous().view(1, N, 1, P)
    ones = torch.ones((1, 1, P), dtype=torch.float32, device=masks.device)
    w = scores.to(masks.device, torch.float32).reshape(1, N).contiguous()
    weighted = K.ig_accum(g, ones, 0.0, w1=w)[0, 0]
    plain = K.ig_accum(g, ones, 0.0)[0, 0]
    return weighted, plain


def tis_saliency(scores, masks, normalise=False):
    """masks: (N, n_tokens) binary token masks, scores: (N,) -> (n_tokens,) coverage-corrected saliency."""
    weighted, plain = _weighted_and_plain_sums(masks.float(), scores)
    sal = weighted / plain
    if normalise:
        sal = sal - sal.min()
        sal = sal / sal.max()
    return sal


def causal_saliency(p_final, masks):
    """p_final: (N,) causal scores of ONE class, masks: (N, H*W) -> (H*W,) = p @ (masks / masks.sum(0)) / N."""
    weighted, plain = _weighted_and_plain_sums(masks.float(), p_final)
    N = masks.shape[0]
    return weighted / (plain * N)
