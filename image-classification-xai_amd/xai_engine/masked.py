"""Score-weighted mask sums of the RISE-family token/feature maskers, served by K16
(xai_masked_sums_f32): the weighted and the plain sum from ONE streaming read of the stored masks.

  TIS      saliency = sum_n s_n m_n / sum_n m_n            (reference util/attribution_methods/TIS.py:331-366)
  ViT-CX   sal      = sum_n p_n * m_n / (sum_n m_n) / N    (reference ViT_CX/causal_score.py:54-61, one class row)
"""
import torch

from . import kernels as K


def _weighted_and_plain_sums(masks, scores):
    """masks (N,P) fp32 on the device, scores (N,) -> (sum_n s_n m_n / N, sum_n m_n / N), each (P,)."""
    return K.masked_sums(masks.contiguous(), scores.to(masks.device, torch.float32).reshape(-1).contiguous())


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
