"""Optional classifier preparation for throughput runs (never applied implicitly).

`fold_batchnorm(model)`: eval-mode Conv2d -> BatchNorm2d pairs are folded into one convolution
(w' = w * gamma/sqrt(var+eps), b' = beta + (b - mean) * gamma/sqrt(var+eps)) by torch.fx.  It is an
exact algebraic identity, but it changes fp32 rounding inside the classifier (~1e-6 relative on
logits), which for ReLU networks moves individual gates -- so it is opt-in and reported
separately (bench.py --fold-bn 1); parity tests always run the classifier as given.
"""
import copy



def fold_batchnorm(model):
    from torch.fx.experimental.optimization import fuse
    m = copy.deepcopy(model).eval()
    fused = fuse(m)
    for p in fused.parameters():
        p.requires_grad_(False)
    return fused
