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


def use_tuned_miopen_db(rank=0, src=None):
    """Point MIOpen at a private copy of the shipped user find-db (`image-classification-xai_amd/miopen_db`:
    the result of MIOpen's own exhaustive find for the benchmark shapes, recorded once on an MI355X; keyed by
    MIOpen version and problem shape) and return True, so that `torch.backends.cudnn.benchmark = True` costs no
    search time.  One copy per rank: MIOpen locks the files.  Must run before the first convolution."""
    import os
    import shutil
    import tempfile
    src = src or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miopen_db")
    if not os.path.isdir(src) or not os.listdir(src):
        return False
    dst = os.path.join(tempfile.gettempdir(), f"xai_miopen_db_{os.getuid()}_{rank}")
    try:
        shutil.rmtree(dst, ignore_errors=True)
        shutil.copytree(src, dst)
    except OSError:
        return False                 # no writable scratch: stay in immediate mode
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return True
