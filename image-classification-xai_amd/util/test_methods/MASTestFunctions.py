"""Reference module path `util.test_methods.MASTestFunctions` on the HIP engine
(gkern :11, auc :30, MASMetric :55 of the reference file)."""
from xai_engine.blur import gkern  # noqa: F401
from xai_engine.curves import auc  # noqa: F401
from xai_engine.perturb import MASMetric  # noqa: F401
