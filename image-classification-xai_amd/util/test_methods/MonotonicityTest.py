"""Reference module path `util.test_methods.MonotonicityTest` on the HIP engine
(gkern :10, auc :29, MonotonicityMetric :34 of the reference file)."""
from xai_engine.blur import gkern  # noqa: F401
from xai_engine.curves import auc  # noqa: F401
from xai_engine.perturb import MonotonicityMetric  # noqa: F401
