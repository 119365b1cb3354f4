"""Reference module path `util.test_methods.RISETestFunctions` on the HIP engine
(gkern :11, auc :30, RISEMetric :34 of the reference file)."""
from xai_engine.blur import gkern  # noqa: F401
from xai_engine.curves import auc  # noqa: F401
from xai_engine.perturb import RISEMetric  # noqa: F401
