"""Reference module path `util.test_methods.PosNegPertFunctions` on the HIP engine
(auc :10, PositiveNegativePerturbation :14 of the reference file)."""
from xai_engine.curves import auc  # noqa: F401
from xai_engine.perturb import PositiveNegativePerturbation  # noqa: F401
