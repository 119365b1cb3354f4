"""Reference module path `util.attribution_methods.TIS` on the HIP engine (class TIS :14-365)."""
from xai_engine.tis import TIS  # noqa: F401
