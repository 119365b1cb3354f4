"""Mirror of `util.test_methods`; modules not held here (PIC, sanity, ...) come from the next `util` on sys.path."""
from xai_engine._shim import extend as _extend

__path__ = _extend(__path__, __name__)
