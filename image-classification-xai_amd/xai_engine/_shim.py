"""Helpers that let the `util/` mirror next to this package COEXIST with the reference's own
`util` tree when both are on sys.path (build first, reference after).

The reference harness (XAI_Survey/evaluations/evaluatePerturbation.py:17-59) imports ~25 `util.*`
modules; only the hot path is served from here.  Two mechanisms make the rest resolve from the
later sys.path entry, with the reference tree untouched and nothing copied:

* packages: every package of the mirror sets ``__path__ = extend(__path__, __name__)``
  (pkgutil.extend_path): sub-modules this mirror does not hold (util.modified_models,
  util.attribution_methods.AGI, ...CLIP.Game_MM_CLIP, ...) are then found in the same-named
  directory of the next `util` on the path;
* modules: a mirrored module that serves only PART of the reference file
  (CLIP/generate_emap.py: rise / generate_masks; VIT_LRP/ViT_explanation_generator.py: Baselines)
  installs ``__getattr__ = fall_through(__name__, __file__)`` (PEP 562): a name it does not define
  is fetched from the same-named file found further along the parent package's ``__path__``, which
  is imported on first use under ``<name>__next`` (so its relative imports keep working).
"""
import importlib
import importlib.util
import os
import pkgutil
import sys


def extend(path, name):
    """``__path__`` of a mirror package + the same-named directories of later sys.path entries."""
    return pkgutil.extend_path(path, name)


def next_file(module_name, own_file):
    """Path of the first `<leaf>.py` (or `<leaf>/__init__.py`) after `own_file`'s directory on the
    parent package's __path__, or None."""
    parent_name, _, leaf = module_name.rpartition(".")
    parent = sys.modules.get(parent_name) or importlib.import_module(parent_name)
    own_dir = os.path.realpath(os.path.dirname(own_file))
    for d in list(getattr(parent, "__path__", [])):
        if os.path.realpath(d) == own_dir:
            continue
        for cand in (os.path.join(d, leaf + ".py"), os.path.join(d, leaf, "__init__.py")):
            if os.path.isfile(cand):
                return cand
    return None


def fall_through(module_name, own_file):
    """Module-level ``__getattr__`` that defers unknown names to the next same-named file."""
    state = {}

    def __getattr__(attr):
        if attr.startswith("__") and attr.endswith("__"):
            raise AttributeError(attr)
        if "mod" not in state:
            path = next_file(module_name, own_file)
            if path is None:
                raise AttributeError(
                    f"module {module_name!r} (HIP engine mirror) has no attribute {attr!r}, and no other "
                    f"`{module_name.replace('.', '/')}.py` follows it on sys.path to take it from")
            alias = module_name + "__next"
            spec = importlib.util.spec_from_file_location(alias, path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[alias] = mod
            try:
                spec.loader.exec_module(mod)
            except BaseException:
                sys.modules.pop(alias, None)
                raise
            state["mod"] = mod
        try:
            return getattr(state["mod"], attr)
        except AttributeError:
            raise AttributeError(f"neither the HIP engine mirror of {module_name!r} nor {state['mod'].__file__} defines {attr!r}") from None

    return __getattr__
