"""The `util/` mirror must coexist with the reference's own `util` tree: with the build first
on sys.path and another root holding a full `util` tree after it, the import block of
XAI_Survey/evaluations/evaluatePerturbation.py:17-59 -- restated here against a FAKE sibling tree written into
tmp_path (stub modules, nothing copied from the reference) -- resolves the hot-path names to xai_engine and
every other name to the sibling tree."""
import os
import subprocess
import sys
import textwrap

from conftest import PKG

# the sibling tree: same package shapes as the reference (regular packages with __init__.py; CLIP is a namespace
# directory there, util/attribution_methods/CLIP has no __init__.py) and stubs for the names the harness imports
FAKE = {
    "util/__init__.py": "",
    "util/model_utils.py": "WHO = 'sibling'\n",
    "util/visualization.py": "WHO = 'sibling'\n",
    "util/modified_models/__init__.py": "",
    "util/modified_models/resnet.py": "WHO = 'sibling'\ndef resnet50(): return 'sibling resnet50'\n",
    "util/attribution_methods/__init__.py": "",
    "util/attribution_methods/saliencyMethods.py": "WHO = 'sibling'\n",
    "util/attribution_methods/AGI.py": "WHO = 'sibling'\n",
    "util/attribution_methods/GIGBuilder.py": "WHO = 'sibling'\n",
    "util/attribution_methods/XRAIBuilder.py": "WHO = 'sibling'\n",
    "util/attribution_methods/MDAFunctions.py": "WHO = 'sibling'\n",
    "util/attribution_methods/TIS.py": "WHO = 'sibling'\n",
    "util/attribution_methods/lime/__init__.py": "",
    "util/attribution_methods/lime/limeAttr.py": "WHO = 'sibling'\n",
    "util/attribution_methods/VIT_LRP/__init__.py": "",
    "util/attribution_methods/VIT_LRP/ViT_new_timm.py": "def vit_base_patch16_224(): return 'sibling vit16'\ndef vit_base_patch32_224(): return 'sibling vit32'\n",
    "util/attribution_methods/VIT_LRP/ViT_LRP_timm.py": "def vit_base_patch16_224(): return 'sibling lrp16'\ndef vit_base_patch32_224(): return 'sibling lrp32'\n",
    "util/attribution_methods/VIT_LRP/ViT_explanation_generator.py":
        "from .ViT_new_timm import vit_base_patch16_224\nclass LRP: WHO = 'sibling'\nclass Baselines: WHO = 'sibling'\n",
    "util/attribution_methods/ViT_CX/__init__.py": "",
    "util/attribution_methods/ViT_CX/ViT_CX.py": "WHO = 'sibling'\n",
    "util/attribution_methods/ViT_CX/get_feature_map.py": "WHO = 'sibling'\n",
    # CLIP: namespace directory (no __init__.py), relative imports inside generate_emap as in the reference
    "util/attribution_methods/CLIP/Game_MM_CLIP/__init__.py": "",
    "util/attribution_methods/CLIP/Game_MM_CLIP/clip.py": "WHO = 'sibling mm_clip'\n",
    "util/attribution_methods/CLIP/CLIP_Surgery/__init__.py": "",
    "util/attribution_methods/CLIP/CLIP_Surgery/clip.py": "WHO = 'sibling surgery'\n",
    "util/attribution_methods/CLIP/generate_emap.py": textwrap.dedent("""
        from .Game_MM_CLIP import clip as mm_clip
        from .CLIP_Surgery import clip as surgery_clip
        LOADS = []
        LOADS.append(1)
        def imgprocess_keepsize(img): return ('sibling imgprocess_keepsize', mm_clip.WHO)
        def mm_interpret(): return 'sibling mm_interpret'
        def clip_encode_dense(): return 'sibling'
        def grad_eclip(): return 'sibling'
        def mask_clip(): return 'sibling'
        def compute_rollout_attention(): return 'sibling'
        def clip_surgery_map(): return surgery_clip.WHO
        def m2ib_clip_map(): return 'sibling'
        def clip_lrp(): return 'sibling'
        def rise(): return 'sibling rise'
        def generate_masks(): return 'sibling generate_masks'
    """),
    "util/test_methods/__init__.py": "",
    "util/test_methods/MASTestFunctions.py": "WHO = 'sibling'\ndef pgd_attack(): return 'sibling pgd'\n",
    "util/test_methods/PICTestFunctions.py": "WHO = 'sibling'\n",
    "util/test_methods/sanityForMethods.py": "WHO = 'sibling'\n",
}

# evaluatePerturbation.py:17-59, restated (third-party imports left out)
HARNESS_IMPORTS = textwrap.dedent("""
    import sys
    sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[1])          # build first, sibling tree after it
    from util import model_utils                                                                   # :17
    from util.modified_models import resnet                                                        # :21
    from util.attribution_methods.CLIP.Game_MM_CLIP import clip as mm_clip                         # :24
    from util.attribution_methods.CLIP.CLIP_Surgery import clip as surgery_clip                    # :25
    from util.attribution_methods.VIT_LRP.ViT_new_timm import vit_base_patch16_224                 # :31
    from util.attribution_methods.VIT_LRP.ViT_LRP_timm import vit_base_patch16_224 as vit_LRP      # :32
    from util.attribution_methods import saliencyMethods as attr                                   # :39
    from util.attribution_methods.lime import limeAttr                                             # :40
    from util.attribution_methods import GIGBuilder as GIG_Builder                                 # :41
    from util.attribution_methods import AGI as AGI                                                # :42
    from util.attribution_methods import XRAIBuilder as XRAI                                       # :44
    from util.attribution_methods.VIT_LRP.ViT_explanation_generator import Baselines, LRP          # :45
    from util.attribution_methods.ViT_CX.ViT_CX import ViT_CX                                      # :46
    from util.attribution_methods.TIS import TIS                                                   # :47
    from util.attribution_methods import MDAFunctions                                              # :48
    from util.attribution_methods.CLIP.generate_emap import imgprocess_keepsize, mm_interpret, \\
        clip_encode_dense, grad_eclip, mask_clip, compute_rollout_attention, \\
        clip_surgery_map, m2ib_clip_map, clip_lrp                                                  # :50-52
    from util.test_methods import MASTestFunctions as MAS                                          # :55
    from util.test_methods import AICTestFunctions as PIC                                          # :56
    from util.test_methods import MonotonicityTest as MONO                                         # :57
    from util.test_methods import PosNegPertFunctions as PNP                                       # :58
    from util.attribution_methods.CLIP.generate_emap import rise, generate_masks                   # CLIP_example.ipynb cell 0
    from util.test_methods import PICTestFunctions, sanityForMethods                               # evaluatePIC.py / evaluateSanity.py
    from util.attribution_methods.ViT_CX import get_feature_map
    from util import visualization

    import xai_engine.ig, xai_engine.perturb, xai_engine.rise, xai_engine.vit_attr, xai_engine.vit_cx, xai_engine.tis, xai_engine.smooth
    # hot path -> the HIP engine
    assert attr.IG is xai_engine.ig.IG and attr.IDG is xai_engine.ig.IDG and attr.smoothGrad is xai_engine.smooth.smoothGrad
    assert not hasattr(attr, "WHO") and not hasattr(model_utils, "WHO")
    assert MAS.MASMetric is xai_engine.perturb.MASMetric and PIC.AICMetric is xai_engine.perturb.AICMetric
    assert MONO.MonotonicityMetric is xai_engine.perturb.MonotonicityMetric
    assert PNP.PositiveNegativePerturbation is xai_engine.perturb.PositiveNegativePerturbation
    assert rise is xai_engine.rise.rise and generate_masks is xai_engine.rise.generate_masks
    assert Baselines is xai_engine.vit_attr.Baselines and ViT_CX is xai_engine.vit_cx.ViT_CX and TIS is xai_engine.tis.TIS
    # everything else -> the sibling tree
    assert resnet.resnet50() == "sibling resnet50" and vit_base_patch16_224() == "sibling vit16" and vit_LRP() == "sibling lrp16"
    assert mm_clip.WHO == "sibling mm_clip" and surgery_clip.WHO == "sibling surgery"
    for m in (limeAttr, GIG_Builder, AGI, XRAI, MDAFunctions, PICTestFunctions, sanityForMethods, get_feature_map, visualization):
        assert m.WHO == "sibling", m
    assert LRP.WHO == "sibling"
    assert imgprocess_keepsize(None) == ("sibling imgprocess_keepsize", "sibling mm_clip") and mm_interpret() == "sibling mm_interpret"
    assert clip_surgery_map() == "sibling surgery" and clip_lrp() == "sibling"
    assert MAS.pgd_attack() == "sibling pgd"
    import util.attribution_methods.CLIP.generate_emap as ge
    assert sys.modules["util.attribution_methods.CLIP.generate_emap__next"].LOADS == [1]        # loaded once, lazily
    try:
        ge.no_such_name
    except AttributeError as e:
        assert "no_such_name" in str(e)
    else:
        raise AssertionError("expected AttributeError")
    print("harness imports ok")
""")

ALONE = textwrap.dedent("""
    import sys
    sys.path.insert(0, sys.argv[1])
    from util.attribution_methods.CLIP.generate_emap import rise
    import util.attribution_methods.CLIP.generate_emap as ge
    try:
        ge.mm_interpret
    except AttributeError as e:
        assert "follows it on sys.path" in str(e), e
    else:
        raise AssertionError("expected AttributeError")
    try:
        from util.attribution_methods import AGI
    except ImportError:
        pass
    else:
        raise AssertionError("expected ImportError")
    print("alone ok")
""")


def _run(code, *args):
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    return subprocess.run([sys.executable, "-c", code, *args], capture_output=True, text=True, env=env, timeout=300)


def test_harness_import_block_resolves_against_a_sibling_util_tree(tmp_path):
    for rel, body in FAKE.items():
        p = tmp_path / "sibling" / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(body)
    r = _run(HARNESS_IMPORTS, PKG, str(tmp_path / "sibling"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "harness imports ok" in r.stdout


def test_mirror_alone_still_imports_and_says_what_is_missing():
    r = _run(ALONE, PKG)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "alone ok" in r.stdout


CAPTUM = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[1])          # build first, then the tree holding a (stub) captum
    import captum.attr
    theirs = captum.attr.LayerGradCam
    from util import model_utils                                               # the harness's first util import (:17)
    from captum.attr import GuidedBackprop, LayerGradCam                       # :43
    import xai_engine.gradcam as gc
    if os.environ.get("XAI_PATCH_CAPTUM") == "1":
        assert LayerGradCam is gc.LayerGradCam and GuidedBackprop.WHO == "captum"
        assert gc.patch_captum() is gc.LayerGradCam                            # idempotent
    else:
        assert LayerGradCam is theirs and LayerGradCam.WHO == "captum"         # not asked: captum untouched
        assert gc.patch_captum() is theirs
        from captum.attr import LayerGradCam as again
        assert again is gc.LayerGradCam
    print("captum ok")
""")


def test_captum_LayerGradCam_is_rebound_only_when_asked(tmp_path):
    """evaluatePerturbation.py:43 imports LayerGradCam from captum, the one import of the path outside `util.*`: the overlay is
    opt-in (XAI_PATCH_CAPTUM=1 or xai_engine.gradcam.patch_captum()) and touches that one name only.  Stub captum in tmp_path."""
    pkg = tmp_path / "site" / "captum" / "attr"
    pkg.mkdir(parents=True)
    (tmp_path / "site" / "captum" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("class LayerGradCam: WHO = 'captum'\nclass GuidedBackprop: WHO = 'captum'\n")
    for flag in ("0", "1"):
        env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
        env["XAI_PATCH_CAPTUM"] = flag
        r = subprocess.run([sys.executable, "-c", CAPTUM, PKG, str(tmp_path / "site")], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "captum ok" in r.stdout
    # without captum installed the call is a no-op
    r = _run("import sys; sys.path.insert(0, sys.argv[1]); import xai_engine.gradcam as g; assert g.patch_captum() is None; print('none ok')", PKG)
    assert r.returncode == 0 and "none ok" in r.stdout, r.stdout + r.stderr


REAL_TREE = r'''
import sys, types, importlib, importlib.abc, importlib.machinery, importlib.util
sys.dont_write_bytecode = True
BUILD, REF = sys.argv[1], sys.argv[2]
sys.path.insert(0, REF); sys.path.insert(0, BUILD)

class _Anything(types.ModuleType):
    """inert placeholder for a third-party module that is not installed: attributes are placeholders, calls return one"""
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        m = sys.modules.get(full)
        if m is None:
            m = _Anything(full); m.__path__ = []
            sys.modules[full] = m
        return m
    def __call__(self, *a, **k):
        return _Anything(self.__name__ + "()")
    def __mro_entries__(self, bases):
        return (object,)

MISSING = ("torchvision", "clip", "captum", "timm", "cv2", "skimage", "ttach", "fast_pytorch_kmeans", "h5py", "ftfy", "kornia")
class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in MISSING:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
    def create_module(self, spec):
        m = _Anything(spec.name); m.__path__ = []
        return m
    def exec_module(self, module):
        pass
from transformers import CLIPTokenizerFast  # noqa: F401  (installed; must see that torchvision is absent before the placeholders go in)
sys.meta_path.insert(0, Finder())
cv = types.ModuleType("cvxopt"); cv.matrix = lambda *a, **k: None
cv.solvers = types.SimpleNamespace(options={}, qp=None); sys.modules["cvxopt"] = cv

import re
EVAL = REF + "/XAI_Survey/evaluations"
sys.path.insert(2, EVAL)                                       # the scripts run from their own directory (`from utils.metrices import *`)
MISSING_EXTRA = ("matplotlib",) if importlib.util.find_spec("matplotlib") is None else ()
import xai_engine.ig, xai_engine.perturb, xai_engine.vit_attr, xai_engine.vit_cx, xai_engine.tis, xai_engine.smooth
for script in ("evaluatePerturbation.py", "evaluateSanity.py", "evaluateImageNetSeg.py", "qualitativeGeneration.py"):
    src = open(EVAL + "/" + script).read().split("\n")
    n = next(i for i, l in enumerate(src) if re.match(r"^(def |class |with open|if __name__)", l))
    block = "\n".join(src[:n]).replace("os.sys.path.append(os.path.dirname(os.path.abspath('..')))", "pass")
    ns = {}
    exec(compile(block, f"{script}[1:{n}]", "exec"), ns)
    assert ns["attr"].IG is xai_engine.ig.IG and ns["attr"].smoothGrad is xai_engine.smooth.smoothGrad, script
    assert ns["MAS"].MASMetric is xai_engine.perturb.MASMetric, script
    if "PIC" in ns:
        assert ns["PIC"].AICMetric is xai_engine.perturb.AICMetric and ns["MONO"].MonotonicityMetric is xai_engine.perturb.MonotonicityMetric
        assert ns["PNP"].PositiveNegativePerturbation is xai_engine.perturb.PositiveNegativePerturbation
    assert ns["Baselines"] is xai_engine.vit_attr.Baselines and ns["ViT_CX"] is xai_engine.vit_cx.ViT_CX and ns["TIS"] is xai_engine.tis.TIS, script
    names = [k for k in ("resnet", "AGI", "GIG_Builder", "XRAI", "MDAFunctions", "limeAttr", "LRP", "mm_interpret", "imgprocess_keepsize", "clip_lrp", "MASCalibrate",
                         "attr_to_subplot") if k in ns]
    for k in names:
        f = getattr(ns[k], "__file__", None) or sys.modules[ns[k].__module__].__file__
        assert f.startswith(REF), (script, k, f)
    print(f"{script}: lines 1-{n} executed; from the reference tree: {', '.join(names)}")
print("real tree ok")
'''


def test_the_reference_harness_own_import_block_runs_against_the_mirror(tmp_path):
    """Where the reference tree is present (this container: /root/reference; not on the GPU box -> skipped): the import blocks
    of all four harness scripts of XAI_Survey/evaluations (evaluatePerturbation.py:1-59, evaluateSanity.py, evaluateImageNetSeg.py,
    qualitativeGeneration.py), read from the reference and executed as they are with sys.path = [build, reference].  Third-party packages that are not installed (torchvision, clip, captum, timm, cv2, skimage ...) are inert
    placeholders.  Hot-path names must come from xai_engine, everything else from files under the reference root."""
    import pytest
    ref = "/root/reference"
    if not os.path.isfile(os.path.join(ref, "XAI_Survey", "evaluations", "evaluatePerturbation.py")):
        pytest.skip("reference tree not present")
    script = tmp_path / "real_tree.py"
    script.write_text(REAL_TREE)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, str(script), PKG, ref], capture_output=True, text=True, env=env, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "real tree ok" in r.stdout
    for script in ("evaluatePerturbation.py", "evaluateSanity.py", "evaluateImageNetSeg.py", "qualitativeGeneration.py"):
        assert script + ": lines 1-" in r.stdout, r.stdout
