import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "image-classification-xai_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
BAR = 1e-5          # BASELINE.json north_star: saliency maps and AUC within 1e-5 (relative, fp32)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Parity tests run the classifier in MIOpen's immediate mode with deterministic solvers only (SURVEY 7 "hard parts";
    # XAI_TEST_DETERMINISTIC=0 lifts it, used by tests/parity_report.sh to show whether any number moves).
    import torch
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = os.environ.get("XAI_TEST_DETERMINISTIC", "1") != "0"


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_inf(a, b):
    """||a-b||_inf / ||b||_inf -- the parity norm of BASELINE.json (1e-5 relative fp32)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    return float(np.abs(a - b).max() / (den if den > 0 else 1.0))


# ---- measured-error ledger -------------------------------------------------------------------
# Every comparison against a reference-made golden vector (and the oracle comparisons beside them) goes through
# `check`, which records the measured error before asserting.  With XAI_PARITY_REPORT=<file> the ledger is written as
# JSON at session end: profiles/r02_parity.json is that file from the GPU box, and every `tol` below must be
# <= max(BAR, 2 x the error recorded there) (tests/test_cpu_host.py::test_tolerances_are_tied_to_measured_errors).
_LEDGER = []


def check(name, got, want, tol, against="golden", absolute=False):
    """assert ||got-want||_inf / ||want||_inf <= tol (absolute=True: max |got-want|), recording the measurement."""
    if absolute:
        err = float(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)).max())
    else:
        err = rel_inf(got, want)
    _LEDGER.append({"name": name, "against": against, "norm": "abs" if absolute else "rel_inf", "measured": err, "tol": tol})
    assert err <= tol, (name, against, err, tol)
    return err


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get("XAI_PARITY_REPORT")
    if path and _LEDGER:
        import torch
        meta = {"deterministic": bool(torch.backends.cudnn.deterministic), "benchmark": bool(torch.backends.cudnn.benchmark),
                "device": torch.cuda.get_device_name(0) if torch.cuda.is_available() else "cpu", "torch": torch.__version__,
                "exitstatus": int(exitstatus)}
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"meta": meta, "comparisons": _LEDGER}, f, indent=1)


@pytest.fixture(scope="session")
def golden():
    return load_golden
