"""Import the reference's own modules in the build container (TEST INFRASTRUCTURE).

Usage (only from tests/golden/make_golden.py and oracle pinning tests, and only when
/root/reference exists — it does not exist on the GPU box):

    from oracle.refharness.load_reference import load
    ref = load()          # ref.fcmae, ref.MODALITIES, ref.custom_loss, ref.helpers
"""
import importlib
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "fcmae.py"))


def load():
    if not available():
        raise RuntimeError("reference tree not present (expected in the build container only)")
    sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, "me_emulator"), os.path.join(here, "stubs"), "/root"):
        if p not in sys.path:
            sys.path.insert(0, p)
    ns = types.SimpleNamespace()
    ns.fcmae = importlib.import_module("reference.models.fcmae")
    ns.MODALITIES = importlib.import_module("reference.MODALITIES")
    ns.custom_loss = importlib.import_module("reference.custom_loss")
    # helpers.py uses top-level imports (helpers.py:16-29)
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)
    try:
        ns.helpers = importlib.import_module("reference.helpers")
    except Exception as e:  # pragma: no cover - informational
        ns.helpers = None
        ns.helpers_error = e
    return ns
