import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_usable():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without an MI355X skips the gpu-marked tests instead of failing in m2v_create
    (the HIP path has no CPU fallback); `-m gpu` on the GPU box runs them."""
    if _gpu_usable():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no HIP device here; the product has no CPU fallback)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_assertrepr_compare(config, op, left, right):
    """Two long byte strings that differ (a stream against the oracle's): say where, instead of pytest's character diff, which takes many
    minutes for a megabyte - a failing parity test must fail at once, not look like a hang (seen: ten minutes on the GPU box)."""
    if op == "==" and isinstance(left, (bytes, bytearray)) and isinstance(right, (bytes, bytearray)) and max(len(left), len(right)) > 4096:
        import numpy as np
        a, b = np.frombuffer(bytes(left), np.uint8), np.frombuffer(bytes(right), np.uint8)
        n = min(a.size, b.size)
        d = np.flatnonzero(a[:n] != b[:n])
        first = int(d[0]) if d.size else n
        return ["byte strings differ: %d bytes against %d, first difference at byte %d, %d of the first %d bytes differ" % (a.size, b.size, first, d.size, n),
                "left  [%d:%d] = %s" % (first, first + 16, bytes(left[first:first + 16]).hex()),
                "right [%d:%d] = %s" % (first, first + 16, bytes(right[first:first + 16]).hex())]
    return None
