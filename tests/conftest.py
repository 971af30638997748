import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
PKG_NAME = "aframe-gaussian-splatting_amd"


def pkg(sub=None):
    """The product package (hyphenated directory name -> importlib)."""
    import importlib
    return importlib.import_module(PKG_NAME + ("." + sub if sub else ""))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def load_case(name, manifest=None):
    """Golden vectors captured from the reference's own JavaScript (oracle/gen_golden.js)."""
    m = manifest or load_manifest()
    d = m[name]
    out = {"meta": d["meta"]}
    if d["arrays"]:
        raw = open(os.path.join(GOLDEN, name + ".bin"), "rb").read()
        for k, a in d["arrays"].items():
            out[k] = np.frombuffer(raw, dtype="<" + a["dtype"], count=a["count"], offset=a["offset"]).copy()
    return out


def cases_of(kind):
    return sorted(k for k, v in load_manifest().items() if v["kind"] == kind)


@pytest.fixture(scope="session")
def manifest():
    return load_manifest()


_ROWS_CACHE = {}


def cached_rows(fn, n, **kw):
    """The benchmark's big synthetic scenes (1 M / 6 M / 20 M rows) are generated once per test process: several tests draw the same
    one, and generating 20 M rows costs more than everything the GPU then does with them.  At most two scenes are kept."""
    key = (fn, int(n), tuple(sorted(kw.items())))
    if key not in _ROWS_CACHE:
        while len(_ROWS_CACHE) >= 2:
            _ROWS_CACHE.pop(next(iter(_ROWS_CACHE)))
        _ROWS_CACHE[key] = getattr(pkg("synth"), fn)(int(n), **kw)
    return _ROWS_CACHE[key]
