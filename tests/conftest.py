import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden():
    from safetensors.torch import load_file
    from safetensors import safe_open
    import json

    def _load(name, with_meta=False):
        path = os.path.join(GOLDEN, name + ".safetensors")
        t = load_file(path)
        if not with_meta:
            return t
        with safe_open(path, "pt") as f:
            meta = {k: json.loads(v) for k, v in (f.metadata() or {}).items()}
        return t, meta
    return _load
