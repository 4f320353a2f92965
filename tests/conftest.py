import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def state_from_golden(g, prefix="sd0/"):
    import torch

    st = {k[len(prefix):]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix)}
    if "quantizer.0.impl.weight" in st:
        st["quantizer.0.impl.embedding.weight"] = st["quantizer.0.impl.weight"]
    return st


def meta_of(g):
    import json

    return json.loads(bytes(g["meta"]).decode())
