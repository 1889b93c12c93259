import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
GOLD = ROOT / "tests" / "golden"
for p in (str(ROOT), str(ROOT / "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    return json.loads((GOLD / "manifest.json").read_text())


@pytest.fixture(scope="session")
def schema_real():
    return json.loads((GOLD / "schema_real.json").read_text())


@pytest.fixture(scope="session")
def schema_syn():
    return json.loads((GOLD / "schema_synthetic.json").read_text())


def load_case(name):
    z = np.load(GOLD / f"case_{name}.npz")
    return {k: z[k] for k in z.files}


def split_prefix(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}
