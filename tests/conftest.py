import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ckpt_path():
    return os.path.join(GOLDEN, "model.pth")


@pytest.fixture(scope="session")
def oracle():
    from oracle import sgpr_oracle
    return sgpr_oracle


@pytest.fixture(scope="session")
def oracle_sd(oracle, ckpt_path):
    return oracle.load_checkpoint(ckpt_path)


@pytest.fixture(scope="session")
def release_state_dicts(oracle):
    """{"3_20/00": state dict, ...}: the 18 checkpoints of the reference's model/release_model.zip (copied as a data
    fixture next to the goldens generated from it, tests/golden/make_golden.py)."""
    import io
    import zipfile
    out = {}
    with zipfile.ZipFile(os.path.join(GOLDEN, "release_model.zip")) as z:
        for name in sorted(n for n in z.namelist() if n.endswith("model.pth")):
            out[name.split("/", 1)[1].rsplit("/", 1)[0]] = oracle.load_checkpoint(io.BytesIO(z.read(name)))
    return out
