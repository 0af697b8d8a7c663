import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
REFERENCE = os.environ.get("ILQR_REFERENCE", "/root/reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                  if not p.endswith("units.npz") and not os.path.basename(p).startswith(("user_", "quat_utils", "car_population", "car_n200_population", "acrobot_population", "quad_ea_population", "quad_quat_population", "eig_decisions")))


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/), built on demand with gcc."""
    from oracle import c_oracle
    c_oracle.build()
    c_oracle.lib()
    return c_oracle


@pytest.fixture(scope="session")
def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "control"))
