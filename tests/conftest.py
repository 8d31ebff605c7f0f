import glob
import os
import sys

os.environ.setdefault("MIOPEN_FIND_MODE", "2")  # see py4cast_amd/__init__.py (must be set before the first library conv)

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(pattern="rollout_*.npz"):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, pattern)))


def load_golden(path):
    z = np.load(path, allow_pickle=False)
    meta = eval(str(z["meta"]))  # repr of a plain dict written by make_golden.py
    ins = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    outs = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    return meta, ins, outs


@pytest.fixture(scope="session")
def gpu_device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def diag_library():
    """Tests that flip P4C_* A/B switches (an older kernel as the reference of a newer one, a forced geometry) run on the diagnostic
    build of the library: the product library compiles every switch to its default (csrc/common.hpp::diag_env)."""
    from py4cast_amd import _lib

    if not os.path.exists(_lib.DIAG_LIB_PATH):
        pytest.skip("libpy4cast_hip_diag.so not built (make -C py4cast_amd/csrc diag)")
    with _lib.use_diagnostic_library():
        yield
