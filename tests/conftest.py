import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _load_oracle():
    """ORACLE = test infrastructure (oracle/): the CPU restatement of the reference path, bound with the same harness."""
    from pasture_amd._capi import CApi
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return CApi(ctypes.CDLL(path), "orc", product=False)


@pytest.fixture(scope="session")
def oracle():
    return _load_oracle()


@pytest.fixture(scope="session")
def hip():
    """The product: libpasture_amd.so (HIP kernels).  Only meaningful on the GPU box."""
    from pasture_amd import product_api
    return product_api()


@pytest.fixture(scope="session", params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def api(request):
    """Both implementations behind the same Python surface: expectations are computed independently (numpy / literal
    known answers), so the CPU suite pins the oracle and the GPU suite pins the HIP path against the same vectors."""
    if request.param == "oracle":
        return _load_oracle()
    from pasture_amd import product_api
    return product_api()
