import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU checkers are plain gcc builds; make sure they exist (seconds).
    oracle_so = os.path.join(ROOT, "oracle", "libmifc_oracle.so")
    if not os.path.exists(oracle_so) or (
        os.path.isdir("/root/reference") and not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libmifc_ref.so"))
    ):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    from cpulib import CpuLib

    return CpuLib("oracle")


@pytest.fixture(scope="session")
def ref():
    from cpulib import CpuLib, available

    if not available("ref"):
        pytest.skip("oracle/_ref/libmifc_ref.so not built (needs /root/reference at build time)")
    return CpuLib("ref")


@pytest.fixture(scope="session")
def gpu_ctx():
    import mi_fieldcalc_amd as fc

    ctx = fc.Context(0)  # raises without a GPU: GPU tests must not silently fall back
    yield ctx
    ctx.close()


@pytest.fixture
def mifc_env(gpu_ctx):
    """Sets MIFC_* tuning variables for one test.  The library reads its environment when a
    context is created (never per call), so the session context is told to re-read it; the
    previous values are restored -- and re-read -- afterwards.  mifc_env(name, None) unsets."""
    saved = {}

    def set_var(name, value):
        if name not in saved:
            saved[name] = os.environ.get(name)
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = str(value)
        gpu_ctx.reload_env()

    yield set_var
    for name, value in saved.items():
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = value
    gpu_ctx.reload_env()
