import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the oracle parallelises over sites with OpenMP: on a box whose cgroup grants a 16-core share of a 256-core host the
# default -- one thread per host core -- runs every oracle call an order of magnitude slower than 16 threads do
try:
    _cores = len(os.sched_getaffinity(0))
except (AttributeError, OSError):
    _cores = os.cpu_count() or 1
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, _cores))))
sys.path.insert(0, os.path.join(ROOT, "pll-modules_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# PLLHIP_ORACLE_LIB: another build of the oracle (tests/host_sanitizers.sh: ASan + UBSan)
ORACLE_LIB = os.environ.get("PLLHIP_ORACLE_LIB") or os.path.join(ROOT, "oracle", "_build", "libpll_oracle.so")
PRODUCT_LIB = os.path.join(ROOT, "pll-modules_amd", "libpll_hip.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _make(directory):
    subprocess.run(["make", "-s", "-C", directory], check=True)


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (test infrastructure), built on demand"""
    import pllhip_ctypes as pc
    if not os.path.exists(ORACLE_LIB):
        _make(os.path.join(ROOT, "oracle"))
    return pc.PllLib(ORACLE_LIB)


@pytest.fixture(scope="session")
def product_nogpu():
    """the product library, loaded only (no compute): works without a GPU"""
    import pllhip_ctypes as pc
    if not os.path.exists(PRODUCT_LIB):
        _make(os.path.join(ROOT, "pll-modules_amd"))
    return pc.PllLib(PRODUCT_LIB)


@pytest.fixture(scope="session")
def product(product_nogpu):
    """the product library on a GPU box; fails loudly if there is no device"""
    n = product_nogpu.lib.pllhip_device_count()
    assert n > 0, "no HIP device visible: GPU tests must run on an MI355X box"
    return product_nogpu
