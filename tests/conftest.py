import importlib
import os
import sys

import pytest

# a deployment setting of the host process: the HIP runtime maps the streams of a process onto this many hardware queues (default 4), read when it
# initialises; the library itself no longer touches the environment
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver on the GPU box)")


@pytest.fixture(scope="session")
def O():
    from oracle import oracle

    oracle.build()
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def pkg():
    # torch first: it carries its own copy of the HIP runtime, and whichever copy a process loads first must stay the only one (the
    # library links /opt/rocm's; loading that one before torch's leaves the process with two runtimes and no devices)
    import torch  # noqa: F401
    return importlib.import_module("snark-bn254-verifier_amd")


@pytest.fixture(scope="session")
def fixtures():
    import json

    g = os.path.join(ROOT, "tests", "golden")
    fx = json.load(open(os.path.join(g, "fixtures.json")))
    vk = open(os.path.join(g, "plonk_vk.bin"), "rb").read()
    return fx, vk


@pytest.fixture(scope="session")
def hostsim():
    """The product's device arithmetic compiled for the host with the bound tracker (tests/hostsim)."""
    import ctypes
    import subprocess

    d = os.path.join(ROOT, "tests", "hostsim")
    name = os.environ.get("BN254_HOSTSIM_LIB") or "libhostsim.so"      # tests/test_sanitizers.py: libhostsim_san.so
    subprocess.check_call(["make", "-s", "-C", d, "HSFLAGS=-DHS_WITH_CURVE", name])
    return ctypes.CDLL(os.path.join(d, name))
