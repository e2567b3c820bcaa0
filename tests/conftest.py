import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "libsgmse_emu.so")
HIP_LIB = os.path.join(ROOT, "sgmse_amd", "libsgmse_hip.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test (emulator, full-width network); set SGMSE_SLOW=1")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_l2(a, b):
    import torch
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return float((a - b).abs().pow(2).sum().sqrt() / b.abs().pow(2).sum().sqrt().clamp_min(1e-30))


def _have_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def emu():
    """The CPU workgroup-emulator build of the kernel sources (test infrastructure, tests/emu/).  CPU tests use it to
    check kernel logic without a GPU; it is loaded explicitly here and never by the package itself."""
    from sgmse_amd import _lib
    if _lib._lib is not None and not _lib.is_emulator():
        pytest.skip("the HIP library is already loaded in this process")
    if not os.path.exists(EMU_LIB) or os.path.getmtime(EMU_LIB) < max(
            os.path.getmtime(os.path.join(ROOT, "sgmse_amd", "csrc", f)) for f in os.listdir(os.path.join(ROOT, "sgmse_amd", "csrc"))):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "sgmse_amd", "csrc"), "emu"])
    _lib.load_library(EMU_LIB)
    assert _lib.is_emulator()
    return "cpu"


@pytest.fixture(scope="session")
def hip():
    """The product library on a real GPU.  Fails (does not skip) when the extension is missing on a GPU box."""
    from sgmse_amd import _lib
    if not _have_gpu():
        pytest.skip("no GPU visible")
    assert os.path.exists(HIP_LIB), "libsgmse_hip.so missing: run __graft_entry__.build() before the GPU tests"
    _lib.load_library(HIP_LIB)
    assert _lib.backend() == "hip-gfx950"
    return "cuda"
