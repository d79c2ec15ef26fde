import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.dont_write_bytecode = False

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: imports /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    from tools.ref_import import reference_available

    have_ref = reference_available()
    for item in items:
        if "needs_reference" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def _use_backend(name):
    from deepfilternet_amd import _lib

    if name == "emu":
        from tests.hipemu.build_emu import build

        _lib.use_library(build())
        assert _lib.is_emulator()
    else:
        from deepfilternet_amd.build import build

        _lib.use_library(build())
        assert not _lib.is_emulator(), "the GPU tests must run on the HIP build"
        import torch

        assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return name


@pytest.fixture(params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    """'emu': the kernel sources run on the CPU SIMT interpreter (tests/hipemu) — logic check without a GPU.
    'hip': the real libdfx.so on an MI355X (-m gpu)."""
    return _use_backend(request.param)


@pytest.fixture
def hip_backend():
    return _use_backend("hip")
