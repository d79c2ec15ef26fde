import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.dont_write_bytecode = False

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_cmdline_main(config):
    """CPU runs (-m "not gpu") execute the kernel sources on the SIMT interpreter — one slow OS thread per test — so they are spread
    over worker processes (pytest-xdist) unless the caller chose a number himself (-n) or DFX_TEST_WORKERS=0.  GPU runs (-m gpu) stay in
    one process: the engine's persistent launches own the device."""
    expr = getattr(config.option, "markexpr", "") or ""
    if "not gpu" not in expr or hasattr(config, "workerinput") or getattr(config.option, "numprocesses", None) is not None:
        return None
    try:
        import xdist  # noqa: F401
    except ImportError:
        return None
    env = os.environ.get("DFX_TEST_WORKERS")
    n = int(env) if env is not None else min(6, max(1, (os.cpu_count() or 2) - 1))
    if n > 1 and hasattr(config.option, "numprocesses"):
        # build the shared native artefacts once, before the workers start (they would race for the same output files)
        from tests.hipemu.build_emu import build as emu_build

        emu_build()
        from oracle import libdf_oracle

        libdf_oracle.build()
        config.option.numprocesses = n   # (xdist's own pytest_cmdline_main has already run: what it derives from -n is set here too)
        config.option.dist = "load"
        config.option.tx = ["popen"] * n
        os.environ["DFX_TEST_WORKER_THREADS"] = str(max(1, (os.cpu_count() or n) // n))
    return None


def pytest_configure(config):
    if os.environ.get("PYTEST_XDIST_WORKER") and os.environ.get("DFX_TEST_WORKER_THREADS"):
        import torch

        torch.set_num_threads(int(os.environ["DFX_TEST_WORKER_THREADS"]))   # the torch oracle would otherwise oversubscribe the cores
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
