import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fast-lio-sam-qn_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    # torch bundles its own HIP runtime (same soname as the system one libqn_engine.so links against); whichever is loaded first
    # serves the whole process, and torch only finds its GPUs on its own copy - so the tests that hand torch-allocated device
    # buffers to the engine need torch's runtime up BEFORE the engine library is opened.  (bench.py does the same by construction.)
    try:
        from qn_amd import engine
        engine.DEBUG_KNOBS_FROM_ENV = True       # harness opt-in: a whole test run under QN_DEBUG_KNOBS (tools/gpu_ab.sh)
    except Exception:
        pass
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
