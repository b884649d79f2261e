"""pytest configuration: marker registration and shared paths."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# one BLAS thread, like the reference's worker model (core/parallel_utils.py:307-312)
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun / the round-end driver)")


# TJM_SIM=1: run tests written for the GPU on tests/hipsim (the device code interpreted on the host; see tests/simengine.py).
# A debugging aid for containers without a GPU: `TJM_SIM=1 python -m pytest tests/test_hip_engine.py -k tdvp`.
SIM = os.environ.get("TJM_SIM") == "1"
if SIM:
    import torch

    import yaqs_amd.engine as _engine_mod
    import yaqs_amd.tjm as _tjm_mod
    from simengine import SimEngine

    _engine_mod.BatchEngine = SimEngine
    _tjm_mod.BatchEngine = SimEngine
    torch.cuda.is_available = lambda: True
    torch.cuda.mem_get_info = lambda device=None: (4 << 30, 4 << 30)
