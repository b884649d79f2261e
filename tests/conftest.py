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


# Host-memory watchdog: a checker that outgrows the box takes the GPU box down with it (round 2 lost two boxes to an O(chi^4) host-side
# contraction).  A daemon thread samples this process's resident set; above the budget it prints what was running and leaves with a
# non-zero code (os._exit, never a re-exec: a process that has touched the GPU must not be replaced).
RSS_BUDGET_BYTES = int(float(os.environ.get("TJM_TEST_RSS_GB", "40")) * (1 << 30))
_current_test = ["<collection>"]


def _rss_bytes():
    with open("/proc/self/statm") as fh:
        return int(fh.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def _watchdog():
    import time

    while True:
        time.sleep(0.5)
        try:
            rss = _rss_bytes()
        except OSError:
            return
        if rss > RSS_BUDGET_BYTES:
            sys.stderr.write(f"\n[watchdog] host RSS {rss / 2**30:.1f} GiB > {RSS_BUDGET_BYTES / 2**30:.0f} GiB during {_current_test[0]}: "
                             "leaving with exit code 3\n")
            sys.stderr.flush()
            os._exit(3)


def pytest_sessionstart(session):
    import threading

    threading.Thread(target=_watchdog, name="rss-watchdog", daemon=True).start()


def pytest_runtest_setup(item):
    _current_test[0] = item.nodeid


def host_bytes_budget(nbytes, what):
    """Checkers whose cost grows faster than chi^3 call this before they allocate."""
    limit = RSS_BUDGET_BYTES // 4
    assert nbytes < limit, f"{what} would need {nbytes / 2**30:.1f} GiB of host memory (limit {limit / 2**30:.1f} GiB): shrink the check"
