"""Debug probe: one (bad, noisy, sample, native) case of test_non_finite_inputs_fail_loudly_like_the_reference per process,
so that a runtime abort names its case.  python tests/probes/nonfinite_probe.py [case]   (no argument: all cases, one child each)"""
import itertools
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = list(itertools.product(("nan", "inf"), (True, False), (True, False), (False, True)))


def one(idx):
    import faulthandler

    faulthandler.enable()
    import numpy as np

    from oracle import tjm_oracle as o
    from yaqs_amd.api import AnalogSimParams, MPS, NoiseModel, Observable, Z as Zg
    from yaqs_amd.engine import BatchEngine
    from yaqs_amd.tjm import TrajectoryBatch

    bad, noisy, sample, native = CASES[idx]
    L = 4
    mpo = o.ising_mpo(L, 1.0, 0.5)
    noise = NoiseModel([{"name": "lowering", "sites": [i], "strength": 0.2} for i in range(L)])
    init = [t.copy() for t in o.MPSState.product(L, "x+").tensors]
    init[1][0, 0, 0] = float(bad)
    kw = dict(elapsed_time=0.2, dt=0.1, max_bond_dim=4, svd_threshold=1e-9, order=1, sample_timesteps=sample, random_seed=1)
    p = AnalogSimParams(observables=[Observable(Zg(), s) for s in range(L)], num_traj=2, **kw)
    e = BatchEngine(L, 4, 2, mpo)
    try:
        TrajectoryBatch(e, p, noise if noisy else None).run([0, 1], MPS(L, tensors=init), native=native)
        print("case", idx, CASES[idx], "-> returned numbers", flush=True)
    except Exception as ex:  # noqa: BLE001
        print("case", idx, CASES[idx], "->", type(ex).__name__, str(ex)[:100], flush=True)
    e.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(int(sys.argv[1]))
    else:
        for k in range(len(CASES)):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), str(k)], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, AMD_LOG_LEVEL=os.environ.get("AMD_LOG_LEVEL", "1")))
            print(f"== case {k} {CASES[k]} rc={r.returncode}")
            print(r.stdout[-600:])
            if r.returncode != 0:
                print(r.stderr[-3000:])
