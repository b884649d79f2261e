"""yaqs_amd: the YAQS Tensor Jump Method hot path on MI355X.

The names a script imports from ``mqt.yaqs`` (src/mqt/yaqs/__init__.py:17-37) for this path are available from the package root:

    from yaqs_amd import Simulator, State, Hamiltonian, AnalogSimParams, DigitalSimParams, NoiseModel, Observable

Nothing here touches the GPU or loads PyTorch at import time; ``Simulator.run`` does.
"""
from .api import (  # noqa: F401
    AnalogSimParams,
    DigitalSimParams,
    Hamiltonian,
    MPO,
    MPS,
    NoiseModel,
    Observable,
    Result,
    SIMULATION_PRESETS,
    State,
)

__all__ = ["AnalogSimParams", "DigitalSimParams", "Hamiltonian", "MPO", "MPS", "NoiseModel", "Observable", "Result", "SIMULATION_PRESETS", "Simulator", "State"]


def __getattr__(name):
    if name == "Simulator":  # deferred: tjm.py binds the HIP library lazily but keeps the import light anyway
        from .tjm import Simulator

        return Simulator
    raise AttributeError(name)
