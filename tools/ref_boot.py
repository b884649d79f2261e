"""Boot the read-only reference (/root/reference) in THIS container only.

Used solely by tools/make_golden.py to emit golden fixtures.  Never imported by the
product, tests, smoke() or bench.py (the reference does not exist on the GPU box).

The reference's package __init__ pulls qiskit (absent here), so we register empty
namespace packages and import hot-path submodules directly; opt_einsum and numba are
replaced by behaviour-preserving stand-ins (numpy.einsum / no-op jit).
"""
from __future__ import annotations

import sys
import types
import importlib

REF_SRC = "/root/reference/src"


def boot():
    if "mqt.yaqs" in sys.modules and getattr(sys.modules["mqt.yaqs"], "_graft_boot", False):
        return
    import numpy as np

    mqt = types.ModuleType("mqt")
    mqt.__path__ = [REF_SRC + "/mqt"]
    yaqs = types.ModuleType("mqt.yaqs")
    yaqs.__path__ = [REF_SRC + "/mqt/yaqs"]
    yaqs._graft_boot = True
    sys.modules["mqt"] = mqt
    sys.modules["mqt.yaqs"] = yaqs

    oe = types.ModuleType("opt_einsum")

    def contract(*args, **kwargs):
        kwargs.pop("optimize", None)
        return np.einsum(*args, optimize=True)

    oe.contract = contract
    sys.modules["opt_einsum"] = oe

    nb = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.jit = jit
    nb.njit = jit
    nb.prange = range
    nb.get_num_threads = lambda: 1
    nb.set_num_threads = lambda n: None
    sys.modules["numba"] = nb
    sys.modules["mqt.yaqs.core.methods.lanczos_numba"] = None


class _StubFinder:
    """Auto-stub qiskit.* / cma so the digital driver module imports (its DAG front end is never called)."""

    PREFIXES = ("qiskit", "cma", "qiskit_qasm3_import")

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.PREFIXES:
            import importlib.machinery

            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []

        def _getattr(name):
            if name.startswith("__"):
                raise AttributeError(name)
            return type(name, (), {})

        m.__getattr__ = _getattr
        return m

    def exec_module(self, module):
        pass


def boot_digital():
    boot()
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StubFinder())


def ref(name: str):
    boot()
    if name.startswith("digital"):
        boot_digital()
    return importlib.import_module("mqt.yaqs." + name)
