"""MI355X-native SINDy hot path (Theta-build + residual + sparsify + symmetry regularisers).

The directory name carries a hyphen (it is fixed by the build contract), so the package is
imported with ``importlib.import_module("symmetry-ode-discovery_amd")`` or, after the
root-level shim ``symode_amd.py`` is on the path, simply ``import symode_amd``.
"""
import sys as _sys

from . import (autoencoder, batched, constraint, data, dataset, engine, evaluation, library, lie, lstsq, model_utils,  # noqa: F401
               parser_utils, sindy, sweep, train)
from .engine import FLAG_EXP, FLAG_SINE, HipEngine, SymodeError, get_engine, library_flags  # noqa: F401

from .sindy import SINDyRegression, WSINDyWrapper, solve_SINDy, solve_SINDy_one_step  # noqa: F401

__all__ = ["engine", "SINDyRegression", "solve_SINDy", "solve_SINDy_one_step", "HipEngine", "SymodeError", "get_engine", "library_flags", "FLAG_SINE", "FLAG_EXP"]

# make the package reachable under an importable alias
_ALIAS = "symode_amd"
if _sys.modules.get(_ALIAS) is not _sys.modules[__name__]:
    _sys.modules[_ALIAS] = _sys.modules[__name__]
for _name, _mod in list(_sys.modules.items()):
    if _name.startswith(__name__ + "."):
        _sys.modules.setdefault(_ALIAS + _name[len(__name__):], _mod)
