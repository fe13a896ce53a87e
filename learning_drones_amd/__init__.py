"""Import shim: the package directory is ``learning-drones_amd/`` (not a valid Python
identifier), so ``import learning_drones_amd`` resolves here and this module re-points the
package path at that directory and runs its ``__init__``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "learning-drones_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
