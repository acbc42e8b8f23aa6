"""Import shim.  The package directory required by the repo layout is ``wiki-grx-gym_amd/`` (a
hyphen is not a valid Python identifier), so ``import wiki_grx_gym_amd`` resolves here and this
module re-points its ``__path__`` at the real directory and executes its ``__init__.py``."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "wiki-grx-gym_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
