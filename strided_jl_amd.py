"""Import alias: the package directory is ``strided.jl_amd`` (it carries the reference's name,
which is not a valid Python identifier).  ``import strided_jl_amd`` loads that directory as a
regular package under this module's name."""
import importlib.util as _u
import os as _os
import sys as _sys

_d = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "strided.jl_amd")
_spec = _u.spec_from_file_location(__name__, _os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
