"""Importable name of the package directory `python-ray-tracer_amd/` (a hyphen cannot appear in an `import`
statement): loads that directory as the package `python_ray_tracer_amd` — no code lives here."""
import importlib.util as _u
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "python-ray-tracer_amd")
_spec = _u.spec_from_file_location(__name__, _os.path.join(_real, "__init__.py"), submodule_search_locations=[_real])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
