"""Import shim: the package directory is named `sph-pie_amd/` (the layout the build contract asks for), which
is not a valid Python identifier.  `import sph_pie_amd` loads that directory as the package `sph_pie_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sph-pie_amd")
_spec = importlib.util.spec_from_file_location(
    "sph_pie_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sph_pie_amd"] = _mod
_spec.loader.exec_module(_mod)
