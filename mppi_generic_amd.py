"""Import shim: the package directory is ``mppi-generic_amd/`` (not a valid Python identifier), so this module loads it
under the importable name ``mppi_generic_amd`` and replaces itself with it in ``sys.modules``."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "mppi-generic_amd")
_spec = importlib.util.spec_from_file_location(
    "mppi_generic_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mppi_generic_amd"] = _mod
_spec.loader.exec_module(_mod)
