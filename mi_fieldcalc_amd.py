"""Import shim: the package directory is named ``mi-fieldcalc_amd`` (with a
hyphen, as the project layout prescribes), which Python cannot import by name.
``import mi_fieldcalc_amd`` loads that directory as a regular package."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "mi-fieldcalc_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
