"""Import alias: `import fdcap_amd` loads the package directory `4dcapture-fpv_amd/`
(whose name is not a valid Python identifier) as the package `fdcap_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "4dcapture-fpv_amd")
_spec = importlib.util.spec_from_file_location(
    "fdcap_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["fdcap_amd"] = _mod
_spec.loader.exec_module(_mod)
