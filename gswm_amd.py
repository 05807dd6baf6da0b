"""Importable name for the package directory `a-watermark-for-diffusion-models_amd/` (whose name, fixed by the build
contract, is not a Python identifier).  `import gswm_amd` loads that directory as the package `gswm_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "a-watermark-for-diffusion-models_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
