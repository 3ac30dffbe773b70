"""Import shim: ``import offk_amd`` loads the package kept in the directory
``optical-flow-guided-feature-pytorch_amd/`` (a name Python cannot import directly)."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "optical-flow-guided-feature-pytorch_amd")
_spec = importlib.util.spec_from_file_location(
    "offk_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["offk_amd"] = _mod
_spec.loader.exec_module(_mod)
