"""Import shim: the product package lives in the directory `3d-reconstruction_amd/` (not a valid Python
identifier), so it is loaded here under the module name `recon_amd`:

    import recon_amd
    from recon_amd import TensorVMSplit, TensorCP, OctreeRender_trilinear_fast
"""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "3d-reconstruction_amd")
_spec = importlib.util.spec_from_file_location("recon_amd", os.path.join(_PKG_DIR, "__init__.py"),
                                               submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["recon_amd"] = _mod
_spec.loader.exec_module(_mod)
