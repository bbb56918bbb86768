"""Import shim: the package lives in `pytorch-studiogan_amd/` (not a valid Python identifier), so
`import studiogan_amd` loads this file, which installs that directory as the package `studiogan_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pytorch-studiogan_amd")
_spec = importlib.util.spec_from_file_location("studiogan_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["studiogan_amd"] = _mod
_spec.loader.exec_module(_mod)
