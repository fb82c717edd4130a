"""Import alias: the package lives in the directory `parquet-go_amd/`, whose name is not a
valid Python identifier.  Importing this module installs that directory as the package
`parquet_go_amd` (so `import parquet_go_amd.capi` works)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parquet-go_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
