"""Import alias: `import ieache_amd` resolves to the package directory
`ie-ache_amd/` (whose name, fixed by the project layout, is not a valid Python
identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ie-ache_amd")
_spec = importlib.util.spec_from_file_location(
    "ieache_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ieache_amd"] = _mod
_spec.loader.exec_module(_mod)
