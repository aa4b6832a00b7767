"""Import alias: the package directory is `kge-keras_amd/` (not a valid Python identifier), so
`import kge_keras_amd` loads it from there and replaces this module in sys.modules."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "kge-keras_amd")
_spec = _u.spec_from_file_location("kge_keras_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["kge_keras_amd"] = _mod
_spec.loader.exec_module(_mod)
