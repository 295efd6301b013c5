"""Import shim: makes `import orthogonalsphericalshellgrids.jl_amd` resolve to the package whose
directory is literally named `orthogonalsphericalshellgrids.jl_amd/` (a dot is not importable
as a plain package name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    "orthogonalsphericalshellgrids.jl_amd")
_name = __name__ + ".jl_amd"
if _name not in sys.modules:
    _spec = importlib.util.spec_from_file_location(_name, os.path.join(_dir, "__init__.py"),
                                                   submodule_search_locations=[_dir])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_name] = _mod
    _spec.loader.exec_module(_mod)
jl_amd = sys.modules[_name]
