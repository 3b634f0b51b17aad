"""Import shim: the package directory is named `diffusion-conductor_amd/` (not a valid
Python identifier), so `import diffusion_conductor_amd` lands here and this module
replaces itself in sys.modules with the package loaded from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diffusion-conductor_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _pkg
_spec.loader.exec_module(_pkg)
