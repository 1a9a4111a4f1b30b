"""Import helper: the package directory is `ray-tracing-extended_amd` (not a valid identifier), so it is loaded
by path and registered as `rtx_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "ray-tracing-extended_amd")


def load():
    if "rtx_amd" in sys.modules:
        return sys.modules["rtx_amd"]
    spec = importlib.util.spec_from_file_location("rtx_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rtx_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
