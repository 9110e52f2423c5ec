"""Import helper: the package directory is `motifs.jl_amd/` (a dot in the name),
so it is registered under the importable alias `motifs_jl_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "motifs.jl_amd")
ALIAS = "motifs_jl_amd"


def load_pkg():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    spec = importlib.util.spec_from_file_location(
        ALIAS, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod


def load_build():
    """The build helper alone (no ctypes import)."""
    spec = importlib.util.spec_from_file_location(ALIAS + "_build", os.path.join(PKG_DIR, "_build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
