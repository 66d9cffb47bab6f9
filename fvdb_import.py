"""Import helper: the package directory is `fabstir-vectordb_amd/` (hyphen), which the import
statement cannot spell.  `load()` imports it by path and registers the alias
`fabstir_vectordb_amd` in sys.modules."""
import importlib
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
ALIAS = "fabstir_vectordb_amd"


def load():
    if ALIAS in sys.modules:
        return sys.modules[ALIAS]
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    spec = importlib.util.spec_from_file_location(
        ALIAS, os.path.join(ROOT, "fabstir-vectordb_amd", "__init__.py"),
        submodule_search_locations=[os.path.join(ROOT, "fabstir-vectordb_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[ALIAS] = mod
    spec.loader.exec_module(mod)
    return mod
