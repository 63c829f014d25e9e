"""Importable alias of the product package.

The product lives in the directory `frenetix-motion-planner_amd/` (the name the build contract
asks for); a hyphen cannot appear in a Python module name, so this stub re-homes itself onto that
directory: after import, `frenetix_motion_planner_amd` *is* the package defined there.
"""
import importlib.util
import os
import sys

_real = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "frenetix-motion-planner_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_real, "__init__.py"),
                                               submodule_search_locations=[_real])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
