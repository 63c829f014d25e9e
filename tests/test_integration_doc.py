"""INTEGRATION.md shows the ctypes binding a maintainer would add: its structure definitions must be the ABI's (CPU only)."""
import ctypes as C
import os
import re

import numpy as np

from frenetix_motion_planner_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _doc_structs():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    ns = {"C": C, "np": np}
    for b in blocks:
        # keep the class statements (with their continuation lines), drop everything that touches the library
        keep, on = [], False
        for line in b.split("\n"):
            if line.startswith("class "):
                on = True
            elif on and line and not line.startswith((" ", "\t")):
                on = False
            if on:
                keep.append(line)
        exec("\n".join(keep), ns)
    return ns


def _fields(cls):
    return [(f[0], C.sizeof(f[1])) for f in cls._fields_]


def test_documented_structures_are_the_abi():
    ns = _doc_structs()
    for name in ("FxVehicle", "FxProblem", "FxResult", "FxStateUpdate", "FxPackage"):
        assert name in ns, f"INTEGRATION.md no longer shows {name}"
        doc, abi = ns[name], getattr(_abi, name)
        assert C.sizeof(doc) == C.sizeof(abi), name
        assert _fields(doc) == _fields(abi), name


def test_documented_abi_version_and_symbols_exist():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"fx_abi_version\(\) == (\d+)", text)
    assert m and int(m.group(1)) == _abi.FX_ABI_VERSION
    header = open(os.path.join(ROOT, "include", "fxplan.h")).read()
    for sym in set(re.findall(r"\b(fx_[a-z_0-9]+)\(", text)):
        assert re.search(r"\b" + sym + r"\(", header), f"INTEGRATION.md mentions {sym}, which include/fxplan.h does not declare"
