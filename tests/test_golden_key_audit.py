"""Every key of every committed fixture is read by some test.

A golden vector nobody compares against is a hole in the net: round 4 stored `tau_lat` (the delta_tau of each lateral
polynomial) in all 27 plan-step fixtures while no test read it, and the LOW_VEL_MODE mismatch it would have caught went
unnoticed.  This audit is static (it reads the test sources), so it holds for every selection of tests:

  * plan-step fixtures (tests/golden/INDEX.json): every key of every .npz must be subscripted -- fx["key"], or named in the
    tuple of a `for k in ("a", "b"):` loop whose body reads fx[k] -- in a tests/test_*.py file (an expectation) or in
    tests/fixtures.py (an input the scenario is rebuilt from);
  * refpath_golden.npz: every "<path>/<what>" key's <what> is read by tests/test_ref_path.py;
  * cpp_adapter_trace*.npz: every array is referenced by the recorded call trace next to it (tests/dropin/trace_recorder.replay
    resolves "@array" references), and the trace itself is replayed by a test;
  * every other data file under tests/golden is named by some test module.
"""
import glob
import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
ME = os.path.basename(__file__)


def _sources(pattern):
    out = {}
    for p in sorted(glob.glob(os.path.join(HERE, pattern))):
        if os.path.basename(p) != ME:
            out[os.path.basename(p)] = open(p).read()
    return out


def _keys_read(src, var="fx"):
    """keys `var` is subscripted with in `src`: literal subscripts, and the literals of a for-tuple whose body reads var[name]"""
    keys = set(re.findall(var + r"""\[\s*["']([^"']+)["']\s*\]""", src))
    lines = src.split("\n")
    for n, line in enumerate(lines):
        m = re.match(r"\s*for\s+(\w+)\s+in\s+\((.*)", line)
        if not m:
            continue
        name, head = m.group(1), m.group(2)
        k = n
        while ")" not in head and k + 1 < len(lines):   # (a tuple broken over lines)
            k += 1
            head += lines[k]
        body = "\n".join(lines[k + 1:k + 12])
        if re.search(var + r"\[\s*" + name + r"\s*\]", body):
            keys |= set(re.findall(r"""["']([^"']+)["']""", head.split(")")[0]))
    return keys


def test_every_key_of_the_plan_step_fixtures_is_read():
    names = sorted(json.load(open(os.path.join(GOLDEN, "INDEX.json"))))
    assert len(names) >= 27
    expectations = set()
    for src in _sources("test_*.py").values():
        expectations |= _keys_read(src)
    inputs = _keys_read(_sources("fixtures.py")["fixtures.py"])
    unread = {}
    for name in names:
        with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
            missing = sorted(k for k in z.files if k not in expectations and k not in inputs)
        if missing:
            unread[name] = missing
    assert not unread, f"fixture keys no test reads: {unread}"
    # the key whose absence prompted this audit is compared both on the oracle's side and on the HIP side
    assert "tau_lat" in _keys_read(_sources("test_oracle_golden.py")["test_oracle_golden.py"])
    assert "tau_lat" in _keys_read(_sources("test_hip_parity.py")["test_hip_parity.py"])


def test_every_key_of_the_reference_path_fixture_is_read():
    src = _sources("test_ref_path.py")["test_ref_path.py"]
    with np.load(os.path.join(GOLDEN, "refpath_golden.npz")) as z:
        whats = sorted({k.split("/", 1)[1] for k in z.files})
    missing = [w for w in whats if not re.search(r"""/""" + re.escape(w) + r"""["']""", src)]
    assert not missing, f"refpath_golden.npz entries test_ref_path.py never reads: {missing}"


def test_every_array_of_the_adapter_traces_is_referenced_and_replayed():
    tests = "\n".join(_sources("test_*.py").values())
    for path in sorted(glob.glob(os.path.join(GOLDEN, "cpp_adapter_trace*.npz"))):
        doc = open(path.replace(".npz", ".json")).read()
        refs = set(re.findall(r'"@array":\s*"([^"]+)"', doc))
        with np.load(path) as z:
            missing = sorted(set(z.files) - refs)
        assert not missing, f"{os.path.basename(path)}: arrays the recorded trace never references: {missing}"
        assert os.path.basename(path).replace(".npz", "") in tests, f"no test replays {os.path.basename(path)}"


def test_every_data_file_under_golden_is_named_by_a_test():
    tests = "\n".join(list(_sources("test_*.py").values()) + list(_sources("fixtures.py").values()))
    index = set(json.load(open(os.path.join(GOLDEN, "INDEX.json"))))
    orphans = []
    for p in sorted(os.listdir(GOLDEN)):
        stem, ext = os.path.splitext(p)
        if ext not in (".npz", ".json") or p == "INDEX.json" or stem in index:
            continue
        if stem not in tests and p not in tests:
            orphans.append(p)
    assert not orphans, f"fixture files no test names: {orphans}"
