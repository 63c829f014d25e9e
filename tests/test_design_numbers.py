"""DESIGN.md section 5.4 quotes measured numbers; the block between its two marker lines is the output of
tools/design_numbers.py on the tracked evidence files (profiles/r3/).  Prose and evidence cannot drift apart unnoticed."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("design_numbers", os.path.join(ROOT, "tools", "design_numbers.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_design_block_is_the_scripts_output():
    dn = _tool()
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert dn.BEGIN in text and dn.END in text
    block = text[text.index(dn.BEGIN):text.index(dn.END) + len(dn.END)]
    assert block == dn.block(), "DESIGN.md 5.4 is out of date: run python tools/design_numbers.py --write"


def test_block_covers_the_workloads_and_kernels_of_the_bench_line():
    dn = _tool()
    b = dn.block()
    for needle in ("config 1", "config 2", "config 3", "config 4", "config 5", "1 M candidates", "fx_obstacle_kernel", "fx_select_kernel",
                   "fed from host buffers"):
        assert needle in b, needle
    # the column fed from host buffers stands before the resident one
    header = [l for l in b.splitlines() if l.startswith("| workload")][0]
    assert header.index("fed from host buffers") < header.index("resident")
