#!/usr/bin/env python3
"""Where does the host time of a synchronous plan step go? (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from frenetix_motion_planner_amd import synthetic, _abi
from frenetix_motion_planner_amd._lib import lib
from frenetix_motion_planner_amd.engine import FrenetEngine
import ctypes as C
inp = synthetic.make_inputs(ref_kind="arc", v0=10.0, grid=(19, 51, 51))
with FrenetEngine(max_candidates=inp.n_candidates + 64) as eng:
    eng.set_timing('kernel'); eng.upload(inp)
    for _ in range(20): eng.evaluate(); eng.finish()
    n = 300
    te = tf = 0.0; tot = []
    for _ in range(n):
        t0 = time.perf_counter(); eng.evaluate(); t1 = time.perf_counter(); eng.finish(); t2 = time.perf_counter()
        te += t1 - t0; tf += t2 - t1; tot.append(t2 - t0)
    print(f"python evaluate() {te/n*1e6:.1f} us, finish() {tf/n*1e6:.1f} us, total p50 {np.median(tot)*1e6:.1f} us, device {eng.last_kernel_ms*1e3:.1f} us (eval {eng.last_eval_kernel_ms*1e3:.1f})")
    for mode in ('stream', 'kernel', 'off'):
      eng.set_timing(mode)
      te = tf = 0.0; tot = []
      for _ in range(n):
        t0 = time.perf_counter(); eng.evaluate(); t1 = time.perf_counter(); eng.finish(); t2 = time.perf_counter()
        te += t1 - t0; tf += t2 - t1; tot.append(t2 - t0)
      print(f'timing={mode}: python evaluate() {te/n*1e6:.1f} us, finish() {tf/n*1e6:.1f} us, total p50 {np.median(tot)*1e6:.1f} us')
    eng.set_timing(False)
    L = lib(); res = (_abi.FxResult * 1)()
    te = tf = 0.0; tot = []
    for _ in range(n):
        t0 = time.perf_counter(); L.fx_evaluate(eng._ctx); t1 = time.perf_counter(); L.fx_finish_batch(eng._ctx, res); t2 = time.perf_counter()
        te += t1 - t0; tf += t2 - t1; tot.append(t2 - t0)
    print(f"raw C-ABI fx_evaluate {te/n*1e6:.1f} us, fx_finish {tf/n*1e6:.1f} us, total p50 {np.median(tot)*1e6:.1f} us")
