#!/usr/bin/env python3
"""Opcode histogram of the walk loop (the loop with the most FP64 FMAs) of one kernel in a gfx950 .s file.
usage: loop_ops.py file.s <substring of the mangled kernel name> [n_top]"""
import collections, re, subprocess, sys, os
path, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
here = os.path.dirname(os.path.abspath(__file__))
out = subprocess.run([sys.executable, os.path.join(here, "kernel_isa.py"), path, pat, "--dump", "/tmp/_loop_ops.s", "--min", "200"],
                     capture_output=True, text=True).stdout.split("\n")
best = None
for l in out:
    m = re.match(r"LOOP (\S+) \.\. (\S+) \((\d+) blocks, (\d+) instrs\): (.*)", l)
    if m:
        d = eval(m.group(5))
        key = (d.get("FMA64", 0), -int(m.group(4)))
        if best is None or key > best[0]:
            best = (key, m.group(1), m.group(2), int(m.group(4)), d)
_, a, b, n, d = best
lines = open("/tmp/_loop_ops.s").read().split("\n")
ia = next(i for i, l in enumerate(lines) if l.startswith(a + ":"))
ib = next(i for i, l in enumerate(lines) if l.startswith(b + ":"))
# the loop ends at the end of block b: next label after ib
ie = next((i for i in range(ib + 1, len(lines)) if re.match(r"^\.LBB\d+_\d+:", lines[i])), len(lines))
ops = collections.Counter()
for l in lines[ia:ie]:
    s = l.strip()
    if not s or s.startswith((";", ".", "//")):
        continue
    ops[s.split()[0]] += 1
v32 = {k: v for k, v in ops.items() if k.startswith("v_") and "f64" not in k}
f64 = sum(v for k, v in ops.items() if k.startswith("v_") and "f64" in k)
print(f"{pat}: loop {a}..{b} {n} instrs: FP64 {f64}, other VALU {sum(v32.values())}, SALU {sum(v for k, v in ops.items() if k.startswith('s_'))}, "
      f"LDS {sum(v for k, v in ops.items() if k.startswith('ds_'))}, VMEM {sum(v for k, v in ops.items() if k.startswith(('global_', 'scratch_', 'buffer_')))}")
print("   ", sorted(v32.items(), key=lambda kv: -kv[1])[:top])
