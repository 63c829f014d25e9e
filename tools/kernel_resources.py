#!/usr/bin/env python3
"""Parse `-Rpass-analysis=kernel-resource-usage` remarks (stdin or file) into one line per kernel: VGPRs, spills, occupancy, LDS."""
import re, sys, subprocess
txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*?)(?: \[-Rpass|$)", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        m = re.search(r": remark: (.*?) \[-Rpass", line)
    body = m.group(1) if m else line
    m = re.search(r"Function Name: (\S+)", body)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    for key in ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "SGPRs Spill", "VGPRs Spill", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", body)
        if m and cur and key not in rows[cur]:
            rows[cur][key] = int(m.group(1))
names = list(rows)
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.splitlines()
except Exception:
    dem = names
for n, d in zip(names, dem):
    r = rows[n]
    short = re.sub(r"\(.*", "", d).replace("void ", "")
    print(f"{short:60s} VGPR {r.get('VGPRs', -1):4d} spillV {r.get('VGPRs Spill', 0):3d} spillS {r.get('SGPRs Spill', 0):3d} scratch {r.get('ScratchSize [bytes/lane]', 0):4d} "
          f"occ {r.get('Occupancy [waves/SIMD]', -1)} lds {r.get('LDS Size [bytes/block]', 0)}")
