#!/usr/bin/env python3
"""Extract one kernel from a gfx950 .s file (hipcc -save-temps) and summarise its basic blocks: instruction-class
counts per block and the blocks that form loops (backward branches).
usage: kernel_isa.py file.s <substring of the mangled kernel name> [--dump out.s] [--min 8]"""
import re, sys, collections

def classify(op):
    if op.startswith("v_"):
        if "f64" in op:
            if op.startswith(("v_rcp", "v_rsq", "v_sqrt")): return "TRANS64"
            if op.startswith("v_fma"): return "FMA64"
            if op.startswith("v_mul"): return "MUL64"
            if op.startswith("v_add"): return "ADD64"
            return "OTH64"
        return "VALU32"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "SMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "WAIT"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "BR"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "VMEM"
    return "OTHER"

def main():
    path, pat = sys.argv[1], sys.argv[2]
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    minn = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 8
    lines = open(path).read().split("\n")
    start = end = None
    name = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and pat in m.group(1) and start is None:
            start, name = i, m.group(1)
        elif start is not None and l.startswith(".Lfunc_end"):
            end = i
            break
    if start is None:
        print("kernel not found"); return
    body = lines[start:end]
    print(name, len(body), "lines")
    if dump: open(dump, "w").write("\n".join(body))
    blocks, cur, order = {}, None, []
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1); blocks[cur] = []; order.append(cur); continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")): continue
        if cur is None:
            cur = "entry"; blocks[cur] = []; order.append(cur)
        blocks[cur].append(s.split()[0] + " " + " ".join(s.split()[1:]))
    idx = {b: i for i, b in enumerate(order)}
    tot = collections.Counter()
    for b in order:
        for ins in blocks[b]: tot[classify(ins.split()[0])] += 1
    print("TOTAL", dict(tot))
    # backward branches
    for b in order:
        for ins in blocks[b]:
            op = ins.split()[0]
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = ins.split()[-1]
                if tgt in idx and idx[tgt] <= idx[b]:
                    span = order[idx[tgt]: idx[b] + 1]
                    c = collections.Counter()
                    for sb in span:
                        for i2 in blocks[sb]: c[classify(i2.split()[0])] += 1
                    n = sum(c.values())
                    if n >= minn:
                        print(f"LOOP {tgt} .. {b} ({len(span)} blocks, {n} instrs): {dict(c)}")
    m = re.search(r"\.vgpr_count:\s+(\d+)", "\n".join(lines[end:end + 400]))
main()
