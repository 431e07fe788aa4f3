#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file.

  tools/isa_count.py FILE.s SUBSTRING [SUBSTRING...]

Prints, for the first kernel whose mangled name contains every SUBSTRING, the VGPR/SGPR/LDS/
scratch figures and the instruction mix of its hottest loop (the longest backward-branch body),
which for rhs_kernel is the bottom-to-top level march (PF levels per trip).
"""
import re
import sys
from collections import Counter


def main():
    path, subs = sys.argv[1], sys.argv[2:]
    lines = open(path, errors="ignore").read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.endswith(":") is False and not re.match(r"^_Z\w+:", l):
            continue
        m = re.match(r"^(_Z\w+):", l)
        if m and all(s in m.group(1) for s in subs):
            start = i
            name = m.group(1)
            break
    if start is None:
        sys.exit("kernel not found")
    end = start
    while end < len(lines) and not lines[end].startswith("\t.end_amdhsa_kernel") and ".Lfunc_end" not in lines[end]:
        end += 1
    body = lines[start:end]
    print(name)
    for l in lines[end:end + 80]:
        if re.search(r"\.(sgpr|vgpr)_count|NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|\.amdhsa_next_free_vgpr|\.amdhsa_group_segment", l):
            print("  ", l.strip())
    # label positions and backward branches
    pos = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            pos[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in pos and pos[tgt] < i:
                loops.append((i - pos[tgt], pos[tgt], i))
    loops.sort(reverse=True)
    def mix(a, b):
        c = Counter()
        for l in body[a:b + 1]:
            m = re.match(r"^\s+([a-z_0-9]+)\s", l)
            if not m:
                continue
            op = m.group(1)
            if op.startswith("v_"):
                c["VALU"] += 1
                if "f64" in op or "b64" in op: c["  v_*64"] += 1
                if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_log", "v_exp")): c["  trans"] += 1
                if op.startswith("v_mov"): c["  v_mov"] += 1
                if op.startswith("v_cndmask"): c["  v_cndmask"] += 1
                if op.startswith("v_cmp"): c["  v_cmp"] += 1
                if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): c["  lane"] += 1
            elif op.startswith("s_"):
                c["SALU"] += 1
                if op.startswith("s_waitcnt"): c["  s_waitcnt"] += 1
                if op.startswith("s_nop"): c["  s_nop"] += 1
            elif op.startswith("ds_"):
                c["LDS"] += 1
            elif op.startswith(("buffer_", "global_", "flat_", "scratch_")):
                c["VMEM"] += 1
                if op.startswith("scratch_"): c["  scratch"] += 1
        return c
    total = mix(0, len(body) - 1)
    print("whole kernel:", dict(total))
    for n, (length, a, b) in enumerate(loops[:3]):
        print(f"loop {n}: lines {a}-{b} ({length} lines):", dict(mix(a, b)))


if __name__ == "__main__":
    main()
