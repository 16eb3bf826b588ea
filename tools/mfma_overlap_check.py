#!/usr/bin/env python3
"""Scan gfx950 assembly for v_mfma instructions whose destination registers overlap a source operand that is not the
accumulator.  hipcc (ROCm 7.2) allows this for v_mfma_f64_16x16x4_f64 with an inline-constant srcC (found in round 2: a
reverse-sweep instantiation computed one exponent register wrong); the hardware reads A / B while it already writes vdst.
usage: python tools/mfma_overlap_check.py file.s [...]   (exit code 1 if an overlap is found)"""
import re, sys

def rng(op):
    m = re.match(r'([va])\[(\d+):(\d+)\]', op)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3))
    m = re.match(r'([va])(\d+)$', op)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(2))
    return None

bad = 0
for path in sys.argv[1:]:
    func = "?"
    for ln, line in enumerate(open(path), 1):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            func = m.group(1)
        t = line.strip()
        if not t.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        d, a, b, c = rng(ops[0]), rng(ops[1]), rng(ops[2]), rng(ops[3])
        for name, s in (("A", a), ("B", b)):
            if d and s and d[0] == s[0] and not (s[2] < d[1] or s[1] > d[2]):
                print("%s:%d %s: vdst overlaps src%s: %s" % (path, ln, func[:70], name, t))
                bad += 1
        if d and c and d[0] == c[0] and c != d and not (c[2] < d[1] or c[1] > d[2]):
            print("%s:%d %s: vdst partially overlaps srcC: %s" % (path, ln, func[:70], t))
            bad += 1
sys.exit(1 if bad else 0)
