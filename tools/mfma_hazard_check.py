#!/usr/bin/env python3
"""Scan gfx950 assembly for VALU / LDS / VMEM instructions that read a register written by a v_mfma_f64 less than
WAIT instruction slots earlier (s_nop N counts N + 1).  hipcc (ROCm 7.2) does not always insert the wait states between
an f64 MFMA and a VALU read of its LAST destination registers (round 2: a reverse-sweep instantiation read e[3] of a
tile one instruction after the MFMA and got the value from before the last k-step).
Prints, per function, the reads by destination-register pair index and distance.   usage: ... file.s [--wait 11]"""
import re, sys
from collections import Counter

WAIT = 12
ASMW = 3  # slots between an inline-asm VALU write and an MFMA that reads the register (s_nop 1 between them is the least that works)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--wait" in sys.argv:
    WAIT = int(sys.argv[sys.argv.index("--wait") + 1])

def regs(op):
    m = re.match(r'v\[(\d+):(\d+)\]', op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', op)
    return {int(m.group(1))} if m else set()

total = Counter()
for path in args:
    func = "?"
    pend = []   # (dst lo, dst regs, states since issue)
    asmw = []   # (registers written by a VALU instruction inside inline asm, states since)
    inasm = False
    for line in open(path):
        m = re.match(r'^(_Z\w+):', line)
        if m:
            func, pend, asmw = m.group(1), [], []
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
        elif t.startswith(";;#ASMEND"):
            inasm = False
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")] if " " in t else []
        states = 1
        if op == "s_nop":
            states = int(ops[0]) + 1
        elif op.startswith("v_mfma"):
            d = regs(ops[0])
            srcs = set().union(*[regs(o) for o in ops[1:3]])   # A, B (an accumulator chain srcC == dst is fine)
            # round 6: a VALU instruction inside an inline-asm statement wrote a source of this MFMA fewer than ASMW slots ago --
            # gfx950 does not interlock VALU write -> MFMA SrcA / SrcB / SrcC read (tools/ubench_srcc_war.hip: the next slot and
            # one s_nop 0 read the old value), and hipcc pads only behind its own instructions, not behind inline asm
            allsrc = set().union(*[regs(o) for o in ops[1:4]])
            for w_, st2 in asmw:
                if (w_ & allsrc) and st2 < ASMW:
                    total[(func[:60], "asm-valu", 0, st2)] += 1
            for lo, dr, st in pend:
                if srcs & dr and st < WAIT:
                    total[(func[:60], "mfma-src", (min(srcs & dr) - lo) // 2, st)] += 1
            pend = [(lo, dr, st) for lo, dr, st in pend if not (d & dr)]
            pend.append((min(d), d, 0))
            pend = [(lo, dr, st + 1) for lo, dr, st in pend[:-1]] + [pend[-1]]
            asmw = [(w_, st2 + 1) for w_, st2 in asmw if st2 + 1 < 8]
            continue
        elif op.startswith("v_") or op.startswith("ds_") or op.startswith("buffer_") or op.startswith("global_"):
            srcs = set().union(*[regs(o.split()[0]) for o in ops[1:]]) if len(ops) > 1 else set()
            for lo, dr, st in pend:
                hit = srcs & dr
                if hit and st < WAIT:
                    total[(func[:60], "read", (min(hit) - lo) // 2, st)] += 1
            if ops:
                w = regs(ops[0].split()[0])
                pend = [(lo, dr - w, st) for lo, dr, st in pend]
        pend = [(lo, dr, st + states) for lo, dr, st in pend if dr and st + states < 64]
        if inasm and op.startswith("v_") and not op.startswith("v_mfma") and ops:
            asmw.append((regs(ops[0].split()[0]), 0))
        asmw = [(w_, st2 + states) for w_, st2 in asmw if st2 + states < 8]
for (f, kind, pair, st), n in sorted(total.items()):
    print("%-62s %-8s dst pair %d read %2d slots after issue  x%d" % (f, kind, pair, st, n))
# violation: a read of destination pair p fewer than 7 + p slots after the MFMA (the distances this compiler keeps where it
# does insert the wait states)
bad = [k for k in total if (k[1] == "asm-valu") or (k[1] != "asm-valu" and k[3] < 7 + k[2])]
if bad:
    print("VIOLATIONS: %d" % len(bad))
sys.exit(1 if bad else 0)
