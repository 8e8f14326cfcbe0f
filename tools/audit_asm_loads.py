#!/usr/bin/env python3
"""tools/audit_asm_loads.py file.s kernel_substring -- walks the ISA of one kernel and flags every instruction that reads a
register which an earlier ds_read has not yet been waited for (LDS returns in order: s_waitcnt lgkmcnt(n) retires all but
the youngest n reads).  Inline-asm loads are invisible to hipcc, so under register pressure it may copy or consume their
destinations before the hand-placed wait (cdna guide 5.7); a clean audit (0 violations) is part of the build check for
mlp256_kernel.  Straight-line model: loop back-edges are not followed."""
import re
import sys


def regs(tok):
    m = re.match(r'([av])\[(\d+):(\d+)\]', tok)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3))
    m = re.match(r'([av])(\d+)$', tok)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(2))
    return None


def audit(path, name):
    text = open(path).read().split('\n')
    start = next(i for i, l in enumerate(text) if l.startswith('_Z') and name in l and l.rstrip().endswith(l.split(':')[0][-1] + ':' if False else l) and ':' in l)
    lines = []
    for l in text[start:]:
        lines.append(l)
        if 's_endpgm' in l:
            break
    pending, bad = [], 0
    for i, l in enumerate(lines):
        t = l.strip()
        if not t or t[0] in ';.':
            continue
        op = t.split()[0]
        args = [a.strip().rstrip(',') for a in t[len(op):].split(',')]
        if op == 's_waitcnt' and 'lgkmcnt' in t:
            n = int(re.search(r'lgkmcnt\((\d+)\)', t).group(1))
            pending = pending[-n:] if n > 0 else []
            continue
        if op.startswith('ds_read'):
            pending.append((regs(args[0]), i))
            continue
        for a in args[1:]:
            r = regs(a.split()[0]) if a else None
            if not r:
                continue
            for pr, li in pending:
                if pr and pr[0] == r[0] and not (r[2] < pr[1] or r[1] > pr[2]):
                    bad += 1
                    if bad <= 10:
                        print(f"line {i}: {t}   <- pending since line {li}: {lines[li].strip()}")
    print(f"{name}: {len(lines)} lines, violations {bad}")
    return bad


if __name__ == "__main__":
    sys.exit(1 if audit(sys.argv[1], sys.argv[2]) else 0)
