#!/usr/bin/env python3
"""Order of global loads (L), stores (S), atomics (A), s_waitcnt vmcnt(N) (WN), MFMAs (M) and barriers (|B|) in each kernel of a
`hipcc -S --cuda-device-only` listing, from the first s_barrier (or the first MFMA) on - i.e. the steady-state loop.  A `W0` that
follows freshly issued loads means the register prefetch those loads were meant to be is waited for at once (typical cause:
loads under a run-time condition - the wait-count pass then assumes the worst on the merged path).
usage: isa_seq.py file.s [substring ...]"""
import re
import sys

t = open(sys.argv[1]).read()
subs = sys.argv[2:]
for full in re.findall(r'^(_Z\S+):\s*; @', t, re.M):
    if not all(s in full for s in subs):
        continue
    m = re.search(r'^' + re.escape(full) + r':(.*?)s_endpgm', t, re.S | re.M)
    if not m:
        continue
    body = m.group(1).splitlines()
    marks = [i for i, l in enumerate(body) if 's_barrier' in l] or [i for i, l in enumerate(body) if 'v_mfma' in l]
    if not marks:
        continue
    seq = []
    for l in body[marks[0]:]:
        s = l.strip()
        if s.startswith(('global_load', 'buffer_load')): seq.append('L')
        elif s.startswith(('global_store', 'buffer_store')): seq.append('S')
        elif s.startswith('global_atomic'): seq.append('A')
        elif 'vmcnt' in s and s.startswith('s_waitcnt'): seq.append('W' + re.search(r'vmcnt\((\d+)\)', s).group(1))
        elif s.startswith('v_mfma'): seq.append('M')
        elif s.startswith('s_barrier'): seq.append('|B|')
    out, prev, n = [], None, 0
    for x in seq:
        if x == prev:
            n += 1
        else:
            if prev: out.append(prev + (f"x{n}" if n > 1 else ""))
            prev, n = x, 1
    if prev: out.append(prev + (f"x{n}" if n > 1 else ""))
    print(full[:110])
    print("    " + " ".join(out)[:600])
