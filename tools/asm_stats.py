#!/usr/bin/env python3
"""Per-kernel VGPR / scratch / s_waitcnt vmcnt histogram from a `hipcc -S --cuda-device-only` listing.
usage: asm_stats.py file.s [substring ...]   (a kernel is listed when every substring occurs in its mangled name)"""
import re
import sys
from collections import Counter

t = open(sys.argv[1]).read()
subs = sys.argv[2:]
labels = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\S+):\s*; @', t, re.M)]
for k, (pos, name) in enumerate(labels):
    end = labels[k + 1][0] if k + 1 < len(labels) else len(t)
    body = t[pos:end]
    if '.amdhsa_kernel' not in body and 's_endpgm' not in body:
        continue
    if not all(s in name for s in subs):
        continue
    code = body.split('.section')[0]
    meta = t[pos:]
    desc = re.search(r'\.amdhsa_kernel ' + re.escape(name) + r'\n(.*?)\.end_amdhsa_kernel', t, re.S)
    d = desc.group(1) if desc else ''
    g = lambda key: (re.search(r'\.amdhsa_' + key + r' (\d+)', d) or [None, '?'])[1]
    waits = Counter(re.findall(r's_waitcnt vmcnt\((\d+)\)', code))
    mfma = len(re.findall(r'v_mfma', code))
    print(f"{name[:90]}\n   vgpr {g('next_free_vgpr')} accum_offset {g('accum_offset')} scratch {g('private_segment_fixed_size')} "
          f"lds {g('group_segment_fixed_size')} mfma {mfma} vmcnt {sorted(waits.items(), key=lambda x: int(x[0]))}")
