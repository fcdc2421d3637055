"""Instruction mix of a kernel's hottest basic block (the one with the most MFMAs) in a .s file.
usage: isa_count.py file.s kernel-name-substring [outputs-per-lane-per-mfma-group]"""
import re
import sys
from collections import Counter

t = open(sys.argv[1]).read()
key = sys.argv[2]
m = None
for mm in re.finditer(r'^(\S+):\s*;\s*@\S+\n(.*?)s_endpgm', t, flags=re.S | re.M):
    if key in mm.group(1):
        m = mm
        break
if m is None:
    sys.exit("kernel not found")
body = m.group(2)
blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
best = None
for i in range(1, len(blocks), 2):
    b = blocks[i + 1]
    n = b.count('v_mfma')
    if best is None or n > best[0]:
        best = (n, blocks[i], b)
n, name, b = best
c = Counter()
for line in b.split('\n'):
    line = line.strip()
    if not line or line.startswith(';') or line.startswith('.'):
        continue
    c[line.split()[0]] += 1
valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
print(m.group(1), name, 'mfma', n, 'VALU', valu, 'ds', sum(v for k, v in c.items() if k.startswith('ds_')),
      'vmem', sum(v for k, v in c.items() if k.startswith('global_') or k.startswith('buffer_')),
      'salu', sum(v for k, v in c.items() if k.startswith('s_')))
for k, v in c.most_common(70):
    print("  %-28s %d" % (k, v))
