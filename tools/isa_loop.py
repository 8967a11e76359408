"""Summarise the hottest basic block (most MFMAs) of one kernel in a -save-temps .s file.
usage: python tools/isa_loop.py file.s '<mangled-name-substring>'"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
key = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(key) + r'\S*):', s, re.M)
i = m.start()
j = s.index('s_endpgm', i)
body = s[i:j].split('\n')
blocks, cur = [], []
for l in body:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur); cur = [l]
    else:
        cur.append(l)
blocks.append(cur)
best = max(blocks, key=lambda b: sum('v_mfma' in x for x in b))
c = Counter()
seq = []
for l in best:
    mm = re.match(r'^\s+([a-z_0-9]+)(.*)', l)
    if not mm:
        continue
    op = mm.group(1)
    c[op] += 1
    if op.startswith('s_waitcnt'):
        seq.append(' W[' + mm.group(2).strip() + '] ')
    elif 'mfma' in op:
        seq.append('M')
    elif op.startswith('ds_read'):
        seq.append('L')
    elif op.startswith('s_nop'):
        seq.append('n')
    elif op.startswith('global_load') or op.startswith('buffer_load'):
        seq.append('G')
    elif op.startswith('s_barrier'):
        seq.append(' BAR ')
    else:
        seq.append('.')
print('block lines', len(best), 'mfma', c['v_mfma_f32_16x16x4_f32'])
print(c.most_common(20))
print(''.join(seq)[:2500])
