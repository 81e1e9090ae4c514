#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: isa_blocks.py extract.s k_fast_cells [--ops]   -> block label, VALU / SALU / LDS / VMEM counts, branch targets"""
import re, sys, collections
src, name = sys.argv[1], sys.argv[2]
show_ops = '--ops' in sys.argv
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\d+%s\w*:' % name, l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm') and 'end' in ''.join(lines[i:i+6]) or lines[i].startswith('.Lfunc_end'))
blocks = []; cur = ['entry', collections.Counter(), [], collections.Counter()]
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append(cur); cur = [m.group(1), collections.Counter(), [], collections.Counter()]; continue
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    k = 'VALU' if op.startswith('v_') else 'SALU' if op.startswith('s_') else 'LDS' if op.startswith('ds_') else 'VMEM' if re.match(r'(global|buffer|flat|scratch)_', op) else 'other'
    cur[1][k] += 1; cur[3][op] += 1
    if op.startswith('s_cbranch') or op == 's_branch': cur[2].append(op.replace('s_cbranch_', '').replace('s_', '') + '->' + t.split()[1])
blocks.append(cur)
tot = collections.Counter()
for b in blocks:
    tot.update(b[1])
    print('%-12s V %4d  S %4d  L %3d  M %3d   %s' % (b[0], b[1]['VALU'], b[1]['SALU'], b[1]['LDS'], b[1]['VMEM'], ' '.join(b[2])))
    if show_ops:
        print('      ' + ' '.join('%s:%d' % kv for kv in b[3].most_common(12) if kv[0].startswith('v_')))
print('static total', dict(tot))
