"""Instruction-class histogram of one kernel in a hipcc -S listing (static counts; loops counted once).
usage: python tools/isa_count.py file.s <substring of the mangled kernel name> [--loop]"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'\n(_Z\S*' + re.escape(pat) + r'\S*):', s)
body = s[m.end():]
body = body[:body.index('s_endpgm')]
cnt = collections.Counter()
region = 'setup'
reg = collections.defaultdict(collections.Counter)
for line in body.split('\n'):
    if 'LVGMARK' in line:
        region = line.split('LVGMARK')[1].strip(); continue
    line = line.split(';')[0].strip()
    if not line or line.startswith('.') or line.endswith(':'):
        continue
    op = line.split()[0]
    cls = 'mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'wait' if op.startswith('s_waitcnt') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem'
    reg[region][cls] += 1
    if op.startswith('v_mfma'): cnt['mfma'] += 1
    elif op.startswith('v_'): cnt['valu'] += 1; cnt['  valu:' + op.split('_e')[0]] += 1
    elif op.startswith('s_waitcnt'): cnt['s_waitcnt'] += 1
    elif op.startswith('s_barrier'): cnt['s_barrier'] += 1
    elif op.startswith('s_'): cnt['salu'] += 1
    elif op.startswith('ds_'): cnt['lds'] += 1; cnt['  lds:' + op] += 1
    elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): cnt['vmem'] += 1; cnt['  vmem:' + op] += 1
    else: cnt['other:' + op] += 1
print(m.group(1)[:120])
top = [(k, v) for k, v in cnt.items() if not k.startswith('  ')]
for k, v in sorted(top, key=lambda kv: -kv[1]): print(f'{k:12s} {v}')
for k, v in sorted([(k, v) for k, v in cnt.items() if k.startswith('  ')], key=lambda kv: -kv[1])[:40]: print(f'{k:34s} {v}')

print('regions (static):')
for r, c in reg.items(): print(f'  {r:10s}', dict(c))
