"""Instruction mix of the loops that contain MFMAs in a -save-temps .s file: python tools/isa_loop_count.py file.s <kernel substring>"""
import collections, re, sys
s = open(sys.argv[1]).read()
key = sys.argv[2]
starts = [m.start() for m in re.finditer(r'^_Z\S*' + re.escape(key) + r'\S*:', s, flags=re.M)]
for st in starts:
    end = s.find('.end_amdhsa_kernel', st)
    lines = s[st:end].split('\n')
    name = lines[0].split(':')[0]
    label_at = {l[:-1].split(':')[0]: k for k, l in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:', l)}
    # backward branches define loops
    loops = []
    for k, l in enumerate(lines):
        m = re.match(r'\s+s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in label_at and label_at[m.group(1)] < k:
            loops.append((label_at[m.group(1)], k))
    print(name, 'lines', len(lines), 'mfma total', sum('v_mfma' in l for l in lines))
    for a, b in loops:
        body = lines[a:b + 1]
        n = sum('v_mfma' in l for l in body)
        if not n:
            continue
        cnt = collections.Counter()
        for l in body:
            l = l.strip()
            if not l or l.startswith(('.', ';')) or l.endswith(':'):
                continue
            op = l.split()[0]
            k = 'SALU' if op.startswith('s_') else 'MFMA' if 'mfma' in op else 'LDS' if op.startswith('ds_') else 'VMEM' if op.startswith(('global_', 'buffer_')) else 'VALU'
            cnt[k] += 1
        print('  loop lines %d-%d:' % (a, b), dict(cnt))
