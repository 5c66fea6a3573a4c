#!/usr/bin/env python3
"""List the backward-branch loops of one kernel in a hipcc -S listing with their instruction mix.
usage: tools/isa_loops.py listing.s kernel_label_substring [min_instructions]"""
import re, sys, collections
path, sub = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and sub in l.split(':')[0])
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
label_at = {}
for i in range(start, end):
    m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
    if m: label_at[m.group(1)] = i
def is_inst(l):
    return re.match(r'^\s+[a-z]', l) and not l.strip().startswith(('.', ';'))
for i in range(start, end):
    m = re.match(r'^\s+(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', lines[i])
    if m and m.group(2) in label_at and label_at[m.group(2)] < i:
        body = [l for l in lines[label_at[m.group(2)]:i + 1] if is_inst(l)]
        if len(body) < minsz: continue
        c = collections.Counter()
        for l in body:
            op = l.split()[0]
            if op.startswith('v_') and '_f64' in op: c['valu_f64'] += 1
            elif op.startswith('v_'): c['valu_other'] += 1
            elif op.startswith('s_mov') : c['s_mov'] += 1
            elif op.startswith('s_waitcnt') or op.startswith('s_nop'): c['wait/nop'] += 1
            elif op.startswith('s_'): c['salu_other'] += 1
            elif op.startswith('ds_'): c['lds'] += 1
            elif op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')): c['vmem'] += 1
            else: c[op] += 1
        print(f"{m.group(2)} lines {label_at[m.group(2)]}-{i}: {len(body)} instructions", dict(c))
