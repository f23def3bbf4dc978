"""Instruction-class histogram of one kernel (whole body) from hipcc -S output: tools/isa_hist.py file.s mangled-prefix."""
import sys, collections
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ':' in l][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
h = collections.Counter(); ops = collections.Counter()
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l.startswith(('.', ';')) or l.endswith(':'):
        continue
    op = l.split()[0]
    ops[op] += 1
    cls = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_')
           else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other')
    h[cls] += 1
print(dict(h))
print(ops.most_common(40))
