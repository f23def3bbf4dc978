"""Per-basic-block instruction-class counts of one kernel from hipcc -S output (labels, branches):
tools/isa_blocks.py file.s mangled-prefix"""
import sys, collections
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ':' in l][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
def cls(op):
    return ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_')
            else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_')) else 'other')
name, h, br = 'entry', collections.Counter(), []
def flush():
    if sum(h.values()):
        print(f"{name:12s} {dict(h)}  -> {' '.join(br)}")
for l in lines[start + 1:end + 1]:
    l = l.strip()
    if not l or l.startswith((';', '.p2align', '.loc', '.cfi')):
        continue
    if l.startswith('.LBB') and ':' in l:
        flush()
        name, h, br = l.split(':')[0], collections.Counter(), []
        continue
    if l.startswith('.'):
        continue
    op = l.split()[0]
    h[cls(op)] += 1
    if op.startswith(('s_cbranch', 's_branch')):
        br.append(op.replace('s_cbranch_', '') + ':' + l.split()[1])
flush()
