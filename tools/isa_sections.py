"""Instruction counts between consecutive s_memtime stamps of a -DGDM_STAMPS build: tools/isa_sections.py file.s prefix."""
import sys, collections
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ':' in l][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
sec, h = 0, collections.Counter()
def cls(op):
    return ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_')
            else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_')) else 'other')
for l in lines[start + 1:end]:
    l = l.strip()
    if not l or l.startswith(('.', ';')) or l.endswith(':'):
        continue
    op = l.split()[0]
    if op == 's_memtime':
        print(f"section {sec}: {dict(h)}")
        sec += 1; h = collections.Counter()
        continue
    h[cls(op)] += 1
print(f"section {sec}: {dict(h)}")
