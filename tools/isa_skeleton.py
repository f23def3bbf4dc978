"""Print the memory/MFMA/wait skeleton of one kernel from hipcc -S output (tools/isa_skeleton.py file.s mangled-prefix)."""
import sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ':' in l][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
keep = ('s_waitcnt', 's_barrier', 'v_mfma', 'global_load', 'global_store', 'ds_write', 'ds_read', 'ds_bpermute',
        'buffer_', 's_cbranch', '.LBB')
out = [l.strip().split(';')[0][:50] for l in lines[start:end] if l.strip().startswith(keep)]
res, prev, cnt = [], None, 0
for o in out:
    op = o.split(' ')[0]
    key = op if not (op.startswith('s_waitcnt') or op.startswith('.LBB') or op.startswith('s_cbranch')) else o
    if key == prev:
        cnt += 1
    else:
        if prev is not None:
            res.append(f"{prev} x{cnt}" if cnt > 1 else prev)
        prev, cnt = key, 1
res.append(f"{prev} x{cnt}")
print(' | '.join(res[:int(sys.argv[3]) if len(sys.argv) > 3 else 200]))
