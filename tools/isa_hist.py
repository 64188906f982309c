"""Instruction histogram per basic block of one kernel in a hipcc -S listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++20 -S --cuda-device-only [-D...] ppgs_amd/csrc/ppg_kernels.hip -o /tmp/k.s
    python tools/isa_hist.py /tmp/k.s ffn_mixed_kernelI8PrecBF16Li16ELb1 [min_instructions]
"""
import collections
import re
import sys

path, pattern = sys.argv[1], sys.argv[2]
minimum = int(sys.argv[3]) if len(sys.argv) > 3 else 150
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(pattern) + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, name, cur = [], 'entry', []
for i in range(start + 1, end):
    l = lines[i]
    m = re.match(r'^(\.LBB\S+):', l)
    if m:
        blocks.append((name, cur))
        name, cur = m.group(1), []
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
keys = ['v_mfma', 'ds_read', 'ds_write', 'global_load_lds', 'global_load', 'global_store', 'scratch', 'v_accvgpr',
        'v_cvt_pk', 'v_max', 'v_pk', 'v_add', 'v_fma', 'v_mul', 'v_sub', 'v_mov', 'v_perm', 'v_exp', 's_waitcnt',
        's_nop', 's_barrier', 's_']
total = collections.Counter()
for name, ins in blocks:
    c = collections.Counter()
    for op in ins:
        for k in keys:
            if op.startswith(k):
                c[k] += 1
                break
        else:
            c['other:' + op.split('_')[0] + '_' + (op.split('_')[1] if '_' in op else '')] += 1
    total.update(c)
    if len(ins) >= minimum:
        print(f'{name}: {len(ins)} instr  ' + '  '.join(f'{k}={v}' for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
print('TOTAL', sum(total.values()), dict(total.most_common(30)))
