"""v_accvgpr_read / v_accvgpr_write / MFMA counts per kernel of a hipcc -S listing: python tools/acc_moves.py file.s
(a kernel that may use all 512 registers gets the accumulation-register form for every builtin MFMA; VALU consumers of a
result then pay one v_accvgpr_read per value -- DESIGN 4.1)"""
import re
import subprocess
import sys

cur, stats = None, {}
for line in open(sys.argv[1]):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1)
        stats[cur] = [0, 0, 0]
    elif cur:
        if 'v_accvgpr_read' in line:
            stats[cur][0] += 1
        elif 'v_accvgpr_write' in line:
            stats[cur][1] += 1
        elif 'v_mfma' in line:
            stats[cur][2] += 1
for key, (reads, writes, mfma) in stats.items():
    if mfma:
        name = subprocess.run(['c++filt', key], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0]
        print(f'{reads:6d} reads {writes:6d} writes {mfma:6d} mfma  {name[:100]}')
