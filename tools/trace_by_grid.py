"""Per (kernel, grid) mean duration of a rocprofv3 --kernel-trace CSV: python tools/trace_by_grid.py <dir> [skip_first_fraction]"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
files = glob.glob(root + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[int(len(rows) * skip):]
groups = collections.defaultdict(list)
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name).split('(')[0]
    wg = int(r['Workgroup_Size_X'])
    grid = tuple(int(r['Grid_Size_' + d]) // max(int(r['Workgroup_Size_' + d]), 1) for d in 'XYZ')
    groups[(name, grid, wg)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
total = sum(sum(v) for v in groups.values())
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print(f'{len(rows)} dispatches, kernel time {total / 1e6:.3f} ms in a span of {span / 1e6:.3f} ms')
for (name, grid, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
    print(f'{sum(v) / total * 100:5.1f} %  {len(v):5d} x {sum(v) / len(v) / 1e3:8.1f} us  (min {min(v) / 1e3:7.1f})  grid {grid} x {wg}  {name[:90]}')
