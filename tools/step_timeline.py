"""Timeline of steady-state steps from a rocprofv3 --kernel-trace CSV (two pipelines: which kernels overlap, where a
queue idles between two dependent kernels, what the fork / join costs):

    python tools/step_timeline.py <dir> [steps to print, default 2]

A step starts at a frontend_kernel dispatch.  Per step: every dispatch with its queue, start and end relative to the
step's start, and the idle time of its queue in front of it; then the sums -- busy time per queue, gaps per queue, the
time between the step's last kernel and the next step's first one."""
import csv
import glob
import re
import sys

root = sys.argv[1]
nprint = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([A-Za-z0-9_]+)(<[^(]*>)?', name)
    base = m.group(1) if m else name[:40]
    args = (m.group(2) or '') if m else ''
    args = re.sub(r'Prec(BF16|F16|F32|X2)', r'\1', args)
    return (base + args)[:44]


starts = [i for i, r in enumerate(rows) if 'frontend_kernel' in r['Kernel_Name']]
if len(starts) < nprint + 2:
    sys.exit(f'{len(starts)} frontend dispatches in {len(rows)} rows: nothing to print')
# the last complete steps
steps = [(starts[k], starts[k + 1]) for k in range(len(starts) - 1)]
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
spans = []
for lo, hi in steps:
    t0 = int(rows[lo]['Start_Timestamp'])
    spans.append((int(rows[hi]['Start_Timestamp']) - t0) / 1e3)
tail = spans[len(spans) // 2:]
print(f'{len(steps)} steps; step period over the last half: mean {sum(tail) / len(tail):.1f} us, min {min(tail):.1f}, max {max(tail):.1f}')
for lo, hi in steps[-nprint:]:
    t0 = int(rows[lo]['Start_Timestamp'])
    last_end = {}
    busy = {}
    gaps = {}
    print(f'--- step of {(int(rows[hi]["Start_Timestamp"]) - t0) / 1e3:.1f} us, {hi - lo} dispatches')
    end_all = 0
    for r in rows[lo:hi]:
        q = r[qkey]
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        gap = s - last_end[q] if q in last_end else float('nan')
        grid = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
        print(f'  q{q:>3} {s:8.1f} .. {e:8.1f}  ({e - s:6.1f} us, queue idle {gap:5.1f})  {grid:5d} wg  {short(r["Kernel_Name"])}')
        if q in last_end:
            gaps[q] = gaps.get(q, 0.0) + gap
        busy[q] = busy.get(q, 0.0) + e - s
        last_end[q] = e
        end_all = max(end_all, e)
    period = (int(rows[hi]['Start_Timestamp']) - t0) / 1e3
    for q in sorted(busy):
        print(f'  queue {q}: busy {busy[q]:.1f} us, idle between its kernels {gaps.get(q, 0.0):.1f} us, last end {last_end[q]:.1f}')
    print(f'  last kernel ends {end_all:.1f}, next step starts {period:.1f}: {period - end_all:.1f} us between steps')
