"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv
import collections
import sys

for path in sys.argv[1:]:
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    counts = collections.defaultdict(lambda: collections.defaultdict(int))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row['Kernel_Name']
            if 'anonymous' not in name:
                continue
            short = name.split('::')[-1].split('(')[0]
            sums[short][row['Counter_Name']] += float(row['Counter_Value'])
            counts[short][row['Counter_Name']] += 1
    print('==', path)
    for kernel in sums:
        parts = [f'{c}={sums[kernel][c] / counts[kernel][c]:.4g}' for c in sorted(sums[kernel])]
        print(f'{kernel}: ' + ' '.join(parts))
