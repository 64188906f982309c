"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel: mean per dispatch, and how many dispatches the
mean is over.  The header names the build the counters belong to (sha256 of ppgs_amd/libppgs_amd.so, which a clean
`make` reproduces byte for byte): bench.py only quotes a summary whose build is the one it has loaded."""
import collections
import csv
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def library_sha16(path=None):
    path = path or os.environ.get('PPGS_AMD_LIB') or os.path.join(ROOT, 'ppgs_amd', 'libppgs_amd.so')
    with open(path, 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


if __name__ == '__main__':
    print(f'# lib_sha256_16={library_sha16()}')
    # the arithmetic mode the counted command ran in (bench.py quotes a summary only for a step of the same mode)
    print(f'# precision={os.environ.get("PMC_PRECISION", "bf16")}')
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
            dispatches = max(counts[kernel].values())
            print(f'{kernel}: dispatches={dispatches} ' + ' '.join(parts))
