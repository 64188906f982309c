"""Audit of the built library's ISA for `s_waitcnt vmcnt(0)` that follows a global store inside a kernel: the pattern
the compiler writes when loaded values are used behind stores under control flow (its wait-count insertion then no
longer knows how many stores follow a load and waits for the stores' acknowledgement too -- 300 cycles per store
instruction in ppg_gemm32.hip's epilogue before its loads were forced to land ahead of the first store).

    python tools/wait_scan.py [library]      per kernel: stores, vmcnt(0) waits, vmcnt(0) waits with a store before them
                                             since the previous wait
"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def scan(path):
    """{demangled kernel name: Counter(stores, vmcnt0, vmcnt0_after_store)} of the gfx950 code objects in `path`."""
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, 'lib.so')
        shutil.copy(path, local)
        subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', local], check=True, cwd=tmp, capture_output=True)
        rows = []
        for name in sorted(os.listdir(tmp)):
            if 'gfx950' not in name:
                continue
            listing = subprocess.run([f'{LLVM}/llvm-objdump', '-d', os.path.join(tmp, name)], check=True, capture_output=True, text=True).stdout
            kernel, stats = None, None
            for line in listing.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
                if m:
                    if kernel:
                        rows.append((kernel, stats))
                    kernel, stats = m.group(1), collections.Counter()
                    pending = False
                    continue
                if kernel is None:
                    continue
                text = line.strip()
                if text.startswith(('global_store', 'buffer_store', 'flat_store')):
                    stats['stores'] += 1
                    pending = True
                elif text.startswith('s_waitcnt') and 'vmcnt(0)' in text:
                    stats['vmcnt0'] += 1
                    if pending:
                        stats['vmcnt0_after_store'] += 1
                    pending = False
                elif text.startswith('s_waitcnt') and 'vmcnt' in text:
                    pending = False
            if kernel:
                rows.append((kernel, stats))
        demangle = subprocess.run(['c++filt'], input='\n'.join(k for k, _ in rows), capture_output=True, text=True).stdout.splitlines()
        return {re.sub(r'\(anonymous namespace\)::', '', nice).split('(')[0].replace('void ', ''): stats
                for (_, stats), nice in zip(rows, demangle)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main(path):
    for name, stats in sorted(scan(path).items(), key=lambda item: -item[1]['vmcnt0_after_store']):
        if stats['stores']:
            print(f"{stats['vmcnt0_after_store']:4d} of {stats['vmcnt0']:4d} vmcnt(0) waits behind a store, {stats['stores']:4d} stores   {name[:110]}")


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'ppgs_amd', 'libppgs_amd.so'))
