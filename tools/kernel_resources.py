"""Registers, spills, scratch and LDS of the kernels of a built library (the code objects' metadata notes):

    python tools/kernel_resources.py [library.so] [kernel-name-substring]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'


def kernels(path):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', local], check=True, cwd=tmp, capture_output=True)
        for obj in sorted(os.listdir(tmp)):
            if 'gfx950' not in obj:
                continue
            notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', os.path.join(tmp, obj)], check=True, capture_output=True, text=True).stdout
            for block in re.split(r'\n\s*- \.agpr_count:', notes)[1:]:
                block = '.agpr_count:' + block
                entry = {}
                for key in ('agpr_count', 'vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size',
                            'group_segment_fixed_size', 'name'):
                    m = re.search(r'\.' + key + r':\s*(\S+)', block)
                    if m:
                        entry[key] = m.group(1)
                yield entry
    finally:
        shutil.rmtree(tmp)


if __name__ == '__main__':
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'ppgs_amd', 'libppgs_amd.so')
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    for k in kernels(lib):
        name = subprocess.run(['c++filt', k.get('name', '?')], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0].replace('void ', '')
        if want in name:
            print(f"{name[:70]:70s} vgpr {k.get('vgpr_count'):>4} agpr {k.get('agpr_count'):>4} sgpr {k.get('sgpr_count'):>4} "
                  f"spill v {k.get('vgpr_spill_count'):>4} s {k.get('sgpr_spill_count'):>3} scratch {k.get('private_segment_fixed_size'):>5} lds {k.get('group_segment_fixed_size'):>6}")
