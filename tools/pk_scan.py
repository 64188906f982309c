"""Audit of the product kernels against the gfx950 packed-fp32 / MFMA co-residency failure.

Measured with tools/probes/pk_mfma_probe.hip (stand-alone, no PyTorch, nothing of this package): a v_pk_add_f32 /
v_pk_mul_f32 / v_pk_fma_f32 whose SOURCE 1 has its op_sel bit set -- the low result half reads the HIGH half of
source 1: a swap (op_sel:[.,1,.] op_sel_hi:[.,0,.]) or a broadcast of the high half (op_sel:[.,1,.]
op_sel_hi:[.,1,.]) -- intermittently (~1e-5 per executed instruction) computes with the wrong half while ANOTHER
wave of the same SIMD issues v_mfma_*_16x16x32 (bf16 / f16), i8 16x16x64 or a dense 32x32x16 stream: a wave of
another kernel on another HIP stream, or a wave of the same workgroup.  The same selections on source 0 or source 2,
every selection with op_sel[1] = 0, plain fp32 / f64 / packed-f16 arithmetic, DPP and LDS exchanges never failed in
4e8 lane-checks each.  hipcc writes the vulnerable form by itself (SLP-vectorised float pairs with a swapped operand).

Policy (round 5): a kernel is EXPOSED when it contains the form, whatever its register allocation.  (Until round 4 a
kernel with > 256 VGPRs + AGPRs per lane was exempt as "one wave per SIMD".  That bounds only the kernel's OWN
occupancy: a wave of ANOTHER kernel on another stream needs just 512 - vgprs registers and the LDS that is left, and
the 16-register, LDS-free aggressor of tools/probes fits beside a 380- or a 478-register kernel.)  `--strict` exits 1
if any kernel is exposed; tests/test_host.py runs it on the built library.

    python tools/pk_scan.py [--strict] [library.so]      audit the gfx950 code objects inside the built library
                                                         (default ppgs_amd/libppgs_amd.so; llvm-objdump, seconds)
    python tools/pk_scan.py --source [--flags "..."] [file.hip ...]
                                                         recompile sources to ISA with the Makefile's per-file flags
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
PK = re.compile(r'^\s*(v_pk_(?:add|mul|fma)_f32)\s+([^/;]*)')


def vulnerable(rest):
    sel = re.search(r'op_sel:\[([0-9,]+)\]', rest)
    if not sel:
        return False
    bits = [int(v) for v in sel.group(1).split(',')]
    return len(bits) > 1 and bits[1] == 1


def makefile_flags(path):
    """per-file extra device flags of ppgs_amd/csrc/Makefile (FLAGS_<file> = ...)"""
    text = open(os.path.join(ROOT, 'ppgs_amd', 'csrc', 'Makefile')).read()
    m = re.search(r'^FLAGS_' + re.escape(os.path.basename(path)) + r'\s*=\s*(.*)$', text, re.M)
    return m.group(1).split() if m else []


def count(lines, is_start, is_end):
    """{kernel: stats} from an ISA listing"""
    name, stats = None, {}
    for line in lines:
        started = is_start(line)
        if started:
            name = started
            stats[name] = dict(pk=0, vuln=0, mfma=0, vgprs=None)
            continue
        if name is None:
            continue
        if is_end(line):
            name = None
            continue
        t = line.strip()
        if t.startswith('v_mfma') or t.startswith('v_smfmac'):
            stats[name]['mfma'] += 1
        m = PK.match(line)
        if m:
            stats[name]['pk'] += 1
            stats[name]['vuln'] += vulnerable(m.group(2))
    return stats


def scan_source(path, flags):
    with tempfile.NamedTemporaryFile(suffix='.s') as out:
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++20', '-S', '--cuda-device-only',
                        '-Wno-inline-asm', '-Wno-pass-failed'] + flags + [path, '-o', out.name],
                       check=True, stderr=subprocess.DEVNULL)
        text = open(out.name).read()
    lines = text.split('\n')
    stats = count(lines, lambda l: (re.match(r'^(_Z\S+):', l) or [None, None])[1], lambda l: l.startswith('.Lfunc_end'))
    for m in re.finditer(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', text):
        pass
    # registers from the metadata block: .agpr_count precedes .name, .vgpr_count follows
    for block in re.split(r'\n  - ', text):
        name = re.search(r'\.name:\s+(\S+)', block)
        vgpr = re.search(r'\.vgpr_count:\s+(\d+)', block)
        if name and vgpr and name.group(1) in stats:
            stats[name.group(1)]['vgprs'] = int(vgpr.group(1))       # (on gfx950 .vgpr_count is the unified total)
    return stats


def scan_library(path):
    """the gfx950 code objects embedded in a built library"""
    stats = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, 'lib.so')
        shutil.copy(path, local)
        subprocess.run([f'{LLVM}/llvm-objdump', '--offloading', local], check=True, cwd=tmp, capture_output=True)
        for obj in sorted(glob.glob(os.path.join(tmp, '*gfx950*'))):
            listing = subprocess.run([f'{LLVM}/llvm-objdump', '-d', obj], check=True, capture_output=True, text=True).stdout
            part = count(listing.split('\n'),
                         lambda l: (re.match(r'^[0-9a-f]+ <(_Z\S+)>:', l) or [None, None])[1],
                         lambda l: False)
            notes = subprocess.run([f'{LLVM}/llvm-readelf', '--notes', obj], check=True, capture_output=True, text=True).stdout
            for block in re.split(r'\n  - ', notes):
                name = re.search(r'\.name:\s+(\S+)', block)
                vgpr = re.search(r'\.vgpr_count:\s+(\d+)', block)
                if name and vgpr and name.group(1) in part:
                    part[name.group(1)]['vgprs'] = int(vgpr.group(1))
            stats.update(part)
    return stats


def report(stats, exposed):
    for kernel, s in stats.items():
        if s['vgprs'] is None or not (s['pk'] or s['mfma']):      # (not a kernel entry point, or nothing to say)
            continue
        demangled = subprocess.run(['c++filt', kernel], capture_output=True, text=True).stdout.strip()
        short = re.sub(r'\(anonymous namespace\)::', '', demangled).split('(')[0][:80]
        verdict = 'clean' if s['vuln'] == 0 else 'EXPOSED'
        if verdict == 'EXPOSED':
            exposed.append(short)
        print(f'  {short:80s} v_pk_*_f32 {s["pk"]:5d}  vulnerable {s["vuln"]:4d}  mfma {s["mfma"]:5d}  vgprs {s["vgprs"]:3d}  {verdict}')


def main(argv):
    strict = '--strict' in argv
    extra = []
    if '--flags' in argv:
        extra = argv[argv.index('--flags') + 1].split()
        argv = argv[:argv.index('--flags')] + argv[argv.index('--flags') + 2:]
    paths = [a for a in argv if not a.startswith('--')]
    exposed = []
    kernels = 0
    if '--source' in argv:
        for path in paths or sorted(glob.glob(os.path.join(ROOT, 'ppgs_amd', 'csrc', '*.hip'))):
            print(os.path.basename(path))
            stats = scan_source(path, makefile_flags(path) + extra)
            kernels += sum(s['vgprs'] is not None for s in stats.values())
            report(stats, exposed)
    else:
        for path in paths or [os.path.join(ROOT, 'ppgs_amd', 'libppgs_amd.so')]:
            print(os.path.relpath(path, ROOT) if path.startswith(ROOT) else path)
            stats = scan_library(path)
            kernels += sum(s['vgprs'] is not None for s in stats.values())
            report(stats, exposed)
    print(f'{kernels} kernels audited, {len(exposed)} exposed')
    return 1 if strict and (exposed or not kernels) else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
